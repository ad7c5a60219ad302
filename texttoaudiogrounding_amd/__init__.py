"""texttoaudiogrounding_amd -- MI355X (gfx950) native hot path of text-to-audio grounding.

log-mel frontend -> Cnn8Rnn audio encoder -> word-embedding text encoder -> frame x phrase
similarity head -> frame BCE, forward + backward, as hand-written HIP kernels behind the
reference's own module interface (``models.audio_encoder.Cnn8Rnn`` ... ``losses.FrameBceLoss``).

``install_aliases()`` registers the sub-packages under the reference's top-level names
(``models``, ``losses``, ``utils``) so that a run_strong.py-style YAML config
(``type: models.audio_text_model.BiEncoder``) resolves to this implementation unchanged.
"""
import sys

__version__ = "0.1.0"


def install_aliases(force: bool = False):
    """Make ``import models.audio_encoder`` etc. resolve to this package (drop-in for the YAML configs)."""
    import importlib
    names = {
        "models": "texttoaudiogrounding_amd.models",
        "models.panns": "texttoaudiogrounding_amd.models.panns",
        "models.utils": "texttoaudiogrounding_amd.models.utils",
        "models.audio_encoder": "texttoaudiogrounding_amd.models.audio_encoder",
        "models.text_encoder": "texttoaudiogrounding_amd.models.text_encoder",
        "models.match": "texttoaudiogrounding_amd.models.match",
        "models.align": "texttoaudiogrounding_amd.models.align",
        "models.audio_text_model": "texttoaudiogrounding_amd.models.audio_text_model",
        "models.cross_encoder": "texttoaudiogrounding_amd.models.cross_encoder",
        "models.sim_pooling": "texttoaudiogrounding_amd.models.sim_pooling",
        "models.hf_modeling_grounding": "texttoaudiogrounding_amd.models.hf_modeling_grounding",
        "losses": "texttoaudiogrounding_amd.losses",
        "utils": "texttoaudiogrounding_amd.utils",
        "utils.eval_util": "texttoaudiogrounding_amd.utils.eval_util",
        "utils.train_util": "texttoaudiogrounding_amd.utils.train_util",
        "utils.build_vocab": "texttoaudiogrounding_amd.utils.build_vocab",
        # ``datasets`` itself stays whatever it is (the reference's directory has no __init__ and is shadowed by the
        # installed HF package of that name); only the sub-module names the YAML configs spell are registered
        "datasets.single_phrase_dataset": "texttoaudiogrounding_amd.datasets.single_phrase_dataset",
        "datasets.collate_function": "texttoaudiogrounding_amd.datasets.collate_function",
        "datasets.text_tokenizer": "texttoaudiogrounding_amd.datasets.text_tokenizer",
    }
    for alias, target in names.items():
        if alias in sys.modules and not force:
            continue
        sys.modules[alias] = importlib.import_module(target)

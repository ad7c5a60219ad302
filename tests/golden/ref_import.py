"""Import the reference (/root/reference) in the BUILD container with stubs for the
third-party modules that are not installed (SURVEY.md section 8c recipe).

Only used by tests/golden/make_golden.py and by the optional ``test_oracle_vs_reference``
tests, which skip when /root/reference is absent (it never exists on the GPU box).
The torchaudio stub delegates to oracle.tag_oracle's restatement of torchaudio's
algorithm (rows F1/F2 are therefore NOT pinned by the reference -- see the oracle header).
"""
import importlib
import os
import sys
import types

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    import torch
    import torch.nn as nn
    import transformers
    from transformers import AutoModel, ClapModel, ClapProcessor  # noqa: F401 (touch lazy loader first)

    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import tag_oracle as O

    class MelSpectrogram(nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0,
                     f_max=None, n_mels=128, norm=None, mel_scale="htk", **kw):
            super().__init__()
            self.p = dict(sample_rate=sample_rate, n_fft=n_fft, win_length=win_length or n_fft,
                          hop_length=hop_length or (win_length or n_fft) // 2, f_min=f_min,
                          f_max=float(f_max if f_max is not None else sample_rate // 2),
                          n_mels=n_mels, norm=norm, mel_scale=mel_scale)
            self.kind = None
            for k, v in O.FRONTEND.items():
                if all(float(v[a]) == float(self.p[a]) if a not in ("norm", "mel_scale") else v[a] == self.p[a]
                       for a in v):
                    self.kind = k
            self.window = torch.hann_window(self.p["win_length"])
            self.fb = O.melscale_fbanks(n_fft // 2 + 1, self.p["f_min"], self.p["f_max"], n_mels,
                                        sample_rate, norm, mel_scale)

        def forward(self, x):
            p = self.p
            spec = torch.stft(x, p["n_fft"], p["hop_length"], p["win_length"],
                              window=self.window.to(x.dtype), center=True, pad_mode="reflect",
                              normalized=False, onesided=True, return_complex=True)
            power = spec.abs().pow(2.0)
            return torch.matmul(power.transpose(-1, -2), self.fb.to(x.dtype)).transpose(-1, -2)

    class AmplitudeToDB(nn.Module):
        def forward(self, x):
            return O.amplitude_to_db(x)

    class SpecAugmentation(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            raise RuntimeError("specaug is off on this path")

    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms", MelSpectrogram=MelSpectrogram,
                          AmplitudeToDB=AmplitudeToDB)
    _stub("torchlibrosa", SpecAugmentation=SpecAugmentation)
    for name in ("toml", "h5py", "fire", "sed_eval", "librosa"):
        _stub(name)
    hy = _stub("hydra")
    hy.utils = _stub("hydra.utils")
    _stub("sentence_transformers", SentenceTransformer=object)
    pe = _stub("psds_eval", PSDSEval=object, plot_psd_roc=None)
    pe.psds = _stub("psds_eval.psds", WORLD="world", PSDSEvalError=Exception)
    sse = _stub("sed_scores_eval")
    sse.intersection_based = _stub("sed_scores_eval.intersection_based")
    sse.utils = _stub("sed_scores_eval.utils")
    sse.utils.auc = _stub("sed_scores_eval.utils.auc", staircase_auc=None)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mp = _stub("matplotlib")
        mp.pyplot = _stub("matplotlib.pyplot")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # texttoaudiogrounding_amd.install_aliases() registers this repo's modules under the reference's names: drop such aliases
    # so that the names below resolve to /root/reference
    for k in [k for k, m in sys.modules.items() if k.split(".")[0] in ("models", "losses", "utils")
              and getattr(m, "__name__", "").startswith("texttoaudiogrounding_amd")]:
        del sys.modules[k]
    mods = {}
    for name in ("models.panns", "models.utils", "models.match", "models.align", "losses",
                 "models.audio_encoder", "models.text_encoder", "models.audio_text_model",
                 "utils.eval_util"):
        mods[name] = importlib.import_module(name)
    return mods

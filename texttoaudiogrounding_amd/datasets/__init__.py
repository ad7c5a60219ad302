"""Host-side readers of the reference's on-disk formats and its batch layout: label JSON ->
frame labels (datasets/single_phrase_dataset.py:20-88), vocabulary pickle -> token ids (datasets/text_tokenizer.py:9-58,
utils/build_vocab.py:7-53), padded batches (datasets/collate_function.py:7-85), fp16 waveform packs
(utils/data/pack_waveform.py:46-73).  These feed ``Runner.forward`` exactly the dict the reference's loaders produce."""
from .collate_function import TextCollate, VarLenPadCollate  # noqa: F401
from .single_phrase_dataset import AudioPhraseDataset, AudioPhraseEvalDataset, phrase_frame_label  # noqa: F401
from .text_tokenizer import DictTokenizer  # noqa: F401
from .waveform_store import WaveformStore, write_waveform_pack  # noqa: F401

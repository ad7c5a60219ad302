"""AudioGrounding (clip, phrase) datasets (mirror of datasets/single_phrase_dataset.py:20-88 in the reference).

Label JSON (README.md:51-59): a list of clips ``{"audiocap_id", "audio_id", "tokens", "phrases": [{"phrase", "start_index",
"end_index", "segments": [[start_s, end_s], ...]}]}``; one dataset item per (clip, phrase).  The training variant adds the
frame-level target: ``n_frame = floor(duration / time_resolution) + 1`` zeros with ``[round(start / res), round(end / res))``
set to 1 per segment (Python ``round`` = half-to-even on the float quotient, like the reference)."""
import json
import math
from typing import Sequence

import numpy as np

from .waveform_store import WaveformStore


def phrase_frame_label(segments: Sequence[Sequence[float]], n_samples: int, sample_rate: int, time_resolution: float) -> np.ndarray:
    """datasets/single_phrase_dataset.py:78-85."""
    n_frame = math.floor(n_samples / sample_rate / time_resolution) + 1
    label = np.zeros(n_frame, dtype=int)
    for start, end in segments:
        label[round(start / time_resolution): round(end / time_resolution)] = 1
    return label


class AudioPhraseEvalDataset:
    def __init__(self, waveform, label, sample_rate: int = 32000):
        self.store = waveform if isinstance(waveform, WaveformStore) else WaveformStore(waveform)
        with open(label) as f:
            self.data = json.load(f)
        self.sample_rate = sample_rate
        self.idxs = [(ai, pi) for ai, clip in enumerate(self.data) for pi in range(len(clip["phrases"]))]

    def __len__(self):
        return len(self.idxs)

    def __getitem__(self, index):
        ai, pi = self.idxs[index]
        clip = self.data[ai]
        phrase = clip["phrases"][pi]
        return {"audio_id": clip["audio_id"], "audiocap_id": clip["audiocap_id"], "start_index": phrase["start_index"],
                "end_index": phrase["end_index"], "waveform": self.store[clip["audio_id"]], "phrase": phrase["phrase"],
                "caption": clip["tokens"]}


class AudioPhraseDataset(AudioPhraseEvalDataset):
    def __init__(self, waveform, label, time_resolution: float = 0.02, sample_rate: int = 32000):
        super().__init__(waveform, label, sample_rate)
        self.time_resolution = time_resolution

    def __getitem__(self, index):
        item = super().__getitem__(index)
        ai, pi = self.idxs[index]
        item["label"] = phrase_frame_label(self.data[ai]["phrases"][pi]["segments"], item["waveform"].shape[0],
                                           self.sample_rate, self.time_resolution)
        return item

"""Waveform packs (utils/data/pack_waveform.py:46-73 in the reference): every clip is stored as a **float16** array keyed by
``audio_id`` and a TSV (``audio_id<TAB>hdf5_path``) maps ids to pack files; readers hand the samples back as float32
(datasets/single_phrase_dataset.py:44-45).

The reference's container is HDF5.  ``h5py`` is not installed in this image, so two containers are read behind one interface:
``*.h5`` / ``*.hdf5`` through h5py when it is importable (same call the reference makes: ``File(path)[audio_id][()]``), and
``*.npz`` -- the same float16 arrays under the same keys in numpy's zip container (``write_waveform_pack``), which is what the
tests and the offline tools here use.  ``fetch_f16`` returns the stored half-precision samples untouched: the device-side
widening to fp32 + zero padding is ``ops.waveform_f16_to_f32_padded`` (bit-identical to ``astype(float32)``; half the
host->device bytes)."""
import csv
import os
from typing import Dict, Iterable, Tuple

import numpy as np


def load_dict_from_csv(path: str, cols: Tuple[str, str]) -> Dict[str, str]:
    """utils/train_util.py:24-27: two columns of a tab-separated file as a dict."""
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f, delimiter="\t"))
    return {r[cols[0]]: r[cols[1]] for r in rows}


def write_waveform_pack(path: str, clips: Iterable[Tuple[str, np.ndarray]], csv_path: str = None) -> str:
    """Store ``(audio_id, waveform)`` pairs as float16 arrays (+ the id -> pack TSV next to it).  ``path`` ending in .npz;
    .h5/.hdf5 needs h5py."""
    clips = [(k, np.asarray(w).astype(np.float16)) for k, w in clips]
    if path.endswith(".npz"):
        np.savez(path, **dict(clips))
    else:
        import h5py
        with h5py.File(path, "w") as store:
            for k, w in clips:
                store[k] = w
    csv_path = csv_path or os.path.splitext(path)[0] + ".csv"
    with open(csv_path, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", lineterminator="\n")
        w.writerow(["audio_id", "hdf5_path"])
        for k, _ in clips:
            w.writerow([k, os.path.abspath(path)])
    return csv_path


class WaveformStore:
    """audio_id -> samples, over the packs listed in a waveform TSV; pack handles are opened once and cached."""

    def __init__(self, waveform_csv: str):
        self.aid_to_pack = load_dict_from_csv(waveform_csv, ("audio_id", "hdf5_path"))
        self._open = {}

    def _pack(self, path: str):
        if path not in self._open:
            if path.endswith(".npz"):
                self._open[path] = np.load(path)
            else:
                try:
                    import h5py
                except ImportError as e:
                    raise RuntimeError(f"{path}: reading HDF5 waveform packs needs h5py (not installed here); repack with "
                                       "write_waveform_pack(<name>.npz, ...)") from e
                self._open[path] = h5py.File(path, "r")
        return self._open[path]

    def fetch_f16(self, audio_id: str) -> np.ndarray:
        arr = self._pack(self.aid_to_pack[audio_id])[audio_id]
        return np.asarray(arr[()] if hasattr(arr, "shape") and not isinstance(arr, np.ndarray) else arr)

    def __getitem__(self, audio_id: str) -> np.ndarray:
        """float32 samples, as the reference's ``np.array(read_from_h5(...), dtype=np.float32)``."""
        return np.array(self.fetch_f16(audio_id), dtype=np.float32)

    def __contains__(self, audio_id: str) -> bool:
        return audio_id in self.aid_to_pack

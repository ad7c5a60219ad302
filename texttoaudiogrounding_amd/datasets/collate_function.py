"""Batch layout of the reference's loaders (mirror of datasets/collate_function.py:7-85): list of item dicts -> dict of
batched values.  Keys in ``pad_keys`` are zero-padded to the longest item (``<key>`` tensor + ``<key>_len`` numpy array);
``TextCollate`` sends the ``text_key`` column (phrases) through the tokenizer (-> ``text`` / ``text_len``) and records
``text_key``; every other column becomes a tensor when its items are numpy arrays, else a numpy array of the values."""
from typing import Dict, List

import numpy as np
import torch


def pad_sequence(data):
    """utils/train_util.py:211-216."""
    seqs = [torch.as_tensor(x) for x in data]
    lens = torch.as_tensor([s.shape[0] for s in seqs]).long()
    return torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True), lens


class VarLenPadCollate:
    def __init__(self, pad_keys=(), sort_key=None):
        self.pad_keys, self.sort_key = list(pad_keys), sort_key

    def _columns(self, batch: List[Dict]):
        if self.sort_key is not None:
            batch.sort(key=lambda item: len(item[self.sort_key]), reverse=True)
        cols: Dict[str, list] = {}
        for item in batch:
            for k, v in item.items():
                cols.setdefault(k, []).append(v)
        return cols

    def _plain(self, values):
        arr = np.array(values)
        return torch.as_tensor(arr) if isinstance(values[0], np.ndarray) else arr

    def __call__(self, batch: List[Dict]):
        cols = self._columns(batch)
        out = dict(cols)
        for k in batch[0].keys():
            if k in self.pad_keys:
                out[k], lens = pad_sequence(cols[k])
                out[f"{k}_len"] = lens.numpy()
            else:
                out[k] = self._plain(cols[k])
        return out


class TextCollate(VarLenPadCollate):
    def __init__(self, tokenizer, text_key="text", pad_keys=(), sort_key=None):
        super().__init__(pad_keys, sort_key)
        self.tokenizer, self.text_key = tokenizer, text_key

    def __call__(self, batch: List[Dict]):
        cols = self._columns(batch)
        out = dict(cols)
        out["text_key"] = self.text_key
        for k in batch[0].keys():
            if k in self.pad_keys:
                out[k], lens = pad_sequence(cols[k])
                out[f"{k}_len"] = lens.numpy()
            elif k == self.text_key:
                out.update(self.tokenizer(cols[k]))          # the phrase strings stay under their own key
            else:
                out[k] = self._plain(cols[k])
        return out

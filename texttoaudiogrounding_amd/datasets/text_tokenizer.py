"""DictTokenizer (mirror of datasets/text_tokenizer.py:9-58 in the reference): whitespace-split phrases -> vocabulary ids,
zero-padded to the longest phrase; ``List[str]`` -> text (B,L) / text_len (B,); ``List[List[str]]`` (the same number of
phrases per clip) -> text (B,N,L) / text_len (B,N).  int64 tensors, pad id 0 (= ``<pad>``)."""
import pickle
from typing import List, Union

import torch

from ..utils.build_vocab import Vocabulary


class DictTokenizer:
    def __init__(self, vocabulary) -> None:
        self.vocabulary = Vocabulary()
        if isinstance(vocabulary, dict):
            self.vocabulary.load_state_dict(vocabulary)
        else:
            with open(vocabulary, "rb") as f:
                self.vocabulary.load_state_dict(pickle.load(f))

    def _encode(self, phrases: List[str]):
        ids = [[self.vocabulary(tok) for tok in p.split()] for p in phrases]
        lens = torch.tensor([len(x) for x in ids], dtype=torch.long)
        text = torch.zeros(len(ids), int(lens.max()) if len(ids) else 0, dtype=torch.long)
        for row, x in zip(text, ids):
            row[:len(x)] = torch.tensor(x, dtype=torch.long)
        return text, lens

    def __call__(self, texts: Union[List[str], List[List[str]]]):
        assert isinstance(texts, list), "the input must be List[str] or List[List[str]]"
        if isinstance(texts[0], str):
            text, lens = self._encode(texts)
        elif isinstance(texts[0], list):
            n = len(texts[0])
            assert all(len(t) == n for t in texts), "the text number in each list must be the same"
            text, lens = self._encode([p for t in texts for p in t])
            text, lens = text.reshape(len(texts), n, -1), lens.reshape(len(texts), n)
        else:
            raise TypeError("the input must be List[str] or List[List[str]]")
        return {"text": text, "text_len": lens}

    def inverse_transform(self, texts):
        out = []
        for row in texts:
            words = []
            for idx in row:
                idx = int(idx)
                if idx == 0:
                    break
                words.append(self.vocabulary.idx2word[idx])
            out.append(" ".join(words))
        return out

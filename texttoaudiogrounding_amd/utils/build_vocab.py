"""Vocabulary of the word-embedding text encoder (mirror of utils/build_vocab.py:7-53 in the reference): a ``word -> index``
dict pickled as it is; ``<pad>`` = 0 and ``<unk>`` = 1 come first, then words in order of first appearance over the label
items' ``tokens`` (or ``caption``) strings.  Unknown words map to ``<unk>``."""
import json
import pickle
from typing import Dict, Iterable, List


class Vocabulary:
    def __init__(self):
        self.word2idx: Dict[str, int] = {}
        self.idx2word: Dict[int, str] = {}
        self.idx = 0

    def add_word(self, word: str):
        if word in self.word2idx:
            return
        self.word2idx[word] = self.idx
        self.idx2word[self.idx] = word
        self.idx += 1

    def __call__(self, word: str) -> int:
        idx = self.word2idx.get(word)
        return self.word2idx["<unk>"] if idx is None else idx

    def __len__(self):
        return len(self.word2idx)

    def state_dict(self):
        return self.word2idx

    def load_state_dict(self, state_dict: Dict[str, int]):
        self.word2idx = state_dict
        self.idx2word = {i: w for w, i in state_dict.items()}
        self.idx = len(state_dict)


def build_vocabulary(items: Iterable[dict]) -> Vocabulary:
    vocab = Vocabulary()
    for special in ("<pad>", "<unk>"):
        vocab.add_word(special)
    for item in items:
        for token in item["tokens" if "tokens" in item else "caption"].split():
            vocab.add_word(token)
    return vocab


def process(items: List[dict], output: str):
    """utils/build_vocab.py:37-53: build and pickle the state dict."""
    vocab = build_vocabulary(items)
    with open(output, "wb") as f:
        pickle.dump(vocab.state_dict(), f)
    return vocab


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("labels", nargs="+", type=str)
    ap.add_argument("output", type=str)
    a = ap.parse_args()
    data = []
    for label in a.labels:
        with open(label) as f:
            data.extend(json.load(f))
    v = process(data, a.output)
    print(f"Total vocabulary size: {len(v)}")

"""On-disk formats and batch layout: texttoaudiogrounding_amd.datasets / utils.build_vocab against
what the REFERENCE's own loaders produced on the same files (tests/golden/formats/, made by make_golden_formats.py from the
imported reference classes).  Host-side; the device-side float16 widening has its own -m gpu test at the bottom."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from tests.golden import make_golden_formats as G

FMT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats")


@pytest.fixture()
def corpus(tmp_path):
    from texttoaudiogrounding_amd.datasets import write_waveform_pack
    csv_path = write_waveform_pack(str(tmp_path / "pack.npz"), G.synthetic_clips())
    return csv_path, os.path.join(FMT, "label.json"), os.path.join(FMT, "vocab.pkl")


def test_vocabulary_pickle_matches_reference():
    from texttoaudiogrounding_amd.utils.build_vocab import Vocabulary, build_vocabulary
    with open(os.path.join(FMT, "vocab.pkl"), "rb") as f:
        ref_state = pickle.load(f)                               # written by the reference's utils.build_vocab.process
    mine = build_vocabulary(G.TRAIN_ITEMS)
    assert mine.state_dict() == ref_state and list(mine.state_dict())[:2] == ["<pad>", "<unk>"]
    v = Vocabulary()
    v.load_state_dict(ref_state)
    assert v("dog") == ref_state["dog"] and v("zeppelin") == ref_state["<unk>"] == 1 and len(v) == len(ref_state)
    assert v.idx2word[v("rain")] == "rain"


@pytest.mark.parametrize("res", [0.02, 0.04])
def test_dataset_collate_tokenizer_batch_equals_reference(corpus, res):
    from texttoaudiogrounding_amd.datasets import AudioPhraseDataset, DictTokenizer, TextCollate
    csv_path, label, vocab = corpus
    gold = np.load(os.path.join(FMT, "batch.npz"))
    tag = f"res{int(res * 100):02d}"
    ds = AudioPhraseDataset(csv_path, label, time_resolution=res, sample_rate=G.SR)
    assert len(ds) == 6
    batch = TextCollate(DictTokenizer(vocab), text_key="phrase", pad_keys=["waveform", "label"])([ds[i] for i in range(len(ds))])
    assert batch["text_key"] == "phrase"
    assert batch["waveform"].dtype == torch.float32 and batch["text"].dtype == torch.int64 and batch["label"].dtype == torch.int64
    assert torch.equal(batch["waveform"], torch.from_numpy(gold["waveform_f16"].astype(np.float32)))
    for k in ("label", "text", "text_len"):
        assert np.array_equal(np.asarray(batch[k]), gold[f"{tag}/{k}"]), k
    for k in ("waveform_len", "label_len", "audiocap_id", "start_index", "end_index"):
        assert isinstance(batch[k], np.ndarray) and np.array_equal(batch[k], gold[f"{tag}/{k}"]), k
    assert list(batch["audio_id"]) == list(gold[f"{tag}/audio_id"]) and list(batch["phrase"]) == list(gold[f"{tag}/phrase"])
    # the frame-label rule on its own: n_frame = floor(duration / res) + 1, [round(start / res), round(end / res)) set
    n0 = G.CLIPS[0][1]
    assert batch["label_len"][0] == int(np.floor(n0 / G.SR / res)) + 1


def test_tokenizer_nested_lists_inverse_and_sorted_collate(corpus):
    from texttoaudiogrounding_amd.datasets import AudioPhraseEvalDataset, DictTokenizer, TextCollate
    csv_path, label, vocab = corpus
    gold = np.load(os.path.join(FMT, "batch.npz"))
    tok = DictTokenizer(vocab)
    m = tok(G.MULTI)
    assert np.array_equal(m["text"].numpy(), gold["multi/text"]) and np.array_equal(m["text_len"].numpy(), gold["multi/text_len"])
    assert tok.inverse_transform(m["text"].reshape(-1, m["text"].shape[-1])) == list(gold["multi/inverse"])
    with pytest.raises(AssertionError):
        tok([["a", "b"], ["c"]])
    ev = AudioPhraseEvalDataset(csv_path, label, sample_rate=G.SR)
    b = TextCollate(tok, text_key="phrase", pad_keys=["waveform"], sort_key="waveform")([ev[i] for i in range(len(ev))])
    assert np.array_equal(b["waveform_len"], gold["sorted/waveform_len"]) and np.array_equal(b["text"].numpy(), gold["sorted/text"])
    assert np.array_equal(b["audiocap_id"], gold["sorted/audiocap_id"]) and "label" not in b


def test_pack_stores_float16_and_label_json_layout(corpus, tmp_path):
    from texttoaudiogrounding_amd.datasets import WaveformStore
    csv_path, label, _ = corpus
    store = WaveformStore(csv_path)
    for aid, w in G.synthetic_clips():
        assert store.fetch_f16(aid).dtype == np.float16 and np.array_equal(store.fetch_f16(aid), w)
        assert store[aid].dtype == np.float32 and np.array_equal(store[aid], w.astype(np.float32))
    with open(label) as f:
        items = json.load(f)
    assert set(items[0]) == {"audiocap_id", "audio_id", "tokens", "phrases"}
    assert set(items[0]["phrases"][0]) == {"phrase", "start_index", "end_index", "segments"}
    h5 = tmp_path / "w.csv"
    h5.write_text("audio_id\thdf5_path\nYabc.wav\t/nonexistent/pack.h5\n")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            WaveformStore(str(h5))["Yabc.wav"]


def test_reference_import_paths_resolve_after_install_aliases():
    """eval.yaml names ``datasets.single_phrase_dataset.AudioPhraseEvalDataset``, ``datasets.collate_function.TextCollate``,
    ``datasets.text_tokenizer.DictTokenizer`` (eg_configs/.../eval.yaml:3-17)."""
    import importlib
    import sys
    import texttoaudiogrounding_amd as T
    T.install_aliases(force=True)
    try:
        assert importlib.import_module("datasets.single_phrase_dataset").AudioPhraseEvalDataset.__module__.startswith("texttoaudiogrounding_amd")
        assert hasattr(importlib.import_module("datasets.collate_function"), "TextCollate")
        assert hasattr(importlib.import_module("datasets.text_tokenizer"), "DictTokenizer")
        assert hasattr(importlib.import_module("utils.build_vocab"), "Vocabulary")
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("models", "losses", "utils")
                  or (k.startswith("datasets.") and getattr(sys.modules[k], "__name__", "").startswith("texttoaudiogrounding_amd"))]:
            del sys.modules[k]


@pytest.mark.gpu
def test_waveform_f16_unpack_on_device_is_bit_exact(dev):
    """Ragged float16 clips -> (B,S) float32 zero-padded on the device == what the host collate builds; odd lengths and
    offsets (misaligned 16-byte vector reads), a 1-sample clip, an explicit S shorter than the longest clip, subnormals."""
    from texttoaudiogrounding_amd import ops
    g = np.random.RandomState(7)
    lens = [320001, 1, 7, 159999, 8, 320000, 15, 16, 17, 4097]
    clips = [(0.3 * g.randn(n)).astype(np.float16) for n in lens]
    clips[2][:] = np.array([6e-8, -6e-8, 65504, -65504, 0.0, -0.0, 1.0], dtype=np.float16)
    wave, wl = ops.waveform_f16_to_f32_padded(clips, dev)
    assert wave.shape == (len(lens), max(lens)) and wl.cpu().tolist() == lens
    want = torch.nn.utils.rnn.pad_sequence([torch.from_numpy(c.astype(np.float32)) for c in clips], batch_first=True)
    assert torch.equal(wave.cpu(), want)
    cut, _ = ops.waveform_f16_to_f32_padded(clips, dev, length=100000)
    assert torch.equal(cut.cpu(), want[:, :100000])
    with pytest.raises(RuntimeError):
        ops.waveform_f16_to_f32_padded([c.astype(np.float32) for c in clips], dev)

"""The reference's post-processing function names (utils/eval_util.py:18-116) as exposed by
texttoaudiogrounding_amd.utils.eval_util, driven exactly like run_strong.py:234-252 and pinned by the fixture the imported
reference produced (tests/golden/postproc.npz).  Host-side: runs without a GPU."""
import numpy as np
import pytest
import torch


def test_reference_evaluate_chain_by_name(golden_dir):
    """median_filter(frame_sim[idx].unsqueeze(0).cpu(), window_size, th)[0] -> connect_clusters(., n_connect) ->
    find_contiguous_regions(.) == the reference's rows for every (row, window, n_connect, threshold) of the fixture."""
    import sys
    import texttoaudiogrounding_amd as T
    T.install_aliases(force=True)
    from utils import eval_util                                   # the reference's import path (run_strong.py:20)
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "losses", "utils")]:
        del sys.modules[k]                                        # do not leak the aliases into other tests
    gold = np.load(f"{golden_dir}/postproc.npz")
    rows, lens, th, segs = gold["rows"], gold["row_len"], gold["thresholds"], gold["segments"]
    off = np.concatenate([[0], np.cumsum(lens)])
    count = 0
    for ri in range(len(lens)):
        x = torch.from_numpy(rows[off[ri]:off[ri + 1]])
        for window in (1, 3, 4):
            for n_connect in (7, 13):
                sub = segs[(segs[:, 0] == ri) & (segs[:, 2] == window) & (segs[:, 3] == n_connect)]
                for ti, t in enumerate(th):
                    filtered = eval_util.median_filter(x.unsqueeze(0).cpu(), window_size=window, threshold=t)[0]
                    change = eval_util.find_contiguous_regions(eval_util.connect_clusters(filtered, n_connect))
                    want = sub[sub[:, 1] == ti][:, 4:6]
                    assert np.array_equal(np.asarray(change).reshape(-1, 2), want), (ri, window, n_connect, ti)
                    count += len(want)
    assert count == len(segs)


def test_stage_functions_match_their_third_party_definitions():
    """binarize == sklearn.preprocessing.binarize, median_filter == scipy.ndimage.median_filter for the three shape rules of
    utils/eval_util.py:55-63, connect_ on hand-checked clusters."""
    from texttoaudiogrounding_amd.utils import eval_util as E
    pre = pytest.importorskip("sklearn.preprocessing")
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.RandomState(3)
    x2 = rng.rand(1, 97).astype(np.float32)
    x2t = rng.rand(61, 3).astype(np.float32)
    x3 = rng.rand(2, 45, 3).astype(np.float32)
    for th in (0.11, 0.5, np.float64(0.37)):
        assert np.array_equal(E.binarize(x2, th), pre.binarize(x2, threshold=th))
        assert np.array_equal(E.binarize(x3, th), np.array([pre.binarize(s, threshold=th) for s in x3]))
        for w in (1, 2, 3, 4, 5, 8):
            assert np.array_equal(E.median_filter(x2, w, th), ndi.median_filter(pre.binarize(x2, threshold=th), size=(1, w)))
            assert np.array_equal(E.median_filter(x2t, w, th), ndi.median_filter(pre.binarize(x2t, threshold=th), size=(w, 1)))
            b3 = np.array([pre.binarize(s, threshold=th) for s in x3])
            assert np.array_equal(E.median_filter(x3, w, th), ndi.median_filter(b3, size=(1, w, 1)))
    assert E.connect_([], n=3) == []
    assert E.connect_([(1, 5), (7, 10)], n=1) == [(1, 5), (7, 10)]
    assert E.connect_([(1, 5), (7, 10)], n=2) == [(1, 10)]
    assert E.connect_([(0, 2), (3, 4), (9, 12), (13, 14)], n=1) == [(0, 4), (9, 14)]
    v = np.array([0, 1, 1, 0, 0, 1, 0, 0, 0, 1, 1], dtype=np.float32)
    assert np.array_equal(E.find_contiguous_regions(v), [[1, 3], [5, 6], [9, 11]])
    assert np.array_equal(E.connect_clusters_(v, 2), [0, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1])
    assert np.array_equal(E.connect_clusters(v, 3), [0] + [1] * 10)
    m = np.stack([v, v[::-1]], axis=1)                              # (time, classes): along axis -2
    got = E.connect_clusters(m, 2)
    assert np.array_equal(got[:, 0], E.connect_clusters_(v, 2)) and np.array_equal(got[:, 1], E.connect_clusters_(v[::-1], 2))
    assert E.find_contiguous_regions(np.zeros(5)).shape == (0, 2)
    assert np.array_equal(E.find_contiguous_regions(np.ones(5)), [[0, 5]])


def test_predictions_to_time_dataframe_and_array():
    from texttoaudiogrounding_amd.utils import eval_util as E
    pd = pytest.importorskip("pandas")
    df = pd.DataFrame({"filename": ["a", "b"], "onset": [2, 10], "offset": [5, 12]})
    out = E.predictions_to_time(df, 0.04)
    assert np.allclose(out.onset, [0.08, 0.4]) and np.allclose(out.offset, [0.2, 0.48])
    assert len(E.predictions_to_time(pd.DataFrame({"onset": [], "offset": []}), 0.04)) == 0
    assert np.allclose(E.predictions_to_time(np.array([[2, 5]]), 0.04), [[0.08, 0.2]])

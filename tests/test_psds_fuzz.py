"""``grounding_eval.psds_intersection`` fuzzed against a deliberately NAIVE second implementation written from the metric's
definition (single-class intersection-based PSDS, alpha_ct = alpha_st = 0; Bilen et al. 2020, as utils/eval_util.py:136-225
asks psds_eval for it): time is discretised to integer milliseconds, intersections are counted by walking the milliseconds
of every event in pure Python, and the area is integrated by scanning a sorted list -- no numpy broadcasting, no shared
helper with the product code.  psds_eval itself is absent (no network): **parity with that package stays unpinned**; this
test pins the product function against an independent reading of the same definition on thousands of random tables."""
import numpy as np
import pytest

from texttoaudiogrounding_amd.utils import grounding_eval as GE


class Tie(Exception):
    """A coverage ratio equals its threshold exactly in integer milliseconds: the float ratio of the product code (and of
    psds_eval) may land on either side of it -- such a case has no implementation-independent answer."""


def naive_psds(prediction_tables, ground_truth, durations, dtc, gtc, max_efpr):
    def ms(x):
        return int(round(x * 1000))

    def at_least(covered, frac, length):
        if abs(covered - frac * length) < 1e-6:
            raise Tie()
        return covered >= frac * length

    hours = sum(durations[f] for f in ground_truth) / 3600.0
    n_gt = sum(len(v) for v in ground_truth.values())
    points = []
    for det_table in prediction_tables.values():
        tp, fp = 0, 0
        for f in set(det_table) | set(ground_truth):
            dets = [(ms(a), ms(b)) for a, b in det_table.get(f, [])]
            gts = [(ms(a), ms(b)) for a, b in ground_truth.get(f, [])]
            relevant = []
            for on, off in dets:
                covered = 0
                for t in range(on, off):                       # each millisecond of the detection, against every ground truth
                    for g_on, g_off in gts:
                        if g_on <= t < g_off:
                            covered += 1
                if gts and at_least(covered, dtc, off - on):
                    relevant.append((on, off))
                else:
                    fp += 1
            for g_on, g_off in gts:
                covered = 0
                for t in range(g_on, g_off):
                    for on, off in relevant:
                        if on <= t < off:
                            covered += 1
                if at_least(covered, gtc, g_off - g_on):
                    tp += 1
        points.append((fp / hours, tp / n_gt))
    # PSD-ROC: best TPR reachable at an eFPR <= x, as a step function through the origin
    xs = sorted({0.0} | {round(x, 6) for x, _ in points})
    best = []
    for x in xs:
        best.append(max([0.0] + [y for px, y in points if round(px, 6) <= x]))
    if max_efpr is None:
        max_efpr = xs[-1]
    if max_efpr <= 0:
        return best[-1]
    area = 0.0
    for i, x in enumerate(xs):
        if x >= max_efpr:
            break
        nxt = xs[i + 1] if i + 1 < len(xs) else max_efpr
        area += (min(nxt, max_efpr) - x) * best[i]
    return area / max_efpr


def random_case(rng, n_files, n_ops):
    """Non-overlapping events on a 10 ms grid per file (ground truths among themselves, detections among themselves --
    what post-processing produces: disjoint segments per (clip, threshold))."""
    def events(k, dur):
        cuts = np.sort(rng.choice(np.arange(1, int(dur * 100)), size=2 * k, replace=False))
        return np.array([[cuts[2 * i] / 100.0, cuts[2 * i + 1] / 100.0] for i in range(k)]).reshape(-1, 2)

    files = [f"f{i}" for i in range(n_files)]
    durations = {f: float(rng.randint(4, 11)) for f in files}
    gt = {f: events(rng.randint(1, 4), durations[f]) for f in files}
    tables = {}
    for o in range(n_ops):
        tables[float(o)] = {f: events(rng.randint(0, 5), durations[f]) for f in files if rng.rand() < 0.9}
    return tables, gt, durations


@pytest.mark.parametrize("seed", range(12))
def test_psds_intersection_equals_naive_definition(seed):
    rng = np.random.RandomState(100 + seed)
    tables, gt, durations = random_case(rng, n_files=rng.randint(1, 5), n_ops=rng.randint(1, 9))
    checked = 0
    for dtc, gtc in ((0.5, 0.5), (0.1, 0.1), (0.7, 0.3), (0.503, 0.497)):
        for max_efpr in (None, 400.0, 1000.0, 50.0):
            got = GE.psds_intersection(tables, gt, durations, dtc, gtc, max_efpr)
            try:
                want = naive_psds(tables, gt, durations, dtc, gtc, max_efpr)
            except Tie:
                continue
            assert abs(got - want) < 1e-9, (seed, dtc, gtc, max_efpr, got, want)
            checked += 1
    assert checked >= 4                                    # (0.503, 0.497) cannot tie on a 10 ms grid


def test_psds_fuzz_cases_are_not_degenerate():
    vals = []
    for seed in range(12):
        rng = np.random.RandomState(100 + seed)
        tables, gt, durations = random_case(rng, n_files=rng.randint(1, 5), n_ops=rng.randint(1, 9))
        vals.append(GE.psds_intersection(tables, gt, durations, 0.1, 0.1, None))
    assert max(vals) > 0.2 and len({round(v, 6) for v in vals}) > 6, vals

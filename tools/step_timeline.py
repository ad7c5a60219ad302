#!/usr/bin/env python3
"""Timeline of ONE steady-state training step from a `rocprofv3 --kernel-trace` CSV (k_kernel_trace.csv): which kernels ran on which
HIP queue, when, and what ran beside them.  Used to read the two-stream schedule of the bf16 mode (main chain + weight-gradient
side stream): how long the main queue sits idle, how long the short kernels of the main chain wait beside the side stream's
long-lived workgroups, what the chip does during those waits.

    rocprofv3 --kernel-trace --output-format csv -d D -o k -- python bench.py --dtype bf16 --steps 4 --warmup 2 --no-...
    python tools/step_timeline.py D/k_kernel_trace.csv [step_index_from_end=2] [--rows]

A step = the dispatches between two consecutive `adam_kernel` launches.  (rocprofv3 of this image segfaults at EXIT after the trace
files are written: the CSV is complete.  The traced host enqueues ~3 x slower than an untraced one, so gaps in the traces of the
5-10 ms steps are partly the tracer's; the kernel durations and the cross-queue overlaps are the measurement.)
"""
import csv
import sys


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    cut = n.find("(")
    if cut > 0 and not n.startswith("void at::"):
        n = n[:cut]
    return n[:58]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 2
    show_rows = "--rows" in sys.argv
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
        r["wgs"] = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * \
                   (int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) * \
                   (int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    if len(adam) < back + 1:
        print("not enough steps in the trace"); return
    lo, hi = adam[-back - 1] + 1, adam[-back] + 1
    step = rows[lo:hi]
    t0 = step[0]["s"]; t1 = max(r["e"] for r in step)
    queues = sorted({r["Queue_Id"] for r in step}, key=lambda q: -sum(1 for r in step if r["Queue_Id"] == q))
    print(f"step of {len(step)} dispatches, {(t1 - t0) / 1e6:.3f} ms wall, queues {queues} "
          f"({', '.join(str(sum(1 for r in step if r['Queue_Id'] == q)) for q in queues)} dispatches)")
    # union busy time, per-queue busy time
    ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
    busy = 0; depth = 0; last = t0; both = 0
    for t, d in ev:
        if depth > 0: busy += t - last
        if depth > 1: both += t - last
        depth += d; last = t
    print(f"chip busy (>= 1 kernel in flight) {busy / 1e6:.3f} ms, >= 2 kernels in flight {both / 1e6:.3f} ms, "
          f"idle {(t1 - t0 - busy) / 1e6:.3f} ms")
    for q in queues:
        qs = [r for r in step if r["Queue_Id"] == q]
        qb = sum(r["e"] - r["s"] for r in qs)
        gaps = [(b["s"] - a["e"]) for a, b in zip(qs, qs[1:])]
        print(f"queue {q}: {len(qs)} kernels, sum of durations {qb / 1e6:.3f} ms, first start +{(qs[0]['s'] - t0) / 1e6:.3f} ms, "
              f"last end +{(qs[-1]['e'] - t0) / 1e6:.3f} ms, gaps between its kernels {sum(g for g in gaps if g > 0) / 1e6:.3f} ms")
    # short kernels (few workgroups) of the main queue: duration in the step vs the fastest launch of the same kernel in the trace
    fastest = {}
    for r in rows:
        k = (r["Kernel_Name"], r["wgs"])
        fastest[k] = min(fastest.get(k, 1 << 62), r["e"] - r["s"])
    main_q = queues[0]
    stretch = {}
    for r in step:
        if r["Queue_Id"] != main_q: continue
        k = (r["Kernel_Name"], r["wgs"])
        d = r["e"] - r["s"]
        s = stretch.setdefault(short(r["Kernel_Name"]), [0, 0, 0])
        s[0] += 1; s[1] += d; s[2] += fastest[k]
    print("\nmain queue, per kernel: launches, time in this step, time at each launch's fastest observation, excess")
    for n, (c, d, f) in sorted(stretch.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:24]:
        print(f"  {n:58s} {c:3d}  {d / 1e3:9.1f} us  {f / 1e3:9.1f} us  +{(d - f) / 1e3:8.1f}")
    tot_d = sum(v[1] for v in stretch.values()); tot_f = sum(v[2] for v in stretch.values())
    print(f"  main queue total {tot_d / 1e6:.3f} ms against {tot_f / 1e6:.3f} ms at fastest observations")
    if show_rows:
        print("\n  +start ms   dur us  queue  wgs   kernel        (| = side-queue kernels in flight at its start)")
        for r in step:
            beside = [short(o["Kernel_Name"])[:28] for o in step if o is not r and o["Queue_Id"] != r["Queue_Id"]
                      and o["s"] <= r["s"] < o["e"]]
            print(f"  {(r['s'] - t0) / 1e6:8.3f} {(r['e'] - r['s']) / 1e3:8.1f}  q{r['Queue_Id']:>2} {r['wgs']:6d}  "
                  f"{short(r['Kernel_Name']):58s} | {', '.join(beside)}")


if __name__ == "__main__":
    main()

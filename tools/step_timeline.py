"""Timeline of ONE training step from a rocprofv3 kernel trace: every launch with its duration and the idle gap in front of it.

usage: python tools/step_timeline.py KERNEL_TRACE.csv [delimiter-substring] [--all]
  (rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...  ->  DIR/**/*_kernel_trace.csv)

A step ends with the launch whose name contains the delimiter (default: the optimizer's multi-tensor kernel).  The step
printed is the median-length one of the trace's last ten; the summary gives launches, summed kernel time, summed gaps and
the per-kernel totals of that step.  Profiling aid only (not product code)."""
import csv, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:90]


def main():
    path = sys.argv[1]
    delim = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "optim_multi_kernel"
    show_all = "--all" in sys.argv
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if delim in r[2]]
    steps = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
    steps = steps[-10:]
    if not steps:
        sys.exit("no complete step found")
    steps.sort(key=lambda ab: rows[ab[1] - 1][1] - rows[ab[0] - 1][1])
    a, b = steps[len(steps) // 2]
    prev_end = rows[a - 1][1]
    t0 = prev_end
    ksum = gsum = 0
    per = defaultdict(lambda: [0, 0, 0])
    for s, e, n in rows[a:b]:
        gap = s - prev_end
        if show_all:
            print(f"{(s - t0) / 1e3:10.1f} us  gap {gap / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {short(n)}")
        ksum += e - s
        gsum += max(gap, 0)
        p = per[short(n)]
        p[0] += 1
        p[1] += e - s
        p[2] += max(gap, 0)
        prev_end = max(prev_end, e)
    print(f"step: {b - a} launches, wall {(prev_end - t0) / 1e6:.3f} ms, kernels {ksum / 1e6:.3f} ms, gaps {gsum / 1e6:.3f} ms")
    for n, (c, t, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e3:9.1f} us  x{c:3d}  avg {t / c / 1e3:7.1f}  gaps-before {g / 1e3:7.1f}  {n}")


if __name__ == "__main__":
    main()

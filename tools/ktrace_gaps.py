"""Idle time of the GPU between consecutive kernels of a rocprofv3 kernel-trace CSV, over the LAST `frac` of the trace
(the timed steps): total busy time, total gap time, and the gaps summed by the kernel they FOLLOW -- a host round trip
(synchronise, read back, enqueue) shows up as a long gap behind the kernel whose result the host waited for.
Usage: python tools/ktrace_gaps.py <kernel_trace.csv> [frac=0.5]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name") or r.get("Name")
        t0 = int(r.get("Start_Timestamp") or r.get("Start"))
        t1 = int(r.get("End_Timestamp") or r.get("End"))
        short = re.sub(r"HIP_vector_type<(\w+), (\d)u>", r"\1\2", name)
        short = re.sub(r"\(.*$", "", short)
        rows.append((t0, t1, short))
rows.sort()
rows = rows[int(len(rows) * (1.0 - frac)):]
busy = sum(t1 - t0 for t0, t1, _ in rows)
gaps = defaultdict(lambda: [0, 0, 0, []])
total_gap = 0
for (a0, a1, an), (b0, b1, bn) in zip(rows, rows[1:]):
    g = max(b0 - a1, 0)
    total_gap += g
    e = gaps[an]
    e[0] += g
    e[1] += 1
    e[2] = max(e[2], g)
    e[3].append(g)
span = rows[-1][1] - rows[0][0]
print(f"# last {frac:.0%} of the trace: {len(rows)} dispatches, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, "
      f"gaps {total_gap / 1e6:.3f} ms ({100.0 * total_gap / max(span, 1):.1f} % of the span)")
print("# gap behind kernel: total us, count, mean us, max us, MEDIAN us (the steady state: one-off host waits do not move it)")
for name, (g, n, mx, gl) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    med = sorted(gl)[len(gl) // 2] if gl else 0
    print(f"{g / 1e3:10.1f} {n:6d} {g / max(n, 1) / 1e3:8.2f} {mx / 1e3:8.1f} {med / 1e3:8.2f}  {name[:120]}")
# optional third argument "at=<fraction>": the dispatch sequence around that point of the WHOLE trace (start offset, duration, gap behind)
for a in sys.argv[3:]:
    if a.startswith("at="):
        allrows = []
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("Name")
                allrows.append((int(r.get("Start_Timestamp") or r.get("Start")), int(r.get("End_Timestamp") or r.get("End")), re.sub(r"\(.*$", "", name)[:70]))
        allrows.sort()
        k = int(len(allrows) * float(a[3:]))
        win = allrows[k:k + 36]
        print(f"# dispatches {k} .. {k + len(win)} of {len(allrows)}: start us (relative), duration us, gap to the next us, kernel")
        for (t0, t1, nm), nxt in zip(win, win[1:] + [None]):
            print(f"{(t0 - win[0][0]) / 1e3:10.1f} {(t1 - t0) / 1e3:8.1f} {((nxt[0] - t1) / 1e3) if nxt else 0.0:8.1f}  {nm}")

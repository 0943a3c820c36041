"""Per-kernel mean duration over the LAST n dispatches of each kernel in a rocprofv3 kernel-trace CSV
(the --stats table averages over the whole run, settling steps included).
Usage: python tools/ktrace_tail.py <kernel_trace.csv> [n=50]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = defaultdict(list)
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name") or r.get("Name")
        t0 = int(r.get("Start_Timestamp") or r.get("Start"))
        t1 = int(r.get("End_Timestamp") or r.get("End"))
        rows[name].append((t0, t1 - t0))
out = []
for name, v in rows.items():
    v.sort()
    tail = [d for _, d in v[-n:]]
    short = re.sub(r"HIP_vector_type<(\w+), (\d)u>", r"\1\2", name)
    short = re.sub(r"\(.*$", "", short)
    out.append((sum(tail) / len(tail), len(v), short))
out.sort(reverse=True)
print(f"# mean of the last {n} dispatches per kernel (ns), total dispatches, kernel")
for mean, calls, name in out[:16]:
    print(f"{mean:12.0f} {calls:7d}  {name[:150]}")

#!/usr/bin/env python
"""Per-kernel summary (count / avg / min / max / total) from a rocprofv3 rocpd
SQLite database or a kernel_trace CSV.  Usage: rocpd_summary.py <db-or-csv> [out.txt]"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    q = ("select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    return [(n, float(t)) for n, t in db.execute(q)]


def rows_from_csv(path):
    out = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            out.append((r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    for n, t in rows:
        agg[n].append(t)
    tot = sum(sum(v) for v in agg.values())
    lines = [f"# {path}: {len(rows)} dispatches, {tot / 1e6:.3f} ms of kernel time",
             f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s} {'%':>6s}"]
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{n[:100]:100s} {len(v):6d} {sum(v) / len(v) / 1e3:10.1f} {min(v) / 1e3:10.1f} "
                     f"{max(v) / 1e3:10.1f} {sum(v) / 1e6:9.3f} {100 * sum(v) / tot:6.2f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()

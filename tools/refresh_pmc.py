#!/usr/bin/env python
"""Build profiles/pmc_traffic.json from the rocprofv3 PMC passes of tools/gpu_pmc.sh (run on the GPU box).

  python tools/refresh_pmc.py gpurun_out/pmc_<tag> gpurun_out/pmc_traffic.json [--settled gpurun_out/pmc_<tag2>] [--tail 20]

The first directory holds the passes over the from-rest bench run, --settled the passes over a `--settle 2000` run; of every
kernel only the LAST --tail dispatches are averaged (the timed steps; a settled run's counters would otherwise be
averaged over the 2,000 settling steps as well).

Per kernel of the WCSPH step: FETCH_SIZE / WRITE_SIZE (KB per launch), SQ_INSTS_VALU (wave-level VALU instructions
per launch), SQ_ACTIVE_INST_VALU and SQ_LDS_IDX_ACTIVE shares of the kernel's cycles.  The file carries the
fingerprint of the kernel sources it was measured on (sph_taichi_amd.build._fingerprint()); bench.py quotes its numbers
only while that fingerprint equals the one of the library it runs -- a stale file yields `traffic: null`.
"""
from __future__ import annotations

import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = {2: "GM_DENSITY", 3: "GM_DENSITY_EOS", 6: "GM_FORCE_FUSED", 14: "GM_FORCE_FUSED_U",
         # DFSPH (--solver dfsph passes: profiles/pmc_traffic_dfsph.json)
         7: "GM_DF_DENSITY", 8: "GM_DF_FACTOR", 9: "GM_DF_DENSITY_CHANGE", 10: "GM_DF_DENSITY_ADV", 11: "GM_DF_DIV_ITER",
         12: "GM_DF_PRESSURE_ITER", 13: "GM_DF_NONPRESSURE", 15: "GM_DF_DIV_ITER_U", 16: "GM_DF_PRESSURE_ITER_U"}


def short_name(kernel: str) -> str:
    m = re.search(r"k_gather_brick<\s*\(?(?:GatherMode\))?(\d+)", kernel)
    if m:
        return f"k_gather_brick<{MODES.get(int(m.group(1)), 'mode ' + m.group(1))}>"
    m = re.search(r"(k_[a-z_0-9]+)", kernel)
    return m.group(1) if m else kernel[:60]


def collect(src, tail):
    per = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [(dispatch id, value)]
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                per[short_name(r["Kernel_Name"])][r["Counter_Name"]].append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"])))
    agg = {}
    for k, cs in per.items():
        agg[k] = {}
        for cname, v in cs.items():
            v.sort()
            agg[k][cname] = [x for _, x in v[-tail:]] if tail > 0 else [x for _, x in v]
    mean = lambda v: sum(v) / len(v) if v else None
    kernels = {}
    for k, c in agg.items():
        if not k.startswith("k_") or "fillBuffer" in k:
            continue
        cyc = mean(c.get("GRBM_GUI_ACTIVE", []))
        cyc = cyc / 8.0 if cyc else None          # summed over the 8 XCDs
        e = {"dispatches": max(len(v) for v in c.values())}
        if c.get("FETCH_SIZE"): e["fetch_kb"] = round(mean(c["FETCH_SIZE"]), 1)
        if c.get("WRITE_SIZE"): e["write_kb"] = round(mean(c["WRITE_SIZE"]), 1)
        if c.get("SQ_INSTS_VALU"): e["valu_wave_insts"] = round(mean(c["SQ_INSTS_VALU"]), 1)
        if c.get("SQ_INSTS_LDS"): e["lds_wave_insts"] = round(mean(c["SQ_INSTS_LDS"]), 1)
        if c.get("SQ_INSTS_SALU"): e["salu_wave_insts"] = round(mean(c["SQ_INSTS_SALU"]), 1)
        if c.get("SQ_INSTS_VMEM_RD"): e["vmem_rd_wave_insts"] = round(mean(c["SQ_INSTS_VMEM_RD"]), 1)
        if c.get("SQ_INSTS_VMEM_WR"): e["vmem_wr_wave_insts"] = round(mean(c["SQ_INSTS_VMEM_WR"]), 1)
        if c.get("SQ_WAVE_CYCLES"): e["wave_cycles_quad"] = round(mean(c["SQ_WAVE_CYCLES"]), 1)
        if c.get("SQ_WAIT_ANY"): e["wait_any_quad"] = round(mean(c["SQ_WAIT_ANY"]), 1)
        if cyc:
            e["kernel_cycles"] = round(cyc, 1)
            if c.get("SQ_ACTIVE_INST_VALU"): e["valu_busy_frac"] = round(mean(c["SQ_ACTIVE_INST_VALU"]) * 4 / 1024 / cyc, 3)
            if c.get("SQ_LDS_IDX_ACTIVE"): e["lds_active_frac"] = round(mean(c["SQ_LDS_IDX_ACTIVE"]) / 256 / cyc, 3)
        if c.get("SQ_LDS_BANK_CONFLICT") and c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share"] = round(mean(c["SQ_LDS_BANK_CONFLICT"]) / max(mean(c["SQ_LDS_IDX_ACTIVE"]), 1), 3)
        if c.get("TCC_HIT_sum") and c.get("TCC_MISS_sum"):
            h, m_ = mean(c["TCC_HIT_sum"]), mean(c["TCC_MISS_sum"])
            e["l2_hit_rate"] = round(h / max(h + m_, 1), 3)
        kernels[k] = e
    return kernels


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--settled", default="")
    ap.add_argument("--tail", type=int, default=20)
    ap.add_argument("--workload", default="c3p_uniform_1.75M")
    ap.add_argument("--solver", default="wcsph", help="dfsph: the passes were taken over `bench.py --solver dfsph` (the file then "
                    "averages ALL dispatches of a kernel unless --tail says otherwise: sweeps enqueued past convergence leave at "
                    "once and would bias a short tail)")
    a = ap.parse_args()
    from sph_taichi_amd import build
    rest = collect(a.src, a.tail)
    states = {"rest": rest}
    if a.settled:
        states["settled"] = collect(a.settled, a.tail)
    out = {
        "_comment": "per-launch counters from rocprofv3 PMC passes (tools/gpu_pmc.sh: each --pmc set in its own run, "
                    "kernel-trace only), mean over each kernel's last --tail dispatches. HBM bytes = FETCH_SIZE[KB]*1024*2 + "
                    "WRITE_SIZE[KB]*1024 (x2 on the read side: "
                    "gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md; checked in round 1 on k_advect / "
                    "k_stable_scatter). valu_busy_frac = SQ_ACTIVE_INST_VALU*4 / 1024 SIMDs / kernel cycles "
                    "(GRBM_GUI_ACTIVE / 8 XCDs); lds_active_frac = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles.",
        "kernel_fingerprint": build._fingerprint(),
        "workload": a.workload,
        "solver": a.solver,
        "source": os.path.basename(os.path.normpath(a.src)) + (" + " + os.path.basename(os.path.normpath(a.settled)) if a.settled else ""),
        "tail": a.tail,
        "kernels": rest,
        "states": states,
    }
    os.makedirs(os.path.dirname(os.path.abspath(a.dst)), exist_ok=True)
    json.dump(out, open(a.dst, "w"), indent=1)
    brief = {s_: {k: {f: v.get(f) for f in ("fetch_kb", "write_kb", "valu_wave_insts", "valu_busy_frac", "lds_active_frac")}
                  for k, v in ks.items() if "gather" in k} for s_, ks in states.items()}
    print(json.dumps(brief, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""VGPR / SGPR / scratch / occupancy / LDS of every k_gather_brick instance (and the other kernels of a file), from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  Usage: python tools/kernel_resources.py [file.hip] [-DSPH_PROFILE ...]"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {3: "DENSITY_EOS", 6: "FORCE_FUSED", 7: "DF_DENSITY", 8: "DF_FACTOR", 9: "DF_DENSITY_CHANGE", 10: "DF_DENSITY_ADV",
         11: "DF_DIV_ITER", 12: "DF_PRESSURE_ITER", 13: "DF_NONPRESSURE", 14: "FORCE_FUSED_U", 15: "DF_DIV_ITER_U",
         16: "DF_PRESSURE_ITER_U", 2: "DENSITY", 4: "NONPRESSURE", 5: "PRESSURE"}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "sph_gather.hip"
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    sys.path.insert(0, ROOT)
    from sph_taichi_amd import build
    flags = [f for f in build.FLAGS if f != "-shared"]
    cmd = [build.hipcc()] + flags + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(build.CSRC, src), "-o", "/dev/null"]
    t = subprocess.run(cmd, capture_output=True, text=True).stderr
    print(f"{'kernel':58s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'waves/SIMD':>10s} {'LDS':>6s}")
    for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
        name = b.split("\n")[0].strip().split(" ")[0]
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        m = re.search(r"k_gather_brick<\(GatherMode\)(\d+), BrickCfg<[^>]*>, (\d+)>", dem)
        label = f"k_gather_brick<{MODES.get(int(m.group(1)), m.group(1))}, var {m.group(2)}>" if m else dem.split("(")[0][:58]
        vals = [g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")]
        print(f"{label:58s} {vals[0]:>5s} {vals[1]:>5s} {vals[2]:>7s} {vals[3]:>10s} {vals[4]:>6s}")


if __name__ == "__main__":
    main()

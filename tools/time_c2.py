#!/usr/bin/env python
"""Where does the time of the deep C2 parity case go?  Oracle (OpenMP) vs HIP per block of steps, with the
neighbourhood statistics of the HIP side (sph_get_stats) before and after the floor impact."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes, test_gpu_fullsize as fs
from sph_taichi_amd import _lib
from oracle.oracle import max_threads

threads = int(sys.argv[1]) if len(sys.argv) > 1 else max_threads()
sd = fs.dragon_bath_scene()
cfg, sc = scenes.build(sd)
o = scenes.make_oracle(cfg, sc, omp_threads=threads)
ps, solver = scenes.make_ps(sd)
t = time.perf_counter(); o.initialize(); t_oi = time.perf_counter() - t
t = time.perf_counter(); solver.initialize(); ps.sync(); t_hi = time.perf_counter() - t
print(f"threads {threads}  init: oracle {t_oi:.2f} s, hip {t_hi:.2f} s", flush=True)
for blk in range(6):
    n = 50
    t = time.perf_counter(); ms = o.step(n); t_o = time.perf_counter() - t
    t = time.perf_counter(); solver.step(n); ps.sync(); t_h = time.perf_counter() - t
    st = _lib.SphStats(); ps._call("sph_get_stats", st)
    err = scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x"))
    print(f"steps {blk*n+n:4d}: oracle {t_o/n*1e3:8.1f} ms/step {[round(m/n,1) for m in ms]}  hip {t_h/n*1e3:7.3f} ms/step  rel-L2(x) {err:.2e}  "
          f"mean list {st.list_entries/max(st.targets,1):.1f} max {st.max_list} list-ovf {st.list_overflow_targets} lds-ovf {st.lds_overflow_targets} max cell {st.max_cell_occupancy}", flush=True)

"""Small helpers shared by bench.py and the slab bench (distributed.run_slab_bench): the benchmark's reference size,
the HBM peak the roofline fractions are quoted against, and the clock-ramp preheat."""
from __future__ import annotations

import sys

REF_PARTICLES = 1_747_584          # C3' 
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

_HEAT = {}


def gpu_preheat(local_rank, ms):
    """Keep the GPU busy for `ms` with work that touches none of the solver's state.  25 steps are 9 ms, and a GPU that
    idled while the host built (or restored) the scene runs its first milliseconds at a lower clock: measured here,
    `--steps 20 --warmup 5` gives 0.386 ms/step straight after the idle and 0.356 behind 30 ms of load (the same 0.356
    behind 100 ms; the kernels' own durations are identical in a trace).  What a long run sees is the second number."""
    if ms <= 0 or _HEAT.get("broken"):
        return
    import time as _t
    import torch
    try:
        dev = torch.device("cuda", local_rank)
        if dev not in _HEAT:
            _HEAT[dev] = torch.randn(2048, 2048, device=dev)
        t_end = _t.perf_counter() + ms * 1e-3
        while _t.perf_counter() < t_end:
            for _ in range(8):
                _ = _HEAT[dev] @ _HEAT[dev]
            torch.cuda.synchronize(dev)
    except Exception as e:      # noqa: BLE001 -- the load is a courtesy to the clock, never a reason to lose the measurement
        _HEAT["broken"] = True
        print(f"[bench] preheat unavailable ({type(e).__name__}: {e}); blocks run cold", file=sys.stderr, flush=True)

"""Probe (GPU box): torch.distributed with backend "cpu:gloo,cuda:nccl" and TWO ranks on ONE GPU -- the control plane (CPU tensors) over gloo while the
RCCL process group is never created (lazily initialised on the first CUDA collective, which never comes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "WORLD_SIZE" not in os.environ:
    from sph_taichi_amd.benchutil import self_launch
    sys.exit(self_launch(os.path.abspath(__file__), [], 2, 120.0, metric="probe"))
import json
import torch, torch.distributed as dist
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
torch.cuda.set_device(0)
dist.init_process_group("cpu:gloo,cuda:nccl")
t = torch.tensor([dist.get_rank() + 1.0])
dist.all_reduce(t)
out = {"get_backend": str(dist.get_backend()), "sum": float(t.item()), "world": dist.get_world_size()}
try:
    out["backend_config"] = str(dist.get_backend_config())
except Exception as e:
    out["backend_config"] = f"{type(e).__name__}"
if dist.get_rank() == 0:
    print(json.dumps(out), flush=True)
dist.destroy_process_group()

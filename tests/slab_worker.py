"""Worker for the multi-process slab tests (launched once per rank by the tests).

mode "transport": CPU only -- TorchTransport over gloo with ragged fake record buffers.
mode "slabs"    : needs a GPU -- every rank runs a SlabSolver on cuda:0 (gloo staging through the
                  host), rank 0 gathers positions by pid and writes them for the parent to compare.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from sph_taichi_amd.distributed import TorchTransport, SlabSolver, RECORD_BYTES
    if mode == "transport":
        tr = TorchTransport("cpu")
        ok = True
        cnt = lambda it: ((3 + rank + it) % 5 if rank > 0 else 0, (7 * rank + 2 * it) % 6 if rank < world - 1 else 0)
        for it in range(5):
            nL, nR = (3 + rank + it) % 5, (7 * rank + 2 * it) % 6          # ragged, sometimes zero
            if it >= 2:         # from the third round on the counts were announced one round ahead
                assert tr._pending is not None
            mk = lambda n, tag: torch.full((max(n, 1) * RECORD_BYTES,), tag, dtype=torch.uint8)
            alloc = lambda from_left, n: torch.zeros(n * RECORD_BYTES, dtype=torch.uint8)
            rL, mL, rR, mR = tr.exchange(mk(nL, 10 + rank) if rank > 0 else None, nL if rank > 0 else 0,
                                         mk(nR, 100 + rank) if rank < world - 1 else None, nR if rank < world - 1 else 0,
                                         alloc)
            if rank > 0:        # my left neighbour sent me its right range
                exp = (7 * (rank - 1) + 2 * it) % 6
                ok &= mL == exp and (exp == 0 or bool((rL[: exp * RECORD_BYTES] == 100 + rank - 1).all()))
            if rank < world - 1:
                exp = (3 + rank + 1 + it) % 5
                ok &= mR == exp and (exp == 0 or bool((rR[: exp * RECORD_BYTES] == 10 + rank + 1).all()))
            if it >= 1 and it < 4:
                tr.start_counts(*cnt(it + 1))
        t = torch.arange(16, dtype=torch.float64) * (rank + 1)      # the bodies' 16 shape-matching sums
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t, torch.arange(16, dtype=torch.float64) * (world * (world + 1) // 2)))
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            open(out, "w").write("ok" if int(flag.item()) == 1 else "fail")
    else:
        import scenes
        import json
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        recut = int(sys.argv[8]) if len(sys.argv) > 8 else 0
        cuts = json.loads(sys.argv[9]) if len(sys.argv) > 9 else None
        s = SlabSolver(sd, rank, world, device=0, recut_every=recut, cuts=cuts)
        s.attach(TorchTransport(torch.device("cuda", 0)))
        s.initialize()
        s.step(steps)
        o = s.owned(("pid", "x", "v", "density"))
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v for k, v in o.items()})
        if rank == 0:
            pid = np.concatenate([g["pid"] for g in gathered])
            np.savez(out, cuts=np.asarray(s.cuts), recuts=s.stats.get("recuts", 0), pid=pid, x=np.concatenate([g["x"] for g in gathered]),
                     v=np.concatenate([g["v"] for g in gathered]),
                     density=np.concatenate([g["density"] for g in gathered]))
        s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

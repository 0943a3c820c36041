"""Worker for the multi-process slab tests (launched once per rank by the tests).

mode "transport": CPU only -- TorchTransport over gloo with ragged fake record buffers.
mode "slabs"    : needs a GPU -- every rank runs a SlabSolver on cuda:0 (gloo staging through the
                  host), rank 0 gathers positions by pid and writes them for the parent to compare.
mode "native"   : needs a GPU -- every rank (process) runs a SlabSolver on cuda:0 over NativeTransport, i.e. the exchange
                  behind the C ABI (csrc/sph_comm.hip), with SPH_RCCL_LIB pointing at tests/fake_rccl/libfake_rccl.so (RCCL
                  itself refuses two ranks on one GPU); gloo only hands the unique id around and gathers the result.
                  argv[8] = JSON options: recut, cuts, check_every, delay_rank / delay_ms (that rank sleeps before it
                  announces and before it exchanges: no ordering assumption may turn a slow rank into a deadlock).
mode "rccl1"    : needs a GPU -- ONE rank on backend "nccl" (= RCCL on ROCm), device tensors end to end:
                  (1) the transport protocol with the rank as its own left neighbour (count announcement, ragged
                  record payloads, fixed-size swap: RCCL send/recv to self), (2) the 16-sum all-reduce, (3) a
                  SlabSolver with shape-matched bodies and re-cut events stepping over TorchTransport (its
                  all-reduces run through RCCL), positions written for the parent to compare with the single domain.
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
    if mode in ("rccl1", "native1"):
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from sph_taichi_amd.distributed import TorchTransport, NativeTransport, SlabSolver, RECORD_BYTES
    if mode == "negotiate":
        # CPU: the transport negotiation must fall back on EVERY rank when ONE rank cannot open the library (argv[6] = that
        # rank) -- nobody may be left alone in a collective
        from sph_taichi_amd import _lib
        from sph_taichi_amd.distributed import negotiate_native_transport
        if rank == int(sys.argv[6]):
            os.environ["SPH_RCCL_LIB"] = "/nonexistent/librccl_missing.so"
        class _PS:
            pass
        ps = _PS()
        ps._lib = _lib.load()          # (builds under a file lock if the library is not there yet)
        tr, why = negotiate_native_transport(ps, "cpu", create_timeout_s=20.0)
        flag = torch.tensor([1 if (tr is None and "stage 0" in why) else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            open(out, "w").write("ok" if int(flag.item()) == 1 else f"fail: {why}")
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "native_absent":
        # the peer that never shows up (test_native_exchange_reports_a_missing_peer_instead_of_hanging): it takes part in
        # the unique id's broadcast and leaves before creating its communicator
        t = torch.zeros(128, dtype=torch.uint8)
        dist.broadcast(t, 0)
        sys.stdout.flush()
        os._exit(0)
    if mode == "native1":
        # The exchange behind the C ABI (csrc/sph_comm.hip) on the hardware there is: ONE rank, its own left neighbour.
        import json
        dev = torch.device("cuda", 0)
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        s = SlabSolver(sd, 0, 1, device=0, recut_every=2, check_every=3)
        tr = NativeTransport(s.ps, dev, loopback=True)
        ok = True
        for it in range(5):                  # ragged payloads, sometimes empty; counts announced one round ahead
            n = (3 + 2 * it) % 5
            payload = (torch.arange(max(n, 1) * RECORD_BYTES, device=dev) % 251 + it).to(torch.uint8)
            torch.cuda.synchronize()
            alloc = lambda from_left, m: torch.zeros(m * RECORD_BYTES, dtype=torch.uint8, device=dev)
            rL, mL, rR, mR = tr.exchange(payload, n, None, 0, alloc)
            s.ps.sync()
            ok &= mL == n and mR == 0 and (n == 0 or bool(torch.equal(rL[: n * RECORD_BYTES], payload[: n * RECORD_BYTES])))
            if it < 4:
                tr.start_counts((3 + 2 * (it + 1)) % 5, 0)
        a = torch.arange(4096, device=dev, dtype=torch.int32).view(torch.uint8)     # DFSPH ghost-velocity refresh path
        b = torch.zeros_like(a)
        torch.cuda.synchronize()
        tr.swap(a, None, b, None)
        s.ps.sync()
        ok &= bool(torch.equal(a, b))
        t = torch.arange(16, dtype=torch.float64, device=dev) * 3.0               # the bodies' 16 shape-matching sums
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t.cpu(), torch.arange(16, dtype=torch.float64) * 3.0))
        ti = torch.tensor([41], dtype=torch.int64, device=dev)                      # the conservation guard's count
        ok &= int(tr.all_reduce_sum(ti).item()) == 41
        ms, nx = tr.halo_time()
        ok &= nx == 5 and ms >= 0.0
        tr.close()
        tr2 = NativeTransport(s.ps, dev)          # a second communicator: the solver's own (world 1, no neighbours)
        s.attach(tr2)
        s.initialize()
        s.step(steps)
        ok &= s.stats.get("recuts", 0) == 0
        o = s.owned(("pid", "x", "v", "density"))
        np.savez(out, ok=np.int32(1 if ok else 0), pid=o["pid"], x=o["x"], v=o["v"], density=o["density"],
                 backend=np.array("native-rccl"))
        tr2.close()
        s.close()
    elif mode == "native":
        import json
        import time
        assert os.environ.get("SPH_RCCL_LIB"), "mode native runs over the stand-in library (SPH_RCCL_LIB)"
        dev = torch.device("cuda", 0)
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        opt = json.loads(sys.argv[8]) if len(sys.argv) > 8 else {}
        s = SlabSolver(sd, rank, world, device=0, recut_every=int(opt.get("recut", 0)), cuts=opt.get("cuts"),
                       check_every=int(opt.get("check_every", 64)))
        tr = NativeTransport(s.ps, dev)                   # unique id: rank 0's, broadcast over gloo (CPU tensor)
        ok = True
        # the transport alone first: ragged payloads (sometimes empty) with the counts announced one round ahead, the
        # fixed-size swap, both all-reduces -- between PROCESSES
        cnt = lambda it: ((3 + rank + it) % 5 if rank > 0 else 0, (7 * rank + 2 * it) % 6 if rank < world - 1 else 0)
        mk = lambda n, tag: torch.full((max(n, 1) * RECORD_BYTES,), tag, dtype=torch.uint8, device=dev)
        alloc = lambda from_left, n: torch.zeros(n * RECORD_BYTES, dtype=torch.uint8, device=dev)
        for it in range(5):
            nL, nR = cnt(it)
            bufL, bufR = mk(nL, 10 + rank), mk(nR, 100 + rank)
            torch.cuda.synchronize()
            rL, mL, rR, mR = tr.exchange(bufL if rank > 0 else None, nL, bufR if rank < world - 1 else None, nR, alloc)
            s.ps.sync()
            if rank > 0:        # my left neighbour sent me its right range
                exp = (7 * (rank - 1) + 2 * it) % 6
                ok &= mL == exp and (exp == 0 or bool((rL[: exp * RECORD_BYTES] == 100 + rank - 1).all()))
            if rank < world - 1:
                exp = (3 + rank + 1 + it) % 5
                ok &= mR == exp and (exp == 0 or bool((rR[: exp * RECORD_BYTES] == 10 + rank + 1).all()))
            if it < 4:
                tr.start_counts(*cnt(it + 1))
        big = 3 * 65536 + 40                                # several ring slots per message (FAKE_RCCL_SLOT_BYTES is small in the tests)
        a = (torch.arange(big, device=dev, dtype=torch.int32) * (rank + 1)).view(torch.uint8)
        bL, bR = torch.zeros_like(a), torch.zeros_like(a)
        torch.cuda.synchronize()
        tr.swap(a if rank > 0 else None, a if rank < world - 1 else None, bL if rank > 0 else None, bR if rank < world - 1 else None)
        s.ps.sync()
        want = lambda r: (torch.arange(big, device=dev, dtype=torch.int32) * (r + 1)).view(torch.uint8)
        if rank > 0:
            ok &= bool(torch.equal(bL, want(rank - 1)))
        if rank < world - 1:
            ok &= bool(torch.equal(bR, want(rank + 1)))
        t = torch.arange(16, dtype=torch.float64, device=dev) * (rank + 1)
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t.cpu(), torch.arange(16, dtype=torch.float64) * (world * (world + 1) // 2)))
        ti = torch.tensor([41 + rank], dtype=torch.int64, device=dev)
        ok &= int(tr.all_reduce_sum(ti).item()) == 41 * world + world * (world - 1) // 2
        tr.close()
        # a second communicator for the solver, through the staged, collective negotiation the bench uses
        from sph_taichi_amd.distributed import negotiate_native_transport
        tr2, why = negotiate_native_transport(s.ps, dev, create_timeout_s=60.0)
        assert tr2 is not None, f"negotiation fell back: {why}"
        if int(opt.get("delay_rank", -1)) == rank:
            # a deliberately slow rank: it announces late and reaches every exchange late
            d_s = float(opt.get("delay_ms", 20)) * 1e-3
            a0, e0 = tr2.start_counts, tr2.exchange
            def slow_counts(*a_, **k_):
                time.sleep(d_s)
                return a0(*a_, **k_)
            def slow_exchange(*a_, **k_):
                time.sleep(d_s)
                return e0(*a_, **k_)
            tr2.start_counts, tr2.exchange = slow_counts, slow_exchange
        s.attach(tr2)
        s.initialize()
        s.step(steps)
        s.ps.sync()
        halo_ms, n_ex = tr2.halo_time()
        o = s.owned(("pid", "x", "v", "density"))
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v for k, v in o.items()})
        if rank == 0:
            np.savez(out, ok=np.int32(int(flag.item())), cuts=np.asarray(s.cuts), recuts=s.stats.get("recuts", 0),
                     exchanges=n_ex, sent=s.stats["sent"], iterations=np.asarray(getattr(s, "dfsph_iterations", (0, 0))),
                     pid=np.concatenate([g["pid"] for g in gathered]), x=np.concatenate([g["x"] for g in gathered]),
                     v=np.concatenate([g["v"] for g in gathered]), density=np.concatenate([g["density"] for g in gathered]))
        tr2.close()
        s.close()
    elif mode == "rccl1":
        import json
        dev = torch.device("cuda", 0)
        assert dist.get_backend() == "nccl"
        tr = TorchTransport(dev, loopback=True)
        ok = True
        for it in range(5):                  # ragged payloads, sometimes empty; counts announced one round ahead
            n = (3 + 2 * it) % 5
            payload = (torch.arange(max(n, 1) * RECORD_BYTES, device=dev) % 251 + it).to(torch.uint8)
            alloc = lambda from_left, m: torch.zeros(m * RECORD_BYTES, dtype=torch.uint8, device=dev)
            rL, mL, rR, mR = tr.exchange(payload, n, None, 0, alloc)
            ok &= mL == n and mR == 0 and (n == 0 or bool(torch.equal(rL[: n * RECORD_BYTES], payload[: n * RECORD_BYTES])))
            if it < 4:
                tr.start_counts((3 + 2 * (it + 1)) % 5, 0)
        a = torch.arange(4096, device=dev, dtype=torch.int32).view(torch.uint8)     # DFSPH ghost-velocity refresh path
        b = torch.zeros_like(a)
        tr.swap(a, None, b, None)
        ok &= bool(torch.equal(a, b))
        t = torch.arange(16, dtype=torch.float64, device=dev) * 3.0               # the bodies' 16 shape-matching sums
        tr.all_reduce_sum(t)
        ok &= bool(torch.equal(t.cpu(), torch.arange(16, dtype=torch.float64) * 3.0))
        sd = json.load(open(sys.argv[6]))
        steps = int(sys.argv[7])
        s = SlabSolver(sd, 0, 1, device=0, recut_every=2, check_every=3)
        s.attach(TorchTransport(dev))
        s.initialize()
        s.step(steps)
        ok &= s.stats.get("recuts", 0) == 0          # one slab: every re-cut event runs its all-reduce and moves nothing
        o = s.owned(("pid", "x", "v", "density"))
        np.savez(out, ok=np.int32(1 if ok else 0), pid=o["pid"], x=o["x"], v=o["v"], density=o["density"],
                 backend=np.array(dist.get_backend()))
        s.close()
    elif mode == "transport":
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

"""`bench.py --gpus N` must not be lost to plumbing (VERDICT r04 "next" #2): the launcher-free spawn (benchutil.self_launch),
the per-rank wall-clock watchdog (benchutil.Watchdog) and the collectively agreed stages of the supplementary c4 object
(distributed.collective_stage) -- exercised here on CPU with gloo and stand-in rank scripts; the GPU versions are in
tests/test_distributed.py (test_bench_*)."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text("import os, sys, json, time\nsys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(body))
    return str(p)


def _launch(tmp_path, body, world=2, total_s=60.0, grace_s=2.0, argv=()):
    drv = tmp_path / "drv.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from sph_taichi_amd.benchutil import self_launch
        sys.exit(self_launch({_script(tmp_path, body)!r}, {list(argv)!r}, {world}, {total_s}, metric="m", grace_s={grace_s}))
    """))
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, str(drv)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr.decode(), time.monotonic() - t0


def test_self_launch_spawns_the_ranks_and_relays_rank0s_line(tmp_path):
    rc, lines, err, _ = _launch(tmp_path, """
        import torch, torch.distributed as dist
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == os.environ["RANK"]
        dist.init_process_group("gloo")
        t = torch.tensor([dist.get_rank() + 1.0]); dist.all_reduce(t)
        print("noise on stdout from rank", dist.get_rank())
        if dist.get_rank() == 0:
            print(json.dumps({"value": float(t.item()), "n_gpus": dist.get_world_size(), "argv": sys.argv[1:]}), flush=True)
        dist.destroy_process_group()
    """, world=3, argv=("--gpus", "3"))
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"value": 6.0, "n_gpus": 3, "argv": ["--gpus", "3"]}


def test_self_launch_ends_the_job_when_a_rank_dies(tmp_path):
    """Rank 1 exits with code 3 while rank 0 'sits in a collective' (sleeps): the supervisor stops rank 0 after the grace
    period, returns rc != 0 and prints ONE line naming the rank, the exit codes and each rank's last stage."""
    rc, lines, err, took = _launch(tmp_path, """
        from sph_taichi_amd.benchutil import Watchdog
        r = int(os.environ["RANK"])
        wd = Watchdog(r, 2, total_s=120.0, metric="m")
        wd.stage("negotiating")
        if r == 1:
            sys.exit(3)
        time.sleep(100)
    """, grace_s=1.5)
    assert rc == 3 and took < 60
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["error"] == "launch" and "rank 1 exited with code 3" in d["detail"]
    assert d["stages"]["0"] == "negotiating" and d["exit_codes"][1] == 3 and d["exit_codes"][0] != 0


def test_watchdog_turns_a_hang_into_a_line_naming_the_stage(tmp_path):
    rc, lines, err, took = _launch(tmp_path, """
        from sph_taichi_amd.benchutil import Watchdog
        r = int(os.environ["RANK"])
        wd = Watchdog(r, 2, total_s=1.5, metric="m", take_sigterm=True)
        wd.stage("tiled: timed steps")
        time.sleep(100)            # both ranks hang
    """)
    assert rc == 124 and took < 60, err[-1500:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["error"] == "watchdog" and d["stage"] == "tiled: timed steps" and d["value"] is None and d["n_gpus"] == 2


def test_watchdog_keeps_a_finished_line_when_the_supplementary_object_hangs(tmp_path):
    rc, lines, err, took = _launch(tmp_path, """
        from sph_taichi_amd.benchutil import Watchdog
        r = int(os.environ["RANK"])
        wd = Watchdog(r, 2, total_s=120.0, metric="m")
        line = {"metric": "m", "value": 123.0, "c4_dambreak": None}
        wd.keep(line, "c4_dambreak")
        wd.stage("c4_dambreak: 2000 settling steps", budget_s=1.0)
        time.sleep(100)
    """)
    assert rc == 0 and took < 60, err[-1500:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 123.0 and d["c4_dambreak"]["error"] == "watchdog"
    assert d["c4_dambreak"]["stage"] == "c4_dambreak: 2000 settling steps"


def test_a_rank_stopped_by_the_launcher_still_names_its_stage(tmp_path):
    """torch.distributed.run ends the surviving ranks with SIGTERM when one rank fails; rank 0, stuck inside a C call, must
    still leave a line: the signal is taken through a wake-up socket by a thread of the watchdog, not by a Python handler."""
    body = _script(tmp_path, """
        import ctypes
        from sph_taichi_amd.benchutil import Watchdog
        wd = Watchdog(0, 2, total_s=120.0, metric="m", take_sigterm=True)
        wd.stage("tiled: initialize (first exchange + sort)")
        print("ready", file=sys.stderr, flush=True)
        while True:                        # the main thread is inside C (a signal that lands on it only makes the call
            ctypes.CDLL(None).sleep(100)   # return early: it goes straight back in, as a collective would be retried)
    """)
    p = subprocess.Popen([sys.executable, body], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert b"ready" in p.stderr.readline()
    p.terminate()
    out, _ = p.communicate(timeout=30)
    d = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][0])
    assert p.returncode == 128 + 15
    assert d["error"] == "terminated" and d["stage"] == "tiled: initialize (first exchange + sort)"


def test_a_rank_that_raises_says_where_and_takes_the_job_down(tmp_path):
    """Watchdog.fail (what bench.py calls from its except branch): rank 0 raises in a named stage while rank 1 waits for it.
    Rank 0 prints {"error": "exception", "stage": ...} and leaves with rc 1 at once; the supervisor stops rank 1."""
    rc, lines, err, took = _launch(tmp_path, """
        from sph_taichi_amd.benchutil import Watchdog
        r = int(os.environ["RANK"])
        wd = Watchdog(r, 2, total_s=120.0, metric="m")
        wd.stage("tiled: transport negotiation")
        if r == 0:
            try:
                raise RuntimeError("librccl said no")
            except BaseException as e:
                wd.fail(f"{type(e).__name__}: {e}")
        time.sleep(100)
    """, grace_s=1.5)
    assert rc == 1 and took < 60, (rc, err[-1500:])
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["error"] == "exception" and d["stage"] == "tiled: transport negotiation"
    assert "librccl said no" in d["detail"] and d["rank"] == 0


def test_collective_stage_fails_on_every_rank_alike(tmp_path):
    """an exception on ONE rank of the c4 object used to be caught by that rank alone while its peers
    waited in the next collective.  collective_stage: every rank raises CollectiveStageError naming stage and rank."""
    rc, lines, err, _ = _launch(tmp_path, """
        import torch, torch.distributed as dist
        from sph_taichi_amd.distributed import collective_stage, CollectiveStageError
        dist.init_process_group("gloo")
        r = dist.get_rank()
        dev = torch.device("cpu")
        assert collective_stage("a", lambda: r * 10, dev) == r * 10

        def boom():
            if r == 1:
                raise MemoryError("out of HBM")
            return "fine"
        try:
            collective_stage("build the slab contexts", boom, dev)
            got = "no error"
        except CollectiveStageError as e:
            got = str(e)
        t = torch.tensor([1.0]); dist.all_reduce(t)          # the ranks are still in step
        outs = [None] * 3
        dist.all_gather_object(outs, got)
        if r == 0:
            print(json.dumps({"got": outs, "sum": float(t.item())}), flush=True)
        dist.destroy_process_group()
    """, world=3)
    assert rc == 0, err[-2000:]
    d = json.loads(lines[0])
    assert d["sum"] == 3.0
    assert all("stage 'build the slab contexts' failed on rank 1" in g for g in d["got"]), d
    assert "MemoryError" in d["got"][1] and "MemoryError" not in d["got"][0]


def test_bench_refuses_a_world_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])      # never die without a line
    assert line["value"] is None and line["error"] == "invocation" and "WORLD_SIZE=3" in line["detail"]


def test_bench_ignores_an_inherited_world_size_without_a_launcher():
    """WORLD_SIZE alone (no RANK) is an outer environment's leftover, not a launcher: `python bench.py` must not try to join
    a job of that size -- it goes on as one rank and fails for the real reason here (no GPU), not for the mismatch."""
    env = dict(os.environ, WORLD_SIZE="4")
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert b"WORLD_SIZE=4" not in p.stderr and b"WORLD_SIZE=4" not in p.stdout
    assert b"needs a GPU" in p.stderr or p.returncode == 0

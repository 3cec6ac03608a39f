"""bench.py --gpus N without a launcher around it: it must start its own N ranks, report the number of ranks that
really joined, and refuse — loudly, non-zero — rather than shrink to one rank.  The launch path is exercised here without GPUs through
--launcher-selftest (same re-exec under torch.distributed.run, same rendezvous on 127.0.0.1, ranks joined over gloo on the CPU)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_spawns_its_own_ranks():
    r = run(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launcher_selftest"] and line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert sorted(w["rank"] for w in line["ranks"]) == [0, 1]


def test_gpus_n_refuses_when_the_devices_are_not_there():
    """On a box with fewer than 2 GPUs (this one has none; the 1-GPU test box has one) `--gpus 2` fails with a one-line reason instead of
    printing an n_gpus = 1 line."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have >= 2:
        import pytest
        pytest.skip("this box can really run two ranks")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-vae"])
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_world_size_mismatch_is_refused():
    r = run(["--gpus", "2", "--launcher-selftest"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_rank_that_never_joins_is_named_and_the_run_fails():
    """8-GPU preflight: the first multi-GPU run of bench.py will be unattended.  A rank that never enters a collective must turn
    into a non-zero exit whose stderr names it — not a silent hang until the driver's own limit.  Rank 1 of 2 sleeps in front of the join; rank 0's
    watchdog fires after --rank-timeout, reads every rank's last stage and names rank 1."""
    r = run(["--gpus", "2", "--launcher-selftest", "--selftest-hang-rank", "1", "--rank-timeout", "4"], timeout=180)
    assert r.returncode != 0
    assert "made no progress" in r.stderr and "rank 1" in r.stderr.split("furthest behind:")[-1]
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
    # every rank left its heartbeat lines
    assert "bench.py[rank 0/2" in r.stderr and "bench.py[rank 1/2" in r.stderr

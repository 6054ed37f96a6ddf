"""`bench.py --gpus N` must run N ranks -- started by bench.py itself when no launcher set WORLD_SIZE -- and, for N > 1,
default to BASELINE.json configs[4] (one Llama-2-70B model sharded TP = N, a sum all-reduce inside the step).  Exercised here
on CPU with `--dry-run` (gloo): rendezvous on 127.0.0.1, the partitioning arithmetic, one all-reduce over all ranks, and the
JSON line.  (the flag used to be dead -- `python bench.py --gpus 8` gave one process and n_gpus = 1.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=240,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line from rank 0, got %d:\n%s" % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_on_configs4():
    out = _run(["--gpus", "2", "--dry-run"])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert out["scaling"] == "strong" and "Llama-2-70B" in out["metric"] and "TP=2" in out["config"]["workload"]
    assert "configs[4]" in out["config"]["workload"] and out["config"]["batch"] == 128
    assert out["tensor_parallel"]["ranks_in_all_reduce"] == 2
    # Megatron split of Llama-2-70B at TP = 2: column-parallel qkv / gate_up, row-parallel o / down
    assert out["config"]["rank_shard_shapes_N_K"] == {"qkv": [(32 + 2 * 4) * 128, 8192], "o": [8192, 32 * 128],
                                                       "gate_up": [28672, 8192], "down": [8192, 14336]}


def test_replicas_mode_and_single_rank_line():
    out = _run(["--gpus", "2", "--dry-run", "--replicas"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "Llama-3-8B" in out["metric"]
    assert out["config"]["parallelism"].startswith("replicas x2")
    one = _run(["--dry-run"])
    assert one["n_gpus"] == 1 and "configs[1]" in one["config"]["workload"] and one["config"]["batch"] == 16


def test_under_a_launcher_the_flag_is_not_respawned():
    """The driver's form: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` (WORLD_SIZE set by the
    launcher): bench.py must not start a second level of ranks."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2

"""`python bench.py --gpus 2` with no external launcher, rehearsed on a single-GPU box: bench.py starts the two rank
processes itself (before touching the GPU); with RVA_SHARE_GPU=1 both use device 0 and the id exchange runs over gloo
with host staging instead of RCCL.  Checks that the sharded flow runs end to end and that the JSON contract holds for
N > 1 in both scaling modes; global-id consistency of the scheme itself is covered by
test_tracker_sharded_ids_match_single_process and the gloo CPU test."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RVA_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--model", "n",
                        "--no-cpu-baseline", "--no-extras", *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0 only)"
    return json.loads(lines[0])


def test_bench_self_spawns_two_ranks_weak():
    j = _run(["--streams", "4"])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["streams_per_gpu"] == 4 and "sharded 4 per GPU over 2 GPUs" in j["config"]["workload"]


def test_bench_self_spawns_two_ranks_strong():
    j = _run(["--scaling", "strong", "--total-streams", "8"])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["streams_per_gpu"] == 4
    assert "strong scaling" in j["config"]["workload"]

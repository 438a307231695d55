"""Rehearsal of `bench.py --gpus 2` on a single-GPU box: two ranks share device 0 (RVA_SHARE_GPU=1, gloo
with host staging instead of RCCL).  Checks that the sharded flow runs end to end and that the JSON
contract holds for N > 1; global-id consistency of the scheme itself is covered by
test_tracker_sharded_ids_match_single_process and the gloo CPU test."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_bench_two_ranks_on_one_gpu():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RVA_SHARE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                                       "--streams", "4", "--model", "n", "--no-cpu-baseline"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "weak" and j["value"] > 0
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints

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
    m = j["multi_gpu"]                                             # the record shows by itself that two ranks ran
    assert m["ranks_seen"] == 2 and m["backend"] == "gloo" and m["shared_gpu_rehearsal"] is True
    assert m["per_rank_frames_per_s"]["min"] > 0 and m["id_exchanges_per_rank"] >= 6 and m["id_exchange_us_per_tick"]["mean"] > 0


def test_bench_self_spawns_two_ranks_strong():
    j = _run(["--scaling", "strong", "--total-streams", "8"])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["streams_per_gpu"] == 4
    assert "strong scaling" in j["config"]["workload"]


@pytest.mark.parametrize("depth", [2, 3])
def test_config3_yolov8m_sharded_ticks_give_the_oracles_tables_and_global_ids(tmp_path, depth):
    """BASELINE configs[3] in miniature: YOLOv8m, 8 x 1080p streams sharded 4 + 4 over two rank processes (device 0 shared,
    gloo), 12 ticks through ``PipelinedTicks`` with the per-tick all-gather of new-track counts.  Every rank dumps the head
    tensors its detector produced and the tables its host received (tests/sharded_worker.py); here ALL 8 streams go
    through the oracle's post-process and ONE oracle tracker (the reference's single shared ``IouTracker``,
    tracker.py:47, pipeline.py:452,502) in canonical order -- tick-major, stream-minor -- and every rank's tables must be
    identical: ids out of the one global counter, age, hits, float64 boxes.  Both with two tick chains (the sharded default) and
    with three (``IdSync.buf`` shared by all slots, k4_assign_ids and the snapshot slot k mod 3 behind the graph replay)."""
    import socket

    import numpy as np

    from oracle import oracle as orc
    T, TOTAL, WORLD = 12, 8, 2
    per = TOTAL // WORLD
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    procs = []
    for r in range(WORLD):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RVA_SHARE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "sharded_worker.py"), "--out", str(tmp_path), "--model", "m",
                                       "--total", str(TOTAL), "--ticks", str(T), "--depth", str(depth)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=1100)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-3000:] for o in outs]
    for r in range(WORLD):
        done = (tmp_path / f"rank{r}.done").read_text()
        assert "captured=True" in done and "backend=gloo" in done and f"world={WORLD}" in done and f"depth={depth}" in done, done
    otr = orc.Tracker(TOTAL, 30, 0.5, 1)
    rows = new_ids = 0
    seen = set()
    for k in range(T):
        recs = [np.load(tmp_path / f"rank{r}_tick{k:03d}.npz") for r in range(WORLD)]
        for g in range(TOTAL):                                     # canonical order: stream-minor
            rec, s = recs[g // per], g % per
            r = orc.postprocess(rec["head"][s].astype(np.float32), 0.25, 0.45, None, (1920, 1080))
            m = r["conf"].astype(np.float64) >= 0.25               # filter_detections
            want = otr.update(g, r["boxes"][m].astype(np.float64), r["conf"][m].astype(np.float64), r["cls"][m].astype(np.int64))
            got = dict(n=int(rec[f"n{s}"]), **{f: rec[f"{f}{s}"] for f in ("id", "cls", "age", "hits", "conf", "boxes")})
            assert orc.table_of(got) == orc.table_of(want), (k, g)
            rows += want["n"]
            fresh = set(int(i) for i in want["id"]) - seen
            new_ids += len(fresh)
            seen |= fresh
    # something was tracked on both ranks and the ids interleave across them (one counter, not one per rank)
    assert rows > 10 * T and new_ids == len(seen) and max(seen) == len(seen)

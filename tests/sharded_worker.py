"""One rank of a sharded run, started by tests/test_gpu_multirank.py (RANK / WORLD_SIZE / MASTER_* in the environment;
RVA_SHARE_GPU=1: every rank on device 0, id exchange over gloo).  Runs ``--ticks`` ticks of its share of ``--total``
1080p streams through ``PipelinedTicks`` (two ticks in flight, captured tails, ``IdSync`` all-gather per tick) and writes,
per tick, the head tensor its detector produced and the track tables the host received: the parent replays the head
tensors of ALL ranks through the oracle in canonical order and requires the same tables, global ids included."""
from __future__ import annotations

import argparse
import copy
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--model", default="m")
    ap.add_argument("--total", type=int, default=8)
    ap.add_argument("--ticks", type=int, default=12)
    ap.add_argument("--target", type=int, default=60)
    ap.add_argument("--depth", type=int, default=2)
    args = ap.parse_args()
    from realtime_video_analytics_32streams_amd import dist as rdist
    from realtime_video_analytics_32streams_amd import ops
    from realtime_video_analytics_32streams_amd.config import DetectorConfig, StreamConfig, TrackerConfig
    from realtime_video_analytics_32streams_amd.detector import HipYoloDetector
    from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
    from realtime_video_analytics_32streams_amd.pipeline import PipelinedTicks, TickPipeline
    from realtime_video_analytics_32streams_amd.tracker import IouTracker
    from realtime_video_analytics_32streams_amd.video_stream import SyntheticNv12Stream
    from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net, calibrate_detection_density

    rank, world, local = rdist.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    S = args.total // world
    first = rank * S
    mk = lambda g: StreamConfig(name=f"cam{g:03d}", url="synthetic://1920x1080", warmup_seconds=0.0)      # noqa: E731
    streams = [mk(first + i) for i in range(S)]
    sources = [SyntheticNv12Stream(s, index=first + i, n_unique=3, device=dev) for i, s in enumerate(streams)]
    for s in sources:
        s.open_sync()
    # every rank calibrates the seeded weights on the SAME frames (the first frame of every stream of the job), through the
    # deterministic fused plan: all ranks hold the same network, as they would with a real checkpoint
    cal = [SyntheticNv12Stream(mk(g), index=g, n_unique=1, device=dev) for g in range(args.total)]
    for s in cal:
        s.open_sync()
    dcfg = DetectorConfig(model_path=f"yolov8{args.model}.pt", backend="hip", model_type="yolov8", half=True,
                          confidence_threshold=0.25, iou_threshold=0.45, warmup=False)
    det = HipYoloDetector(dcfg, net=copy.deepcopy(build_detector_net(args.model, seed=0)), device=local)
    with torch.inference_mode():
        sample, _ = ops.preprocess_nv12([s._ring[0] for s in cal], (640, 640), half=True)
        probs = FusedYoloV8(det.net, sample.shape[0], device=dev, autotune=False)(sample.contiguous())[:, 4:, :].float()
        probs = probs.clamp(2.0 ** -20, 1.0 - 2.0 ** -11)
        calibrate_detection_density(det.net, None, 0.25, args.target, class_logits=torch.log(probs / (1.0 - probs)))
    det.invalidate_engine()
    tcfg = TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1)
    trk = IouTracker(tcfg, max_streams=S, capacity=1024, device=local)
    pipe = TickPipeline(streams, det, trk, sources=sources, id_sync=rdist.IdSync(S, dev) if world > 1 else None,
                        first_global_index=first, n_global_streams=args.total)
    runner = PipelinedTicks(pipe, depth=args.depth, use_graph=True)
    out = Path(args.out)

    def check(k):
        _, tables = runner.collect()
        par = k % runner.nslots                                     # tick chains: every slot has its own plan, head tensor 0 of it
        key = (S, 640, 640) if (runner.net_streams < 2 or par == 0) else (S, 640, 640, par)
        plan = det._plans[key]
        head = (plan._outs[0] if runner.net_streams >= 2 else plan._outs[k & 1]).cpu().numpy()      # tick k's head tensor (fp16)
        rec = {"head": head}
        for s, t in enumerate(tables):
            n = int(t["n"])
            rec[f"n{s}"] = np.int64(n)
            for f in ("id", "cls", "age", "hits", "conf", "boxes"):
                rec[f"{f}{s}"] = np.asarray(t[f])[:n]
        np.savez(out / f"rank{rank}_tick{k:03d}.npz", **rec)

    done = 0
    for k in range(args.ticks):                                    # `depth` ticks in flight; a tick's head tensor is intact until its
        if k - done == runner.depth:                               # slot's next network runs, i.e. until `depth` further ticks are submitted
            check(done); done += 1
        runner.submit()
    while done < args.ticks:
        check(done); done += 1
    torch.cuda.synchronize()
    (out / f"rank{rank}.done").write_text(f"captured={runner._captured} net_streams={runner.net_streams} depth={runner.depth} "
                                          f"backend={torch.distributed.get_backend() if world > 1 else 'none'} world={world}\n")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    main()

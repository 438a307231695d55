"""Experiment: consecutive ticks' forward passes on N streams (each tick whole, 32 frames; no dependence between ticks) against one
stream: does the tail of tick k (20x20 layers, detect branches) overlap the head of tick k+1 (stem, 160x160 / 80x80 layers)?
usage: python tools/two_streams.py [max streams, default 4]      (plans run their detect branches in line, as in PipelinedTicks)"""
import os, sys, time; sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = build_detector_net("s").half().cuda()
plans = [FusedYoloV8(net, 32) for _ in range(N)]
for p in plans: p.concurrent_heads = False
xs = [torch.rand((32, 3, 640, 640), device="cuda").half() for _ in range(N)]
streams = [torch.cuda.Stream() for _ in range(N)]
def run(n_streams, n=90):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        j = i % n_streams
        with torch.cuda.stream(streams[j]): plans[j](xs[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for m in range(1, N + 1): run(m, 12)
for rep in range(2):
    for m in range(1, N + 1):
        print(m, "stream(s):", round(run(m), 4), "ms per forward", flush=True)

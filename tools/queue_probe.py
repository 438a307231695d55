"""Experiment: which torch streams share a hardware queue?  Three forward passes in flight on stream triples taken at different
positions of torch's stream pool (GPU_MAX_HW_QUEUES=8 as bench.py sets it), and a cheap pairwise probe (chains of tiny kernels)."""
import os, sys, time; sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
net = build_detector_net("s").half().cuda()
plans = [FusedYoloV8(net, 32) for _ in range(3)]
for p in plans: p.concurrent_heads = False
xs = [torch.rand((32, 3, 640, 640), device="cuda").half() for _ in range(3)]
pool = [torch.cuda.Stream() for _ in range(20)]
print("stream ids", [hex(s.cuda_stream)[-6:] for s in pool], flush=True)
def run(streams, n=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        j = i % len(streams)
        with torch.cuda.stream(streams[j]): plans[j](xs[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(pool[:3], 12)
for tri in ((0, 1, 2), (0, 8, 16), (0, 4, 8), (1, 9, 17), (3, 4, 5), (0, 1, 9), (5, 6, 7), (0, 2, 4)):
    print(tri, round(run([pool[i] for i in tri]), 4), "ms per forward", flush=True)
one = [torch.zeros(1, device="cuda") for _ in range(2)]
def pair(a, b, n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(a): one[0].add_(1)
        with torch.cuda.stream(b): one[1].add_(1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
pair(pool[0], pool[1])
print("pair probe (ms for 300 + 300 tiny kernels):", {(i, j): round(pair(pool[i], pool[j]), 3) for i, j in ((0, 1), (0, 8), (0, 16), (0, 4), (1, 9), (0, 0))})

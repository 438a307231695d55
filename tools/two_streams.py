"""Experiment: consecutive ticks' forward passes on TWO streams (each tick whole, 32 frames; no dependence between ticks) against one
stream: does the tail of tick k (20x20 layers, detect branches) overlap the head of tick k+1 (stem, 160x160 / 80x80 layers)?"""
import sys, time; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
net = build_detector_net("s").half().cuda()
e1, e2 = FusedYoloV8(net, 32), FusedYoloV8(net, 32)
x = torch.rand((32, 3, 640, 640), device="cuda").half()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(two, n=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if two and (i & 1):
            with torch.cuda.stream(s2): e2(x)
        else:
            with torch.cuda.stream(s1): e1(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(2): run(False, 10); run(True, 10)
for label, two in (("one stream", False), ("two streams", True), ("one stream", False), ("two streams", True)):
    print(label, round(run(two), 4), "ms per forward", flush=True)

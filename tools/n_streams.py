"""Experiment: forward passes of consecutive ticks round-robin on N streams (N plans, detect branches in line)."""
import sys, time, os; sys.path.insert(0, ".")
os.environ.setdefault("RVA_SERIAL_HEADS", "1"); os.environ.setdefault("RVA_TUNE_IN_PLAN", "0")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
net = build_detector_net("s").half().cuda()
N = 4
engs = [FusedYoloV8(net, 32, autotune=(i == 0)) for i in range(N)]
for e in engs[1:]: e.copy_tuning(engs[0])
x = torch.rand((32, 3, 640, 640), device="cuda").half()
ss = [torch.cuda.Stream() for _ in range(N)]
def run(n_streams, n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        j = i % n_streams
        with torch.cuda.stream(ss[j]): engs[j](x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for k in (1, 2, 3, 4): run(k, 12)
for k in (1, 2, 3, 4, 1, 2, 3, 4):
    print(k, "streams:", round(run(k), 4), "ms per forward", flush=True)

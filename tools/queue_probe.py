"""Experiment: which torch streams can run forward passes side by side?  All pairs of the first NS streams of the process (two
plans, passes alternating between the pair), then the best triples / quadruples found greedily from the pair matrix.
usage: python tools/queue_probe.py [NS=8]      (GPU_MAX_HW_QUEUES=8 as bench.py sets it)"""
import itertools, os, sys, time; sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = build_detector_net("s").half().cuda()
plans = [FusedYoloV8(net, 32) for _ in range(4)]
for p in plans: p.concurrent_heads = False
xs = [torch.rand((32, 3, 640, 640), device="cuda").half() for _ in range(4)]
pool = [torch.cuda.Stream() for _ in range(NS)]
def run(streams, n=24):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        j = i % len(streams)
        with torch.cuda.stream(streams[j]): plans[j](xs[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for i in range(NS): run([pool[i]], 4)                     # first use in index order
print("one stream:", [round(run([pool[i]], 12), 3) for i in range(NS)], flush=True)
pm = {}
for i, j in itertools.combinations(range(NS), 2):
    pm[(i, j)] = run([pool[i], pool[j]])
print("pairs (ms per forward):")
for i in range(NS):
    print("  ", i, " ".join(f"{pm[(min(i, j), max(i, j))]:.3f}" if i != j else "  -  " for j in range(NS)), flush=True)
for k in (3, 4):
    res = sorted((run([pool[i] for i in c], 12 * k), c) for c in itertools.combinations(range(min(NS, 6)), k))
    print(k, "streams, best:", [(round(t, 3), c) for t, c in res[:4]], "worst:", [(round(t, 3), c) for t, c in res[-3:]], flush=True)

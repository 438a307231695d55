"""Experiment: one 32-frame plan against 2 x 16 / 4 x 8 frame plans on concurrent streams (same weights): do the
launch gaps and the partial last waves of the small layers fill up when independent half-batches overlap?"""
import sys, time; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
net = build_detector_net(sys.argv[1] if len(sys.argv) > 1 else "s").half().cuda()
x = torch.rand((32, 3, 640, 640), device="cuda").half()
def bench(parts, reps=60):
    n = 32 // parts
    engs = [FusedYoloV8(net, n) for _ in range(parts)]
    xs = [x[i * n:(i + 1) * n].contiguous() for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    ev = [torch.cuda.Event() for _ in range(parts)]
    def run():
        main = torch.cuda.current_stream()
        start = torch.cuda.Event(); start.record(main)
        for e, xi, s, d in zip(engs, xs, streams, ev):
            if parts == 1:
                e(xi)
            else:
                s.wait_event(start)
                with torch.cuda.stream(s):
                    e(xi); d.record(s)
        if parts > 1:
            for d in ev: main.wait_event(d)
    for _ in range(5): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for parts in (1, 2, 4, 1, 2):
    print(parts, "x", 32 // parts, "frames:", round(bench(parts), 3), "ms per 32 frames", flush=True)

import sys, time; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
net = build_detector_net("s").half().cuda()
t0 = time.time(); eng = FusedYoloV8(net, 32); print("build+tune s", round(time.time() - t0, 1))
print("refined:", getattr(eng, "refined", None))
x = torch.rand((32, 3, 640, 640), device="cuda").half()
for _ in range(5): eng(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(50): eng(x)
e1.record(); torch.cuda.synchronize()
print("forward ms", e0.elapsed_time(e1) / 50)

import sys; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = FusedYoloV8(build_detector_net(sys.argv[2] if len(sys.argv) > 2 else "s").half().cuda(), B)
names = {0: "auto", 1: "gather<64,64>", 2: "gather<64,32>", 3: "gather<128,64>", 4: "gather<128,32>", 5: "res<64,64>", 6: "res<64,32>", 7: "res<128,64>", 8: "res<128,32>", 9: "row<64,64>", 10: "row<64,32>", 11: "row<128,64>", 12: "row<128,32>", 13: "g64<64,64>", 14: "g64<64,32>", 15: "g64<128,64>", 16: "g64<128,32>", 17: "row2<64,64>", 18: "row2<64,32>", 19: "row2<128,64>", 20: "row2<128,32>"}
tot = 0
seen = {}
for launch, state, desc in eng._tunable:
    seen[desc] = seen.get(desc, 0) + 1
t = {d: (v, us) for d, v, us in eng.tuning}
for d, n in seen.items():
    v, us = t[d]
    tot += us * n
    print(f"{d:28s} x{n}  {names[v]:14s} {us:8.1f} us")
print("sum conv us per forward:", round(tot, 1))
x = torch.rand((B, 3, 640, 640), device="cuda").half()
for _ in range(3): eng(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): eng(x)
e1.record(); torch.cuda.synchronize()
print("forward ms:", e0.elapsed_time(e1) / 10)

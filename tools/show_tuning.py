import sys; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = FusedYoloV8(build_detector_net(sys.argv[2] if len(sys.argv) > 2 else "s").half().cuda(), B)
names = {0: "auto", 1: "gather<64,64>", 2: "gather<64,32>", 3: "gather<128,64>", 4: "gather<128,32>", 5: "res<64,64>", 6: "res<64,32>", 7: "res<128,64>", 8: "res<128,32>", 9: "row<64,64>", 10: "row<64,32>", 11: "row<128,64>", 12: "row<128,32>", 13: "g64<64,64>", 14: "g64<64,32>", 15: "g64<128,64>", 16: "g64<128,32>", 17: "row2<64,64>", 18: "row2<64,32>", 19: "row2<128,64>", 20: "row2<128,32>", 21: "big<256,128>", 22: "big<128,128>", 23: "big<256,64>", 24: "big<128,64>", 25: "big2<192,128>", 26: "big2<128,128>", 27: "big2<256,64>", 28: "big2<128,64>", 29: "big2<224,128>", 30: "big2<160,128>", 31: "big2<384,64>", 32: "big2<320,64>", 33: "gb<256,128>", 34: "gb<128,128>", 35: "gb<256,64>", 36: "gb<128,64>", 37: "gb2<128,128>", 38: "gb2<256,64>", 39: "gb2<192,128>", 40: "gb32<256,64>", 41: "gb32<128,64>", 42: "gb32_2<256,64>", 43: "s2patch", 44: "s1patch4", 45: "s1patch8", 46: "p64x4db", 47: "p64x8", 48: "p64x4", 49: "p64x8db", 50: "p64x4r2db", 51: "p64x8r2", 52: "run<256,64>", 53: "run<128,64>", 54: "run<384,64>", 55: "run<192,128>", 56: "run<256,128>", 57: "run<128,128>", 58: "run<224,128>", 59: "run<160,128>", 60: "run<320,64>", 61: "p64h32x8", 62: "p64h32x4", 63: "p64h32x4r2", 64: "gb2<256,256>", 65: "gb2<256,256>w128x64", 66: "p64 two sets", 67: "chunk<128,64>", 68: "chunk<64,64>", 69: "chunk<256,64>", 70: "chunk<64,96>", 71: "chunk<128,96>", 72: "chunk<192,64>", 73: "chunk<256,96>", 74: "gbs<128,128>x2", 75: "gbs<128,64>x2", 76: "gbs<128,64>x3", 77: "gbs<64,64>x4", 78: "gbs<64,128>x3", 79: "gbs<64,64>x2", 80: "runp<256,64>", 81: "runp<256,128>", 82: "runp<224,128>", 83: "runp<320,64>", 84: "runp<192,128>", 85: "runp<128,64>", 86: "s2run<256,128>", 87: "s2run<256,64>", 88: "s2run<128,128>", 89: "s2run<128,64>"}
tot = 0
seen = {}
for launch, state, desc in eng._tunable:
    seen[desc] = seen.get(desc, 0) + 1
t = {d: (v, us) for d, v, us in eng.tuning}
import re
gap_tot = 0
rows = []
for d, n in seen.items():
    v, us = t[d]
    tot += us * n
    d0 = d
    d = re.sub(r"^head[12]:", "", d)
    mu = re.match(r"up(\d+)\+(\d+)->(\d+) k1s1 (\d+)x(\d+)", d)
    if mu:      # fused upsample + concat source: the low-res part is read at quarter size
        cl, cs, cout, h, w = map(int, mu.groups())
        cin, k, st, ho, wo = cl + cs, 1, 1, h, w
        by = 2 * (B * (h // 2) * (w // 2) * cl + B * h * w * cs + B * h * w * cout)
    else:
        cin, cout, k, st, h, w = map(int, re.match(r"(\d+)->(\d+) k(\d)s(\d) (\d+)x(\d+)", d).groups())
        ho, wo = ((h - 1) // st + 1, (w - 1) // st + 1) if k == 3 else (h // st, w // st)
        by = 2 * (B * h * w * cin + B * ho * wo * cout)
    fl = 2 * B * ho * wo * cout * cin * k * k
    t_hbm, t_mfma = by / 6.0e6, fl / 2.5e9          # us at 6 TB/s achievable HBM, 2.5 PFLOP/s dense fp16
    roof = max(t_hbm, t_mfma)
    gap = (us - roof) * n
    gap_tot += gap
    rows.append((gap, f"{d:24s} x{n} {names[v]:14s} {us:7.1f} us  hbm {t_hbm:6.1f}  mfma {t_mfma:6.1f}  x{us/roof:4.1f} of roof  gap*n {gap:7.1f} us"))
for _, r in sorted(rows, reverse=True):
    print(r)
for r in getattr(eng, "refined", None) or []:
    print("in-plan refinement:", r[0], names.get(r[1], r[1]), "->", names.get(r[2], r[2]), "forward", r[3], "->", r[4], "us")
print("sum of gaps to the per-layer roofline:", round(gap_tot, 1), "us")
print("sum conv us per forward:", round(tot, 1))
x = torch.rand((B, 3, 640, 640), device="cuda").half()
for _ in range(3): eng(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): eng(x)
e1.record(); torch.cuda.synchronize()
print("forward ms:", e0.elapsed_time(e1) / 10)

#!/usr/bin/env python3
"""Micro-benchmark: every conv layer shape of the fused YOLOv8 plan, timed alone (HIP events around
`reps` back-to-back launches), vs the torch/MIOpen conv+bias+SiLU chain on the same shape."""
import ctypes as C, sys, os
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from realtime_video_analytics_32streams_amd import _native as N, ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def main():
    B = int(os.environ.get("B", 32))
    with_torch = os.environ.get("TORCH", "0") == "1"
    shapes = [  # Cin, Cout, k, s, H
        (32, 64, 3, 2, 320), (64, 64, 1, 1, 160), (32, 32, 3, 1, 160), (96, 64, 1, 1, 160), (64, 128, 3, 2, 160),
        (128, 128, 1, 1, 80), (64, 64, 3, 1, 80), (256, 128, 1, 1, 80), (128, 256, 3, 2, 80), (256, 256, 1, 1, 40),
        (128, 128, 3, 1, 40), (512, 256, 1, 1, 40), (256, 512, 3, 2, 40), (512, 512, 1, 1, 20), (256, 256, 3, 1, 20),
        (768, 512, 1, 1, 20), (1024, 512, 1, 1, 20), (768, 256, 1, 1, 40), (384, 256, 1, 1, 40), (384, 128, 1, 1, 80),
        (192, 128, 1, 1, 80), (128, 128, 3, 2, 80), (256, 256, 3, 2, 40), (128, 64, 3, 1, 80), (128, 128, 3, 1, 80),
        (128, 80, 1, 1, 80), (256, 64, 3, 1, 40), (256, 128, 3, 1, 40), (512, 64, 3, 1, 20), (512, 128, 3, 1, 20),
        (64, 64, 3, 1, 40), (64, 64, 3, 1, 20), (128, 128, 3, 1, 20),
    ]
    L, ctx = N.lib(), ops.context()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = 0
    for cin, cout, k, st, H in shapes:
        x = torch.randn((B, H, H, cin), device="cuda").half()
        Ho = (H - 1) // st + 1 if k == 3 else H // st
        out = torch.empty((B, Ho, Ho, cout), device="cuda", dtype=torch.float16)
        cpad, cinp = L.rva_conv_cout_pad(cout), (cin + 31) // 32 * 32
        w = torch.randn((cpad, k * k, cinp), device="cuda").half() * 0.05
        b = torch.zeros(cpad, device="cuda")
        fn = lambda: ctx.check(L.rva_conv2d_nhwc_f16(ctx.handle, C.c_void_p(x.data_ptr()), cin, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                               C.c_void_p(out.data_ptr()), cout, None, 0, B, H, H, cin, cout, k, st, 1, s))
        us = timeit(fn)
        fl = 2 * B * Ho * Ho * cout * cin * k * k
        by = 2 * (B * H * H * cin + B * Ho * Ho * cout)
        line = f"{cin:4d}->{cout:3d} k{k}s{st} {H:3d}  {us:8.1f} us {fl/us/1e6:7.1f} TF/s {by/us/1e3:6.0f} GB/s"
        if with_torch:
            conv = torch.nn.Conv2d(cin, cout, k, st, k // 2).cuda().half().to(memory_format=torch.channels_last)
            xt = x.permute(0, 3, 1, 2)
            with torch.inference_mode():
                ut = timeit(lambda: F.silu(conv(xt), inplace=True))
            line += f"   | torch {ut:8.1f} us  x{ut/us:4.1f}"
        print(line); tot += us
    print("sum us", round(tot, 1))

main()

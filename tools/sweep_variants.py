"""Time every kernel variant of rva_conv2d_nhwc_f16_v on a few layer shapes (tuning aid)."""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops
L, ctx = N.lib(), ops.context()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(256, 256, 3, 1, 20), (128, 128, 3, 1, 40), (64, 64, 3, 1, 80), (768, 512, 1, 1, 20)]
NV = int(L.rva_conv_num_variants()) if hasattr(L, "rva_conv_num_variants") else 20
B = 32
for cin, cout, k, st, H in shapes:
    x = torch.randn((B, H, H, cin), device="cuda").half()
    Ho = (H - 1) // st + 1 if k == 3 else H // st
    out = torch.empty((B, Ho, Ho, cout), device="cuda", dtype=torch.float16)
    cpad, cinp = L.rva_conv_cout_pad(cout), (cin + 31) // 32 * 32
    w = torch.randn((cpad, k * k, cinp), device="cuda").half() * 0.05
    b = torch.zeros(cpad, device="cuda")
    res = []
    for v in range(1, NV + 1):
        fn = lambda: L.rva_conv2d_nhwc_f16_v(ctx.handle, C.c_void_p(x.data_ptr()), cin, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                             C.c_void_p(out.data_ptr()), cout, None, 0, B, H, H, cin, cout, k, st, 1, v, s)
        if fn() != 0:
            res.append(f"{v}:--")
            continue
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(f"{v}:{e0.elapsed_time(e1) / 20 * 1e3:.1f}")
    print(f"{cin}->{cout} k{k}s{st} {H}: " + " ".join(res), flush=True)

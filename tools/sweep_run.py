"""Interleaved A/B of conv variants on one layer shape in ONE process (rule 24): N rounds of back-to-back launches per variant.
usage: sweep_run.py cin cout k stride H B v1 v2 ..."""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
from realtime_video_analytics_32streams_amd import _native as N, ops
cin, cout, k, st, H, B = [int(v) for v in sys.argv[1:7]]
variants = [int(v) for v in sys.argv[7:]]
L, ctx = N.lib(), ops.context()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn((B, H, H, cin), device="cuda").half()
Ho = (H - 1) // st + 1 if k == 3 else H // st
out = torch.empty((B, Ho, Ho, cout), device="cuda", dtype=torch.float16)
cpad, cinp = L.rva_conv_cout_pad(cout), (cin + 31) // 32 * 32
w = torch.randn((cpad, k * k, cinp), device="cuda").half() * 0.05
b = torch.zeros(cpad, device="cuda")
def run(v):
    return L.rva_conv2d_nhwc_f16_v(ctx.handle, C.c_void_p(x.data_ptr()), cin, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                   C.c_void_p(out.data_ptr()), cout, None, 0, B, H, H, cin, cout, k, st, 1, v, s)
ok = [v for v in variants if run(v) == 0]
torch.cuda.synchronize()
ts = {v: [] for v in ok}
for r in range(12):
    for v in ok:
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10): run(v)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"{cin}->{cout} k{k}s{st} {H}x{H} B{B}: " + "  ".join(f"v{v}: med {np.median(t):.1f} min {np.min(t):.1f} us" for v, t in ts.items()))

"""Stem + first downsampling convolution at the benchmark size (32 x 3 x 640 x 640): one fused launch against the two-launch path."""
import ctypes as C, sys; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops
L, ctx = N.lib(), ops.context()
B, H, W = 32, 640, 640
x = torch.rand((B, 3, H, W), device="cuda").half()
sw = (torch.randn((64, 32), device="cuda") * 0.2).half(); sb = torch.zeros(64, device="cuda")
wp = (torch.randn((64, 9, 32), device="cuda") * 0.05).half(); bp = torch.zeros(64, device="cuda")
x0 = torch.empty((B, 320, 320, 32), device="cuda", dtype=torch.float16)
out = torch.empty((B, 160, 160, 64), device="cuda", dtype=torch.float16)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def fused(): assert L.rva_stem2_f16(ctx.handle, p(x), p(sw), p(sb), p(wp), p(bp), p(out), 64, B, H, W, s) == 0
def two():
    assert L.rva_stem_conv_f16(ctx.handle, p(x), p(sw), p(sb), p(x0), 32, B, H, W, 32, s) == 0
    assert L.rva_conv2d_nhwc_f16_v(ctx.handle, p(x0), 32, p(wp), p(bp), p(out), 64, None, 0, B, 320, 320, 32, 64, 3, 2, 1, 43, s) == 0
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, fn in [(n_, f_) for n_, f_ in (("fused", fused), ("two launches", two), ("fused", fused), ("two launches", two)) if only in ("", n_)]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)

"""The fused 32-channel bottleneck launch alone at the bench shape (32 x 160 x 160, slices of the 96-channel concat buffer) next to the
two patch launches it replaces (variant 45 twice): us per launch / pair."""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops
L, ctx = N.lib(), ops.context()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, W = 32, 160, 160
cat = (torch.randn((B, H, W, 96), device="cuda") * 0.5).half()
tmp = torch.zeros((B, H, W, 32), device="cuda", dtype=torch.float16)
cpad = L.rva_conv_cout_pad(32)
w1 = (torch.randn((cpad, 9, 32), device="cuda") * 0.05).half(); w2 = (torch.randn((cpad, 9, 32), device="cuda") * 0.05).half()
b1 = torch.zeros(cpad, device="cuda"); b2 = torch.zeros(cpad, device="cuda")
x, y = cat.data_ptr() + 64, cat.data_ptr() + 128
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def fused():
    ctx.check(L.rva_c2f_pair32_f16(ctx.handle, C.c_void_p(x), 96, P(w1), P(b1), P(w2), P(b2), C.c_void_p(y), 96, B, H, W, s), "pair")
def two():
    ctx.check(L.rva_conv2d_nhwc_f16_v(ctx.handle, C.c_void_p(x), 96, P(w1), P(b1), P(tmp), 32, None, 0, B, H, W, 32, 32, 3, 1, 1, 45, s), "c1")
    ctx.check(L.rva_conv2d_nhwc_f16_v(ctx.handle, P(tmp), 32, P(w2), P(b2), C.c_void_p(y), 96, C.c_void_p(x), 96, B, H, W, 32, 32, 3, 1, 1, 45, s), "c2")
for name, fn in (("fused", fused), ("two launches", two), ("fused", fused), ("two launches", two)):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 10:.1f} us")

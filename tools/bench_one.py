import ctypes as C, sys, os
sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops
cin, cout, k, st, H, B = [int(v) for v in sys.argv[1:7]]
variant = int(sys.argv[7]) if len(sys.argv) > 7 else 0
L, ctx = N.lib(), ops.context()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn((B, H, H, cin), device="cuda").half()
Ho = (H - 1) // st + 1 if k == 3 else H // st
out = torch.empty((B, Ho, Ho, cout), device="cuda", dtype=torch.float16)
cpad, cinp = L.rva_conv_cout_pad(cout), (cin + 31) // 32 * 32
w = torch.randn((cpad, k * k, cinp), device="cuda").half() * 0.05
b = torch.zeros(cpad, device="cuda")
fn = lambda: ctx.check(L.rva_conv2d_nhwc_f16_v(ctx.handle, C.c_void_p(x.data_ptr()), cin, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                       C.c_void_p(out.data_ptr()), cout, None, 0, B, H, H, cin, cout, k, st, 1, variant, s))
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print(f"variant={variant} dbg={os.environ.get('RVA_CONV_DBG','0')} {cin}->{cout} k{k}s{st} {H}: {e0.elapsed_time(e1)/20*1e3:.1f} us")

import ctypes as C, sys, os
sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops
L, ctx = N.lib(), ops.context()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.rand((32, 3, 640, 640), device="cuda").half()
out = torch.empty((32, 320, 320, 32), device="cuda", dtype=torch.float16)
w = torch.zeros((64, 32), device="cuda", dtype=torch.float16); b = torch.zeros(64, device="cuda")
fn = lambda: ctx.check(L.rva_stem_conv_f16(ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), 32, 32, 640, 640, 32, s))
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print("dbg", os.environ.get("RVA_STEM_DBG", "0"), "stem us", e0.elapsed_time(e1) / 20 * 1e3)

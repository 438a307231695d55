"""SPPF pooling launch alone at the bench shape (32 x 20 x 20 x 256 inside the 1024-wide concat buffer): us per launch.
RVA_SPPF_G1=1 selects the one-group-per-block kernel."""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops
L, ctx = N.lib(), ops.context()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (bb, cc) in ((32, 256), (4, 288)):
    xs = torch.randn((bb, 20, 20, 4 * cc)).half().cuda()
    call = lambda: ctx.check(L.rva_sppf_pool3_nhwc_f16(ctx.handle, C.c_void_p(xs.data_ptr()), 4 * cc, C.c_void_p(xs.data_ptr() + 2 * cc),  # noqa: E731
                                                       C.c_void_p(xs.data_ptr() + 4 * cc), C.c_void_p(xs.data_ptr() + 6 * cc), 4 * cc, bb, 20, 20, cc, s), "sppf")
    for _ in range(10):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"batch {bb} C {cc}: {e0.elapsed_time(e1) * 5:.2f} us per launch")

"""Per-phase s_memtime stamps of the fused stem + first downsampling convolution kernel (k_stem2), tuning aid.
Builds a private library with -DRVA_ROW_STAMPS into tools/_dbg/ (git-ignored).  usage: python tools/stem2_stamps.py [--build-only]"""
import ctypes as C, subprocess, sys
from pathlib import Path
sys.path.insert(0, ".")
from realtime_video_analytics_32streams_amd import _native as N
DBG = Path("tools/_dbg/librva_stamps.so")
if not DBG.exists() or "--build-only" in sys.argv:
    DBG.parent.mkdir(exist_ok=True)
    subprocess.run(["hipcc", *N.HIPCC_FLAGS, "-DRVA_ROW_STAMPS", f"-I{N.ROOT / 'include'}", "-o", str(DBG), *[str(N.CSRC / s) for s in N.SOURCES], "-ldl"], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import numpy as np, torch
L = C.CDLL(str(DBG))
ctx = C.c_void_p()
assert L.rva_create(0, C.byref(ctx)) == 0
B, H, W = 32, 640, 640
x = torch.rand((B, 3, H, W), device="cuda").half()
sw = (torch.randn((64, 32), device="cuda") * 0.2).half(); sb = torch.zeros(64, device="cuda")
wp = (torch.randn((64, 9, 32), device="cuda") * 0.05).half(); bp = torch.zeros(64, device="cuda")
out = torch.empty((B, 160, 160, 64), device="cuda", dtype=torch.float16)
p = lambda t: C.c_void_p(t.data_ptr())
L.rva_stem2_f16.argtypes = [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]
for _ in range(3):
    assert L.rva_stem2_f16(ctx, p(x), p(sw), p(sb), p(wp), p(bp), p(out), 64, B, H, W, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
torch.cuda.synchronize()
host = np.zeros((8, 256), dtype=np.uint64)
assert L.rva_dbg_read_stamps(host.ctypes.data_as(C.c_void_p)) == 0
for slot in range(8):
    t = host[slot].astype(np.int64)
    n = int((t > 0).sum()) // 4
    if n < 3:
        continue
    body = t[:4 * n].reshape(n, 4)[1:-1]                      # drop the fill / drain iterations
    work, tail, bar = body[:, 1] - body[:, 0], body[:, 2] - body[:, 1], body[:, 3] - body[:, 2]
    grp = "S (stem)   " if slot % 2 == 0 else "C (conv 2) "
    print(f"block {slot // 2} group {grp}: iterations {n}  per iteration {int((body[-1, 0] - body[0, 0]) / (len(body) - 1))} cycles;  work {int(work.mean())}  input->LDS {int(tail.mean())}  barrier wait {int(bar.mean())}")

"""Per-phase s_memtime stamps of k_conv3_patch (tuning aid; private -DRVA_ROW_STAMPS build, see tools/row_stamps.py).
usage: python tools/patch_stamps.py [--build-only] cin cout H stride variant"""
import ctypes as C, subprocess, sys
from pathlib import Path
sys.path.insert(0, ".")
from realtime_video_analytics_32streams_amd import _native as N
DBG = Path("tools/_dbg/librva_stamps.so")
if "--build-only" in sys.argv or not DBG.exists():
    DBG.parent.mkdir(exist_ok=True)
    subprocess.run(["hipcc", *N.HIPCC_FLAGS, "-DRVA_ROW_STAMPS", f"-I{N.ROOT / 'include'}", "-o", str(DBG), *[str(N.CSRC / s) for s in N.SOURCES], "-ldl"], check=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
import numpy as np, torch
cin, cout, H, stride, variant = [int(v) for v in sys.argv[1:6]]
B = 32
L = C.CDLL(str(DBG)); ctx = C.c_void_p(); assert L.rva_create(0, C.byref(ctx)) == 0
x = torch.randn((B, H, H, cin), device="cuda").half()
Ho = (H - 1) // stride + 1
out = torch.empty((B, Ho, Ho, cout), device="cuda", dtype=torch.float16)
w = torch.randn(((cout + 63) // 64 * 64, 9, cin), device="cuda").half() * 0.05
b = torch.zeros((cout + 63) // 64 * 64, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.rva_conv2d_nhwc_f16_v.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 8 + [C.c_int, C.c_void_p]
for _ in range(3):
    assert L.rva_conv2d_nhwc_f16_v(ctx, x.data_ptr(), cin, w.data_ptr(), b.data_ptr(), out.data_ptr(), cout, None, 0, B, H, H, cin, cout, 3, stride, 1, variant, s) == 0
torch.cuda.synchronize()
host = np.zeros((8, 256), dtype=np.uint64)
assert L.rva_dbg_read_stamps(host.ctypes.data_as(C.c_void_p)) == 0
names = ["issue(1buf)+wait", "barrier", "issue next (2buf)", "masks+reads+MFMA", "barrier", "silu+stage", "barrier", "stores", "loop end"]
for slot in (0, 3, 7):
    t = host[slot].astype(np.int64)
    n = int((t > 0).sum()) // 9
    if n < 2: continue
    tt = t[:9 * n].reshape(n, 9)
    d = np.diff(tt, axis=1)
    nxt = tt[1:, 0] - tt[:-1, 8]
    print(f"block {slot}: tiles {n}, ticks per tile {(tt[1:,0]-tt[:-1,0]).mean():.0f}: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.mean(0))) + f", {names[8]} {nxt.mean():.0f}")

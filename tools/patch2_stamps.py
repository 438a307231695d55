"""Per-phase s_memtime stamps of k_conv3_patch2 (two wave sets), tuning aid.  usage: python tools/patch2_stamps.py [--build-only] H"""
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
H = int(sys.argv[1]); cin = 64; B = 32; VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 66
L = C.CDLL(str(DBG)); ctx = C.c_void_p(); assert L.rva_create(0, C.byref(ctx)) == 0
x = torch.randn((B, H, H, cin), device="cuda").half()
out = torch.empty((B, H, H, cin), device="cuda", dtype=torch.float16)
w = torch.randn((64, 9, cin), device="cuda").half() * 0.05
b = torch.zeros(64, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.rva_conv2d_nhwc_f16_v.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 8 + [C.c_int, C.c_void_p]
for _ in range(3):
    assert L.rva_conv2d_nhwc_f16_v(ctx, x.data_ptr(), cin, w.data_ptr(), b.data_ptr(), out.data_ptr(), cin, None, 0, B, H, H, cin, cin, 3, 1, 1, VAR, s) == 0
torch.cuda.synchronize()
host = np.zeros((8, 256), dtype=np.uint64)
assert L.rva_dbg_read_stamps(host.ctypes.data_as(C.c_void_p)) == 0
names = ["MFMA phase", "barrier", "issue next patch", "epilogue", "vmcnt wait", "barrier"]
for slot in range(8):
    t = host[slot].astype(np.int64)
    n = int((t > 0).sum()) // 6
    if n < 1: continue
    tt = t[:6 * n].reshape(n, 6)
    d = np.diff(np.concatenate([tt.reshape(-1), [tt[-1, -1]]])).reshape(n, 6)
    print(f"block {slot // 2} set {slot % 2}: iterations {n}: " + " | ".join(", ".join(f"{nm} {v}" for nm, v in zip(names, row)) for row in d))

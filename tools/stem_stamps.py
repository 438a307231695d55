"""Per-phase s_memtime stamps of k_stem (tuning aid; private -DRVA_ROW_STAMPS build, see tools/row_stamps.py)."""
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
L = C.CDLL(str(DBG)); ctx = C.c_void_p(); assert L.rva_create(0, C.byref(ctx)) == 0
B, H = 32, 640
x = torch.rand((B, 3, H, H), device="cuda").half()
out = torch.empty((B, H // 2, H // 2, 32), device="cuda", dtype=torch.float16)
w = (torch.randn((64, 32), device="cuda") * 0.1).half(); b = torch.zeros(64, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.rva_stem_conv_f16.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
for _ in range(3):
    assert L.rva_stem_conv_f16(ctx, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), 32, B, H, H, 32, s) == 0
torch.cuda.synchronize()
host = np.zeros((8, 256), dtype=np.uint64)
assert L.rva_dbg_read_stamps(host.ctypes.data_as(C.c_void_p)) == 0
names = ["tile store+sync", "prefetch issue", "im2col+sync", "frag reads+MFMA", "sync", "silu+stage+sync", "global stores"]
for slot in (0, 3, 7):
    t = host[slot].astype(np.int64)
    n = int((t > 0).sum()) // 7
    tt = t[:7 * n].reshape(n, 7)
    d = np.diff(np.concatenate([tt, np.concatenate([tt[1:, :1], tt[-1:, -1:]])], axis=1), axis=1)[:-1]
    print(f"block {slot}: tiles {n}, cycles per tile {d.sum(1).mean():.0f}: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.mean(0))))

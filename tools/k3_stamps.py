"""Where k3_nms spends its cycles under load: builds a private librva with -DRVA_K3_STAMPS (s_memtime sums per phase, thread 0
of every block) into tools/_dbg/ and runs the post-process on the load sweep's synthetic heads.
usage: python tools/k3_stamps.py [--build-only] [D]      phases: 0 candidates+keys | 1 sort | 2 phase 1 (vs kept list) |
3 survivor compactions + stage B | 4 phase 2a (suppression matrix) | 5 phase 2b (greedy walk + outputs)"""
import ctypes as C, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
DBG = ROOT / "tools" / "_dbg" / "librva_k3stamps.so"
os.environ["RVA_LIB_PATH"] = str(DBG)
from realtime_video_analytics_32streams_amd import _native as N
if not DBG.exists() or "--build-only" in sys.argv:
    DBG.parent.mkdir(exist_ok=True)
    subprocess.run(["hipcc", *N.HIPCC_FLAGS, "-DRVA_K3_STAMPS", f"-I{N.ROOT / 'include'}", "-o", str(DBG), *[str(N.CSRC / s) for s in N.SOURCES], "-ldl"], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import numpy as np, torch
from realtime_video_analytics_32streams_amd import ops, synth
args = [a for a in sys.argv[1:] if not a.startswith("--")]
D = int(args[0]) if args else 256
S = 32
ctx = ops.context()
raw = torch.from_numpy(synth.make_head_batch([9000 + D + i for i in range(S)], layout="CA", n_obj=D)).cuda().half()
post = ops.PostBuffers.allocate(S, raw.shape[2], raw.device)
meta = [N.letterbox(1920, 1080, 640, 640)]
for _ in range(5):
    ops.postprocess(raw, 0.25, 0.45, None, meta, out=post, ctx=ctx)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 8))()
assert N.lib().rva_dbg_k3_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8)[:S, :7].astype(np.float64)
names = ["candidates+keys", "sort", "phase 1 stage A", "compactions+stage B", "phase 2a", "phase 2b", "kept list re-bin"]
print(f"D={D}: candidates/frame {float(post.ncand.float().mean()):.0f} kept/frame {float(post.counts.float().mean()):.1f}; cycles of thread 0, mean over {S} blocks")
for i, n in enumerate(names):
    print(f"  {n:18s} {st[:, i].mean():10.0f}  ({100 * st[:, i].mean() / st.sum(1).mean():5.1f} %)")
print(f"  total              {st.sum(1).mean():10.0f}")

"""K1 alone, back to back (for rocprofv3 --kernel-trace / --pmc runs): 32 x 1080p NV12 -> fp16[32,3,640,640].
usage: k1_only.py [launches] [content|full]   (content = the steady-state kernel that writes the content rows only)"""
import sys; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import ops, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
content = len(sys.argv) > 2 and sys.argv[2] == "content"
surfs = []
for s in range(32):
    y, uv = synth.make_nv12(synth.SEED_BASE + 1000 * s, 1920, 1080, 2048)
    surfs.append(ops.Nv12Surface.from_numpy(y, uv, 1920, 1080))
out = torch.empty((32, 3, 640, 640), dtype=torch.float16, device="cuda")
ops.preprocess_nv12(surfs, (640, 640), True, out=out)          # full launch first: writes the border
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(n): ops.preprocess_nv12(surfs, (640, 640), True, out=out, content_only=content)
e1.record(); torch.cuda.synchronize()
print("kernel:", "k1_ratio_content" if content else "k1_ratio", "warm back-to-back avg us:", e0.elapsed_time(e1) / n * 1e3)

"""K1 alone, back to back (for rocprofv3 --kernel-trace / --pmc runs).
usage: k1_only.py [launches] [content|full|4k|clip]
  content / full : 32 x 1080p NV12 -> fp16[32,3,640,640] (content = the steady-state kernel that writes the content rows only)
  4k             : 8 x 3840x2160 NV12 -> fp16[8,3,640,640]  (k1_ratio<6>, border included)
  clip           : 8 x 3840x2160 NV12 -> fp32[8,3,224,224]  (k1_generic: stretch + ImageNet mean/std, the temporal heads' pre-process)"""
import sys; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
uhd = mode in ("4k", "clip")
W, H, S = (3840, 2160, 8) if uhd else (1920, 1080, 32)
surfs = []
for s in range(S):
    y, uv = synth.make_nv12(synth.SEED_BASE + 1000 * s, W, H, (W + 255) // 256 * 256)
    surfs.append(ops.Nv12Surface.from_numpy(y, uv, W, H))
if mode == "clip":
    out = torch.empty((S, 3, 224, 224), dtype=torch.float32, device="cuda")
    fn = lambda: ops.preprocess_frames(surfs, (224, 224), N.NORM_IMAGENET_F32, N.LAYOUT_NCHW, torch.float32, out=out)
else:
    out = torch.empty((S, 3, 640, 640), dtype=torch.float16, device="cuda")
    ops.preprocess_nv12(surfs, (640, 640), True, out=out)          # full launch first: writes the border
    fn = lambda: ops.preprocess_nv12(surfs, (640, 640), True, out=out, content_only=mode == "content")
fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(n): fn()
e1.record(); torch.cuda.synchronize()
print("mode:", mode, "warm back-to-back avg us:", e0.elapsed_time(e1) / n * 1e3)

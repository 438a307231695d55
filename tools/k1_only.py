"""K1 alone, back to back (for rocprofv3 --kernel-trace / --pmc runs): 32 x 1080p NV12 -> fp16[32,3,640,640]."""
import sys; sys.path.insert(0, ".")
import torch
from realtime_video_analytics_32streams_amd import ops, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
surfs = []
for s in range(32):
    y, uv = synth.make_nv12(synth.SEED_BASE + 1000 * s, 1920, 1080, 2048)
    surfs.append(ops.Nv12Surface.from_numpy(y, uv, 1920, 1080))
out = torch.empty((32, 3, 640, 640), dtype=torch.float16, device="cuda")
scratch = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")   # 512 MiB sweep to push surfaces out of the 256 MiB MALL
for _ in range(3): ops.preprocess_nv12(surfs, (640, 640), True, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(n): ops.preprocess_nv12(surfs, (640, 640), True, out=out)
e1.record(); torch.cuda.synchronize()
print("warm back-to-back avg us:", e0.elapsed_time(e1) / n * 1e3)
cold = []
for _ in range(10):
    scratch.fill_(1)
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record(); ops.preprocess_nv12(surfs, (640, 640), True, out=out); b.record(); torch.cuda.synchronize()
    cold.append(a.elapsed_time(b) * 1e3)
print("cold (after 512 MiB sweep) us:", sorted(cold)[len(cold) // 2])

"""The post-process / tracker tail alone under load, back to back (for rocprofv3 --kernel-trace runs and quick timing):
32 streams, synthetic heads with D planted objects per frame -> K2 -> K3 (k3_nms [-> k3_mask -> k3_reduce]) -> K4 (k4_iou ->
k4_update) -> ids.   usage: load_tail.py [D] [ticks]"""
import sys; sys.path.insert(0, ".")
import numpy as np
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops, synth
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
S = 32
ctx = ops.context()
heads = synth.make_head_batch([9000 + D + i for i in range(S)], layout="CA", n_obj=D)
raw = torch.from_numpy(heads).cuda().half()
post = ops.PostBuffers.allocate(S, raw.shape[2], raw.device)
trk = ops.DeviceTracker(S, 30, 0.5, 1, capacity=1024, ctx=ctx)
meta = [N.letterbox(1920, 1080, 640, 640)]
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(T)]
for t in range(T + 2):
    if t >= 2: ev[t - 2][0].record()
    ops.postprocess(raw, 0.25, 0.45, None, meta, out=post, ctx=ctx)
    if t >= 2: ev[t - 2][1].record()
    trk.update_from_post(list(range(S)), post, 0.25)
    trk.assign_ids()
    if t >= 2: ev[t - 2][2].record()
torch.cuda.synchronize()
tabs = trk.read_all()
print(f"D={D}: kept/frame {float(post.counts.float().mean()):.1f} candidates/frame {float(post.ncand.float().mean()):.0f} tracks/stream "
      f"{np.mean([t['n'] for t in tabs]):.1f}  K2+K3 {np.mean([e[0].elapsed_time(e[1]) for e in ev]) * 1e3:.1f} us/tick  "
      f"K4+ids {np.mean([e[1].elapsed_time(e[2]) for e in ev]) * 1e3:.1f} us/tick")

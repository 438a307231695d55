"""K1 alone: the full-tensor kernel (k1_ratio) and the steady-state content-only kernel (k1_ratio_content), 32 x 1080p
NV12 -> fp16[32,3,640,640]; warm (back to back on the same surfaces) and cold (after a 512 MiB sweep that evicts the
256 MiB Infinity Cache: a READ sweep leaves clean lines behind, a WRITE sweep leaves dirty lines that K1's own traffic has
to push out).  Timed by the dispatch's own start/stop events (rva_profile_next_preprocess)."""
import sys; sys.path.insert(0, ".")
import numpy as np
import torch
from realtime_video_analytics_32streams_amd import _native as N, ops, synth
S = 32
ctx = ops.context()
surfs = []
for s in range(S):
    y, uv = synth.make_nv12(synth.SEED_BASE + 1000 * s, 1920, 1080, 2048)
    surfs.append(ops.Nv12Surface.from_numpy(y, uv, 1920, 1080))
out = torch.empty((S, 3, 640, 640), dtype=torch.float16, device="cuda")
scratch = torch.empty(512 << 20, dtype=torch.uint8, device="cuda").fill_(1)
ops.preprocess_nv12(surfs, (640, 640), True, out=out)
ref = out.clone()


def timed(content, sweep=None, reps=12):
    ts = []
    for _ in range(reps):
        if sweep == "read":
            scratch.view(torch.int64).sum()
        elif sweep == "write":
            scratch.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); torch.cuda.synchronize()
        N.lib().rva_profile_next_preprocess(ctx.handle, a.cuda_event, b.cuda_event)
        ops.preprocess_nv12(surfs, (640, 640), True, out=out, content_only=content)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


print("kernel            state            median_us  min_us   bytes_MB   TB/s(median)  frac_of_8TB/s")
for content, nbytes, name in ((False, 3_840_000 * S, "full k1_ratio   "), (True, 2_764_800 * S, "content-only    ")):
    for sweep, label in ((None, "warm            "), ("read", "cold, read sweep "), ("write", "cold, write sweep")):
        med, mn = timed(content, sweep)
        print(f"{name}  {label} {med:8.2f} {mn:8.2f} {nbytes / 1e6:9.2f} {nbytes / med / 1e6:10.2f} {nbytes / med / 1e6 / 8.0:10.3f}")
    assert torch.equal(out, ref), "content-only launch changed the tensor"
print("tensor identical after content-only launches: ok")

"""K7 (rva_jpeg_encode_bgr) timing on the GPU box: device time per picture (HIP events around the C call, no read-back) and the
whole ``ops.jpeg_encode_bgr`` call (device + the size / stream read-back), next to Pillow (libjpeg-turbo) on one host core.
Usage: python tools/k7_time.py [--out gpurun_out/k7_time.json]"""
import argparse
import ctypes as C
import io
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from realtime_video_analytics_32streams_amd import _native as N, ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ctx, L = ops.context(), N.lib()
    rows = []
    for (w, h) in ((1920, 1080), (640, 360)):
        bgr = synth.make_bgr(3, w, h)
        img = torch.from_numpy(bgr).cuda()
        for q in (75, 95):
            data = ops.jpeg_encode_bgr(img, q)                     # allocates scratch, warms up
            cap = int(L.rva_jpeg_max_bytes(w, h))
            out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            size = torch.zeros(1, dtype=torch.int32, device="cuda")
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            call = lambda: ctx.check(L.rva_jpeg_encode_bgr(ctx.handle, C.c_void_p(img.data_ptr()), int(img.stride(0)), w, h, q,  # noqa: E731
                                                           C.c_void_p(out.data_ptr()), cap, C.c_void_p(size.data_ptr()), s), "k7")
            for _ in range(5):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                call()
            e1.record(); torch.cuda.synchronize()
            dev_us = e0.elapsed_time(e1) * 1e3 / n
            t0 = time.perf_counter()
            for _ in range(20):
                ops.jpeg_encode_bgr(img, q)
            whole_us = (time.perf_counter() - t0) * 1e6 / 20
            row = dict(wh=[w, h], quality=q, bytes=len(data), device_us=round(dev_us, 1), whole_call_us=round(whole_us, 1),
                       input_GBps=round(w * h * 3 / dev_us / 1e3, 1))
            try:
                from PIL import Image
                im = Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]))
                t0 = time.perf_counter()
                for _ in range(10):
                    im.save(io.BytesIO(), format="JPEG", quality=q)
                row["pillow_1core_us"] = round((time.perf_counter() - t0) * 1e6 / 10, 1)
            except Exception:  # noqa: BLE001
                pass
            rows.append(row)
            print(row, flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()

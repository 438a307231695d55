"""Shared helpers for the parity tests (rebuild golden inputs, serialise tables)."""
from __future__ import annotations

import hashlib
import json

import numpy as np

from realtime_video_analytics_32streams_amd import synth


def head_for_case(case) -> np.ndarray:
    """Return the raw 2-D head exactly as the golden generator fed it to the reference."""
    if case["kind"] == "seeded":
        gen = dict(case["gen"])
        if "content" in gen:
            gen["content"] = tuple(gen["content"])
        head = synth.make_head(case["seed"], **gen)  # [A, C]
        assert synth.sha256_of(head) == case["sha"], "numpy drift: regenerated input differs from golden input"
        return np.ascontiguousarray(head.T) if case["layout"] == "CA" else head
    return np.asarray(case["data"], np.float32).reshape(case["shape"])


def table_digest(table) -> str:
    return hashlib.sha256(json.dumps(table).encode()).hexdigest()[:16]


def script_sha(script) -> str:
    sha = hashlib.sha256()
    for row in script:
        for fd in row:
            sha.update(fd.boxes.tobytes()); sha.update(fd.conf.tobytes()); sha.update(fd.cls.tobytes())
    return sha.hexdigest()


def temporal_net(case):
    """This package's network for a G6 case, rebuilt from the seed; its state dict must hash to what the REFERENCE's
    module (scripts/convert_temporal_model_to_onnx.py:34-121) held when the logits were recorded."""
    from realtime_video_analytics_32streams_amd import synth
    from realtime_video_analytics_32streams_amd.temporal import Cnn3dNet, CnnLstmNet
    kw = case["ctor"]
    if case["kind"] == "cnn_lstm":
        ctor = lambda: CnnLstmNet(kw["num_classes"], kw["hidden_size"])      # noqa: E731
    else:
        ctor = lambda: Cnn3dNet(kw["num_classes"])                           # noqa: E731
    net = synth.seeded_module(ctor, case["seed"])
    assert synth.state_sha(net.state_dict()) == case["state_sha"], "torch drift: rebuilt weights differ from the reference's"
    x = synth.seeded_clip(case["clip_shape"], case["clip_seed"])
    assert synth.sha256_of(x.numpy()) == case["clip_sha"]
    return net, x


def plan_rounded_reference(net, x, device=None):
    """The detector network in torch fp32 with the roundings of the fused plan (engine.py / csrc/rva_conv.hip) and no others:
    fp16 weights and input, fp32 accumulate + bias + SiLU, ONE rounding to fp16 per layer (``__floats2half2_rn`` of the
    epilogue), the shortcut added to the rounded activation in fp32 and rounded once more, the last 1x1 convolution of a
    detect branch rounded to fp16 logits, then the DFL expectation / box arithmetic / sigmoid in fp32 (``head_anchor``).
    Returns the UNROUNDED fp32 ``[B, 4 + nc, A]`` result (the plan rounds it to fp16 when it stores it): the comparison
    allows half an fp16 ulp for that last rounding.  What remains between the two is fp32 summation order inside a
    convolution (flips an fp16 rounding now and then) and the fast ``exp`` / ``rcp`` of the epilogues.
    ``net``: a ``yolov8.YoloV8`` (folded here); ``x``: fp16 ``[B,3,H,W]``."""
    import copy

    import torch
    import torch.nn.functional as F

    from realtime_video_analytics_32streams_amd.yolov8 import ConvBnAct

    net = copy.deepcopy(net).fuse().float()
    dev = device or x.device
    net = net.to(dev)
    r16 = lambda t: t.half().float()                                        # noqa: E731

    def cba(m, t):
        conv = m.conv if isinstance(m, ConvBnAct) else m
        y = F.conv2d(t, r16(conv.weight), conv.bias.float(), conv.stride, conv.padding)
        if isinstance(m, ConvBnAct) and m.act:
            y = F.silu(y)
        return r16(y)

    def c2f(m, t):
        y = list(cba(m.cv1, t).chunk(2, 1))
        for b in m.m:
            z = cba(b.cv2, cba(b.cv1, y[-1]))
            y.append(r16(y[-1] + z) if b.add else z)
        return cba(m.cv2, torch.cat(y, 1))

    def sppf(m, t):
        y = [cba(m.cv1, t)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], m.k, 1, m.k // 2))
        return cba(m.cv2, torch.cat(y, 1))

    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")        # noqa: E731
    with torch.inference_mode():
        t = x.to(dev).half().float()
        t = c2f(net.b2, cba(net.b1, cba(net.b0, t)))
        p3 = c2f(net.b4, cba(net.b3, t))
        p4 = c2f(net.b6, cba(net.b5, p3))
        p5 = sppf(net.b9, c2f(net.b8, cba(net.b7, p4)))
        n4 = c2f(net.h12, torch.cat((up(p5), p4), 1))
        n3 = c2f(net.h15, torch.cat((up(n4), p3), 1))
        m4 = c2f(net.h18, torch.cat((cba(net.h16, n3), n4), 1))
        m5 = c2f(net.h21, torch.cat((cba(net.h19, m4), p5), 1))
        outs = []
        for lvl, (f, stride) in enumerate(((n3, 8.0), (m4, 16.0), (m5, 32.0))):
            B, _, h, w = f.shape
            box, cls = net.detect.box[lvl], net.detect.cls[lvl]
            bl = cba(box[2], cba(box[1], cba(box[0], f))).reshape(B, 4, 16, h * w)     # fp16-rounded logits
            cl = cba(cls[2], cba(cls[1], cba(cls[0], f))).reshape(B, net.nc, h * w)
            e = torch.exp(bl - bl.max(2, keepdim=True).values)
            d = (e * torch.arange(16, device=dev, dtype=torch.float32).view(1, 1, 16, 1)).sum(2) / e.sum(2)
            ay, ax = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32) + 0.5,
                                    torch.arange(w, device=dev, dtype=torch.float32) + 0.5, indexing="ij")
            ax, ay = ax.reshape(1, -1), ay.reshape(1, -1)
            x1, y1, x2, y2 = ax - d[:, 0], ay - d[:, 1], ax + d[:, 2], ay + d[:, 3]
            xywh = torch.stack(((x1 + x2) * 0.5 * stride, (y1 + y2) * 0.5 * stride, (x2 - x1) * stride, (y2 - y1) * stride), 1)
            outs.append(torch.cat((xywh, torch.sigmoid(cl)), 1))
        return torch.cat(outs, 2)


def assert_matches_rounded_reference(got16, want32, score_tol=2e-3, box_eps=0.03):
    """``got16``: the plan's fp16 head tensor; ``want32``: :func:`plan_rounded_reference`.  Boxes: half an fp16 ulp of the
    value (the store's rounding: 0.125 px below 512, 0.25 px below 1024) + ``box_eps``; scores: ``score_tol`` absolute."""
    import torch
    got = got16.float()
    w = want32.to(got.device)
    ulp = torch.pow(2.0, torch.floor(torch.log2(w[:, :4].abs().clamp_min(2.0 ** -14))) - 10)
    berr = (got[:, :4] - w[:, :4]).abs()
    bad = berr > 0.5 * ulp + box_eps
    assert not bool(bad.any()), ("boxes", float(berr.max()), int(bad.sum()))
    serr = (got[:, 4:] - w[:, 4:]).abs()
    assert float(serr.max()) < score_tol, ("scores", float(serr.max()))
    return float(berr.max()), float(serr.max())


def fp16_error_report(tag, got16, want_fp32, matched, conf=0.25):
    """What the fp16 plan costs in the units north_star speaks in: |dscore| and |dbox| in input pixels of
    the plan's fp16 head tensor against (a) the plain fp32 module and (b) the rounding-matched reference -- max / mean / p99.9 --
    and how many (anchor, class) threshold decisions at ``conf`` flip against (a).  Printed, returned and -- on the GPU box --
    written to gpurun_out/fp16_error_<tag>.json so that the numbers land in DESIGN.md section 2, not only a bound in a test."""
    import json
    from pathlib import Path
    import torch
    got = got16.float()
    rep = {"tag": tag, "anchors": int(got.shape[0] * got.shape[2]), "classes": int(got.shape[1] - 4), "conf": conf}
    for name, ref in (("vs_fp32_module", want_fp32), ("vs_rounding_matched_reference", matched)):
        r = ref.to(got.device).float()
        ds = (got[:, 4:] - r[:, 4:]).abs().flatten()
        db = (got[:, :4] - r[:, :4]).abs().flatten()
        k_s, k_b = max(int(ds.numel() * 0.999), 1), max(int(db.numel() * 0.999), 1)
        rep[name] = {"score": {"max": float(ds.max()), "mean": float(ds.mean()), "p99_9": float(ds.kthvalue(k_s).values)},
                     "box_px": {"max": float(db.max()), "mean": float(db.mean()), "p99_9": float(db.kthvalue(k_b).values)}}
        flips = ((got[:, 4:] >= conf) != (r[:, 4:] >= conf))
        rep[name]["threshold_flips"] = {"count": int(flips.sum()), "of_decisions": int(flips.numel()),
                                        "scores_at_or_above_conf_in_reference": int((r[:, 4:] >= conf).sum())}
    print("fp16 error report:", json.dumps(rep))
    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    if out.is_dir():
        (out / f"fp16_error_{tag}.json").write_text(json.dumps(rep, indent=1))
    return rep

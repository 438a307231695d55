"""Deterministic synthetic inputs for the detect/track hot path.

Used by three parties that must see *identical* bytes:
  * ``oracle/gen_golden.py`` (runs in the build container, drives the reference's own
    functions on these inputs and records what they return),
  * the parity tests under ``tests/`` (regenerate the same inputs on the GPU box and compare
    the HIP path with the oracle / the recorded goldens),
  * ``bench.py`` (synthetic 1080p NV12 surfaces, YOLOv8 head tensors, tracker scripts).

Everything is driven by ``numpy.random.default_rng(seed)``; every golden stores a SHA-256 of
the regenerated input so that a numpy drift is detected rather than silently compared.

Shapes follow SURVEY.md section 8(d): head tensors ``[84, 8400]`` (xywh in input pixels +
80 class columns), frames 1920x1080 NV12 (Y plane + interleaved UV plane, pitch-linear).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

SEED_BASE = 20251128  # per-stream seed = SEED_BASE + 1000 * stream


def sha256_of(*arrays: np.ndarray) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------------------
# YOLO head tensors
# --------------------------------------------------------------------------------------
def make_head(
    seed: int,
    anchors: int = 8400,
    n_cls: int = 80,
    n_obj: int = 12,
    img_w: int = 640,
    img_h: int = 640,
    content: Tuple[int, int, int, int] = (0, 140, 640, 500),
    dup: Tuple[int, int] = (4, 24),
    bg_max: float = 0.02,
    class0_objects: int = 2,
    edge_objects: int = 2,
) -> np.ndarray:
    """Return ``float32[anchors, 4 + n_cls]`` rows ``(cx, cy, w, h, p0..p{n_cls-1})``.

    ``n_obj`` objects are planted, each on ``dup`` randomly chosen anchors with jittered
    boxes (so NMS has clusters to resolve).  Because the reference treats column 4 as
    "objectness" for 84-column heads a planted object of true class
    ``c >= 1`` sets ``p0`` high as well; ``class0_objects`` extra objects are pure class 0 and
    must therefore be *dropped* by a reference-faithful post-process.  ``edge_objects``
    extra objects straddle the letterbox content border to exercise clipping.
    """
    rng = np.random.default_rng(seed)
    cols = 4 + n_cls
    pred = np.empty((anchors, cols), dtype=np.float32)
    pred[:, 0] = rng.uniform(0, img_w, anchors)
    pred[:, 1] = rng.uniform(0, img_h, anchors)
    pred[:, 2] = rng.uniform(4, 200, anchors)
    pred[:, 3] = rng.uniform(4, 200, anchors)
    pred[:, 4:] = rng.uniform(0, bg_max, (anchors, n_cls))

    x0, y0, x1, y1 = content
    free = rng.permutation(anchors)
    cursor = 0

    def plant(cx, cy, w, h, cls, p0, pc):
        nonlocal cursor
        m = int(rng.integers(dup[0], dup[1] + 1))
        m = min(m, anchors - cursor)
        idx = free[cursor:cursor + m]
        cursor += m
        jit = rng.uniform(-0.06, 0.06, (m, 4))
        pred[idx, 0] = cx + jit[:, 0] * w
        pred[idx, 1] = cy + jit[:, 1] * h
        pred[idx, 2] = w * (1.0 + jit[:, 2])
        pred[idx, 3] = h * (1.0 + jit[:, 3])
        fall = rng.uniform(0.55, 1.0, m)
        if cls == 0:
            pred[idx, 4] = p0 * fall
        else:
            pred[idx, 4] = p0 * rng.uniform(0.9, 1.0, m)
            pred[idx, 4 + cls] = pc * fall

    for _ in range(n_obj):
        w = rng.uniform(20, 220)
        h = rng.uniform(20, 160)
        cx = rng.uniform(x0 + 10, x1 - 10)
        cy = rng.uniform(y0 + 10, y1 - 10)
        cls = int(rng.integers(1, n_cls))
        plant(cx, cy, w, h, cls, rng.uniform(0.6, 0.99), rng.uniform(0.5, 0.99))
    for _ in range(class0_objects):
        w = rng.uniform(20, 220)
        h = rng.uniform(20, 160)
        plant(rng.uniform(x0, x1), rng.uniform(y0, y1), w, h, 0, rng.uniform(0.8, 0.99), 0.0)
    for k in range(edge_objects):
        w = rng.uniform(60, 260)
        h = rng.uniform(60, 260)
        cx = x0 if k % 2 == 0 else x1
        cy = y0 if k % 2 == 0 else y1
        cls = int(rng.integers(1, n_cls))
        plant(cx, cy, w, h, cls, rng.uniform(0.7, 0.99), rng.uniform(0.6, 0.99))
    return pred


def make_head_batch(seeds: Sequence[int], layout: str = "CA", **kw) -> np.ndarray:
    """Stack heads: ``layout='CA'`` -> ``[B, 84, 8400]`` (ONNX export layout), ``'AC'`` ->
    ``[B, 8400, 84]``."""
    heads = [make_head(s, **kw) for s in seeds]
    arr = np.stack(heads, 0)
    if layout == "CA":
        arr = np.ascontiguousarray(arr.transpose(0, 2, 1))
    return arr


# --------------------------------------------------------------------------------------
# NV12 / BGR frames
# --------------------------------------------------------------------------------------
def make_nv12(
    seed: int,
    w: int,
    h: int,
    pitch: Optional[int] = None,
    tick: int = 0,
    n_obj: int = 12,
) -> Tuple[np.ndarray, np.ndarray]:
    """Limited-range NV12 surface: ``y uint8[h, pitch]``, ``uv uint8[h//2, pitch]``.

    Y = smooth 2-D gradient + ``n_obj`` moving filled rectangles (integer velocities,
    wrap-around) + uniform noise +-4, clipped to 16..235; UV = per-rectangle constant chroma
    over a slowly varying background, clipped to 16..240.
    """
    assert w % 2 == 0 and h % 2 == 0, "NV12 needs even dimensions"
    pitch = pitch or w
    assert pitch >= w
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    ybase = 40.0 + 150.0 * (xx / max(w - 1, 1)) * 0.6 + 150.0 * (yy / max(h - 1, 1)) * 0.4
    rect = rng.integers(0, 1 << 30, (n_obj, 8))
    yimg = ybase.copy()
    cw, ch = w // 2, h // 2
    cy_, cx_ = np.mgrid[0:ch, 0:cw]
    u = 128.0 + 30.0 * np.sin(cx_ / max(cw, 1) * 3.0)
    v = 128.0 + 30.0 * np.cos(cy_ / max(ch, 1) * 2.0)
    for r in rect:
        rw = 16 + int(r[0] % max(w // 6, 17))
        rh = 16 + int(r[1] % max(h // 5, 17))
        vx = int(r[2] % 9) - 4
        vy = int(r[3] % 7) - 3
        px = (int(r[4] % w) + vx * tick) % w
        py = (int(r[5] % h) + vy * tick) % h
        x1, y1 = min(px + rw, w), min(py + rh, h)
        yimg[py:y1, px:x1] = 30 + int(r[6] % 190)
        u[py // 2:(y1 + 1) // 2, px // 2:(x1 + 1) // 2] = 40 + int(r[7] % 180)
        v[py // 2:(y1 + 1) // 2, px // 2:(x1 + 1) // 2] = 40 + int((r[7] >> 8) % 180)
    noise = np.random.default_rng(seed * 7919 + tick).integers(-4, 5, (h, w))
    yplane = np.zeros((h, pitch), np.uint8)
    yplane[:, :w] = np.clip(np.rint(yimg) + noise, 16, 235).astype(np.uint8)
    uvplane = np.zeros((h // 2, pitch), np.uint8)
    uvplane[:, 0:w:2] = np.clip(np.rint(u), 16, 240).astype(np.uint8)
    uvplane[:, 1:w:2] = np.clip(np.rint(v), 16, 240).astype(np.uint8)
    return yplane, uvplane


def make_bgr(seed: int, w: int, h: int) -> np.ndarray:
    """Arbitrary-size ``uint8[h, w, 3]`` BGR frame (full range, noisy) for the host-frame
    entry point (``BaseDetector.predict(packet)`` with an ndarray frame)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.empty((h, w, 3), np.float64)
    img[..., 0] = 255.0 * xx / max(w - 1, 1)
    img[..., 1] = 255.0 * yy / max(h - 1, 1)
    img[..., 2] = 127.0 + 120.0 * np.sin((xx + yy) / 17.0)
    img += rng.integers(-20, 21, (h, w, 3))
    for _ in range(6):
        x0 = int(rng.integers(0, w)); y0 = int(rng.integers(0, h))
        x1 = min(w, x0 + int(rng.integers(4, max(w // 3, 5))))
        y1 = min(h, y0 + int(rng.integers(4, max(h // 3, 5))))
        img[y0:y1, x0:x1] = rng.integers(0, 256, 3)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# Tracker scripts
# --------------------------------------------------------------------------------------
@dataclass
class FrameDets:
    boxes: np.ndarray  # float64 [D, 4], values representable in float32 (as the detector emits)
    conf: np.ndarray   # float64 [D]
    cls: np.ndarray    # int64 [D]


def make_tracker_script(
    seed: int,
    n_streams: int,
    n_ticks: int,
    n_obj: int = 8,
    frame_wh: Tuple[int, int] = (1920, 1080),
    n_cls: int = 3,
    p_miss: float = 0.12,
    p_dup: float = 0.10,
    p_skip: float = 0.03,
    p_birth: float = 0.05,
    p_death: float = 0.03,
) -> List[List[FrameDets]]:
    """``script[tick][stream] -> FrameDets`` in the canonical order (tick-major, stream-minor).

    Objects random-walk with constant velocity + jitter, are missed with ``p_miss``, are
    occasionally reported twice in one frame (``p_dup``: exercises the reference tracker's
    non-exclusive matching and same-frame track reuse), streams
    occasionally skip a frame (empty update, ages every track), objects are born / die.
    Detections are ordered by descending confidence (the order NMS hands them over in).
    """
    rng = np.random.default_rng(seed)
    W, H = frame_wh
    objs: List[List[np.ndarray]] = []
    for _ in range(n_streams):
        lst = []
        for _ in range(n_obj):
            lst.append(_new_obj(rng, W, H, n_cls))
        objs.append(lst)
    script: List[List[FrameDets]] = []
    for _t in range(n_ticks):
        row: List[FrameDets] = []
        for s in range(n_streams):
            lst = objs[s]
            for o in lst:
                o[0] += o[4] + rng.normal(0, 1.5)
                o[1] += o[5] + rng.normal(0, 1.5)
            lst[:] = [o for o in lst if rng.random() >= p_death]
            if rng.random() < p_birth * max(n_obj, 1):
                lst.append(_new_obj(rng, W, H, n_cls))
            if rng.random() < p_skip:
                row.append(FrameDets(np.zeros((0, 4)), np.zeros((0,)), np.zeros((0,), np.int64)))
                continue
            b, c, k = [], [], []
            for o in lst:
                if rng.random() < p_miss:
                    continue
                reps = 2 if rng.random() < p_dup else 1
                for _r in range(reps):
                    jx, jy, jw, jh = rng.normal(0, 2.0, 4)
                    cx, cy, w, h = o[0] + jx, o[1] + jy, max(o[2] + jw, 2.0), max(o[3] + jh, 2.0)
                    box = np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], np.float32)
                    box[[0, 2]] = np.clip(box[[0, 2]], 0, W - 1)
                    box[[1, 3]] = np.clip(box[[1, 3]], 0, H - 1)
                    b.append(box)
                    c.append(np.float32(rng.uniform(0.3, 0.99)))
                    k.append(int(o[6]))
            if b:
                order = np.argsort(-np.asarray(c, np.float64), kind="stable")
                bb = np.asarray(b, np.float32)[order].astype(np.float64)
                cc = np.asarray(c, np.float32)[order].astype(np.float64)
                kk = np.asarray(k, np.int64)[order]
            else:
                bb, cc, kk = np.zeros((0, 4)), np.zeros((0,)), np.zeros((0,), np.int64)
            row.append(FrameDets(bb, cc, kk))
        script.append(row)
    return script


def _new_obj(rng, W, H, n_cls):
    return np.array([
        rng.uniform(0, W), rng.uniform(0, H), rng.uniform(30, 300), rng.uniform(30, 300),
        rng.uniform(-6, 6), rng.uniform(-4, 4), float(rng.integers(0, n_cls)),
    ])


# --------------------------------------------------------------------------------------
# seeded modules / clips (temporal-network goldens: oracle/gen_golden.py G6 and the tests that replay them)
# --------------------------------------------------------------------------------------
# reference parameter prefix -> this package's (scripts/convert_temporal_model_to_onnx.py:34-121 vs temporal.py)
TEMPORAL_KEY_MAP = (("cnn.", "stem."), ("lstm.", "rnn."), ("fc.", "head."))


def map_temporal_key(key: str, kind: str = "cnn_lstm") -> str:
    """Name of a reference state-dict entry in CnnLstmNet (``cnn -> stem``, ``lstm -> rnn``, ``fc -> head``); the 3D-CNN
    uses the reference's own names."""
    if kind == "cnn_lstm":
        for a, b in TEMPORAL_KEY_MAP:
            if key.startswith(a):
                return b + key[len(a):]
    return key


def seeded_module(ctor, seed: int):
    """``ctor()`` under ``torch.manual_seed(seed)``, every BatchNorm given non-trivial seeded statistics, eval mode."""
    import torch
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        m = ctor()
    finally:
        torch.random.set_rng_state(state)
    g = torch.Generator().manual_seed(seed + 12345)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    return m.eval()


def seeded_clip(shape, seed: int):
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.randn(tuple(shape), generator=g)


def state_sha(state_dict, key_map=()) -> str:
    """SHA-256 over (mapped name, float32 bytes) of every floating-point entry, in the reference's key order."""
    sha = hashlib.sha256()
    for k, v in state_dict.items():
        if not v.is_floating_point():
            continue                      # num_batches_tracked
        for a, b in key_map:
            if k.startswith(a):
                k = b + k[len(a):]
                break
        sha.update(k.encode())
        sha.update(v.detach().float().contiguous().cpu().numpy().tobytes())
    return sha.hexdigest()

"""Thin torch-facing wrappers over the C ABI: device buffers and streams come from torch,
every computation happens in librva.so (HIP).  No operation here has a torch/CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N


_CHAIN_STREAMS: dict = {}


def _probe_lanes(dev_index: int, pool: list, spin_cycles: int = 150_000) -> list:
    """Which streams of ``pool`` (plus the default stream, index -1) can run kernels side by side?  The HIP runtime multiplexes
    a process's streams onto a few hardware queues ("lanes": four on this stack, whatever GPU_MAX_HW_QUEUES says), and two streams
    on one lane serialise.  Measured, not assumed: a spin kernel (~70 us) on stream i and one on stream j, timed from the host --
    the pair takes one spin when the two have lanes of their own and two when they share one.  Returns the matrix of pair times
    (ms) as a dict; a few milliseconds at start-up."""
    import time
    null = torch.cuda.default_stream(dev_index)
    streams = {-1: null, **{i: st for i, st in enumerate(pool)}}
    for st in streams.values():                                   # first use of every stream, in order
        with torch.cuda.stream(st):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize(dev_index)

    def pair(a, b):
        best = float("inf")
        for _ in range(2):
            torch.cuda.synchronize(dev_index)
            t0 = time.perf_counter()
            with torch.cuda.stream(streams[a]):
                torch.cuda._sleep(spin_cycles)
            if b is not None:
                with torch.cuda.stream(streams[b]):
                    torch.cuda._sleep(spin_cycles)
            torch.cuda.synchronize(dev_index)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best
    one = pair(0, None)
    return {"one": one, "pairs": {(a, b): pair(a, b) for a in streams for b in streams if a < b}}


def chain_streams(device, n: int) -> list:
    """The HIP streams the tick chains of a device run on -- one set per process, shared by everything that runs whole forward
    passes side by side (``PipelinedTicks``, the kernel selection).  Which hardware lane a stream gets is the runtime's
    business and two chains on one lane serialise (three chains measured 18.0 k against 21.3 k frames/s by stream choice alone
    in round 3; four chains 6 070 against 3 690 frames/s at 4 x YOLOv8m in round 4), so the set is CHOSEN: a pool of eight streams
    is probed pairwise (``_probe_lanes``) and the chains take streams that share a lane with no other chain -- and, while such
    streams last, not with the default stream either.  ``RVA_CHAIN_PROBE=0`` takes the first streams created instead."""
    import os
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _CHAIN_STREAMS.get(idx)
    if st is None:
        st = _CHAIN_STREAMS[idx] = {"order": [], "report": None}
        pool = [torch.cuda.Stream(device=idx) for _ in range(8)]
        order = list(range(8))
        if os.environ.get("RVA_CHAIN_PROBE", "1") == "1":
            rep = _probe_lanes(idx, pool)
            one = rep["one"]
            shares = lambda a, b: rep["pairs"][(min(a, b), max(a, b))] > 1.6 * one      # noqa: E731  (two spins back to back: 2 x)
            chosen: list = []
            for avoid_null in (True, False):                       # first pass: lanes of their own AND not the default stream's
                for i in range(8):
                    if i in chosen or any(shares(i, c) for c in chosen) or (avoid_null and shares(-1, i)):
                        continue
                    chosen.append(i)
            order = chosen + [i for i in range(8) if i not in chosen]
            st["report"] = {"spin_ms": round(one, 4), "distinct_lanes_found": len(chosen),
                            "lane_sharing_pairs": sorted([list(k) for k, v in rep["pairs"].items() if v > 1.6 * one]),
                            "chain_streams": order[:4]}
        st["pool"], st["order"] = pool, order
    return [st["pool"][i] for i in st["order"][:n]] if n <= 8 else [st["pool"][i] for i in st["order"]] + \
        [torch.cuda.Stream(device=idx) for _ in range(n - 8)]


def chain_stream_report(device) -> Optional[dict]:
    """What the lane probe found (None before the first ``chain_streams`` call or with RVA_CHAIN_PROBE=0)."""
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _CHAIN_STREAMS.get(idx)
    return st["report"] if st else None


class _Roctx:
    """roctx ranges around the stages of a tick (tracing), behind ``RVA_ROCTX=1``: K1, network, tail of every
    tick show up as named host ranges in a ``rocprofv3 --marker-trace`` timeline.  Off by default (two library calls per
    stage); a missing ``libroctx64.so`` switches it off silently -- it is a tracing aid, not part of the hot path."""

    def __init__(self):
        import os
        self.lib = None
        if os.environ.get("RVA_ROCTX") == "1":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    self.lib = C.CDLL(name)
                    self.lib.roctxRangePushA.argtypes = [C.c_char_p]
                    break
                except (OSError, AttributeError):
                    self.lib = None

    def range(self, name: str):
        return _RoctxRange(self.lib, name)


class _RoctxRange:
    __slots__ = ("lib", "name")

    def __init__(self, lib, name):
        self.lib, self.name = lib, name

    def __enter__(self):
        if self.lib is not None:
            self.lib.roctxRangePushA(self.name.encode())
        return self

    def __exit__(self, *exc):
        if self.lib is not None:
            self.lib.roctxRangePop()
        return False


_ROCTX: Optional[_Roctx] = None


def roctx(name: str) -> _RoctxRange:
    global _ROCTX
    if _ROCTX is None:
        _ROCTX = _Roctx()
    return _ROCTX.range(name)


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live in HBM (got a {t.device} tensor): the HIP hot path has no CPU fallback")


_CTX: dict = {}


def context(device: Optional[int] = None) -> N.Context:
    """Per-device singleton context."""
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible to torch: the MI355X hot path cannot run (no CPU fallback)")
    dev = torch.cuda.current_device() if device is None else device
    if dev not in _CTX:
        _CTX[dev] = N.Context(dev)
    return _CTX[dev]


# ------------------------------------------------------------------------------------------ K1
@dataclass
class Nv12Surface:
    """A decoded frame resident in HBM: pitch-linear Y plane + interleaved UV plane."""
    y: torch.Tensor    # uint8 [h, pitch]
    uv: torch.Tensor   # uint8 [h/2, pitch]
    width: int
    height: int
    mask: Optional[torch.Tensor] = None   # optional ROI mask uint8 [h, w] (0 = outside), see gates.rasterize_polygons

    @property
    def pitch(self) -> int:
        return int(self.y.stride(0))

    @property
    def shape(self) -> Tuple[int, int, int]:  # what packet.frame.shape[:2] is used for
        return (self.height, self.width, 3)

    @staticmethod
    def from_numpy(y: np.ndarray, uv: np.ndarray, width: int, height: int, device="cuda") -> "Nv12Surface":
        return Nv12Surface(torch.from_numpy(np.ascontiguousarray(y)).to(device),
                           torch.from_numpy(np.ascontiguousarray(uv)).to(device), width, height)


def preprocess_nv12(surfaces: Sequence[Nv12Surface], dst_hw=(640, 640), half: bool = True,
                    out: Optional[torch.Tensor] = None, clip: bool = False, ctx: Optional[N.Context] = None,
                    content_only: bool = False):
    """K1 over a tick of NV12 surfaces -> ``out[n,3,H,W]`` (+ letterbox meta unless ``clip``).  ``content_only``: ``out``
    already holds the letterbox border of this geometry (an earlier full call wrote it); write the content rows only."""
    ctx = ctx or context()
    n = len(surfaces)
    w, h = surfaces[0].width, surfaces[0].height
    for s in surfaces:
        _require_cuda(s.y, "NV12 surface")
        if (s.width, s.height) != (w, h):
            raise ValueError("all surfaces of one launch must share one geometry")
    dt = torch.float16 if half else torch.float32
    if out is None:
        out = torch.empty((n, 3, dst_hw[0], dst_hw[1]), dtype=dt, device=surfaces[0].y.device)
    _require_cuda(out, "output tensor")
    assert out.is_contiguous() and out.dtype == dt and tuple(out.shape) == (n, 3, dst_hw[0], dst_hw[1])
    L = N.lib()
    meta = N.Letterbox()
    for b0 in range(0, n, N.RVA_MAX_BATCH):
        chunk = surfaces[b0:b0 + N.RVA_MAX_BATCH]
        yp, _k1 = N.ptr_array([s.y.data_ptr() for s in chunk])
        up, _k2 = N.ptr_array([s.uv.data_ptr() for s in chunk])
        pp, _k3 = N.i32_array([s.pitch for s in chunk])
        optr = C.c_void_p(out[b0].data_ptr())
        if clip:
            rc = L.rva_preprocess_clip_nv12_batch(ctx.handle, yp, up, pp, len(chunk), w, h, optr,
                                                  N.RVA_F16 if half else N.RVA_F32, dst_hw[1], dst_hw[0], _stream_ptr())
        elif any(s.mask is not None for s in chunk):          # apply_roi in front of the resize
            mp, _k4 = N.ptr_array([s.mask.data_ptr() if s.mask is not None else 0 for s in chunk])
            rc = L.rva_preprocess_nv12_masked_batch(ctx.handle, yp, up, pp, mp, len(chunk), w, h, optr,
                                                    N.RVA_F16 if half else N.RVA_F32, dst_hw[1], dst_hw[0], C.byref(meta),
                                                    _stream_ptr())
        else:
            fn = L.rva_preprocess_nv12_content_batch if content_only else L.rva_preprocess_nv12_batch
            rc = fn(ctx.handle, yp, up, pp, len(chunk), w, h, optr, N.RVA_F16 if half else N.RVA_F32, dst_hw[1], dst_hw[0],
                    C.byref(meta), _stream_ptr())
        ctx.check(rc, "rva_preprocess_nv12_batch")
    return (out, None) if clip else (out, meta)


_FRAME_DT = {torch.float16: N.RVA_F16, torch.float32: N.RVA_F32, torch.float64: N.RVA_F64}


def preprocess_frames(frames: Sequence, dst_hw=(224, 224), norm: int = N.NORM_IMAGENET_F32, layout: int = N.LAYOUT_NCHW,
                      dtype: torch.dtype = torch.float32, out: Optional[torch.Tensor] = None,
                      ctx: Optional[N.Context] = None) -> torch.Tensor:
    """Frame pre-process of the classification / temporal heads (``rva_preprocess_frames_*``): stretch
    resize, RGB, /255, (x-mean)/std with the constants and precision ``norm`` names.  ``frames``: NV12 surfaces or
    device BGR uint8 ``[h,w,3]`` tensors of one geometry.  Returns ``[n,3,H,W]`` (``LAYOUT_NCHW``) or ``[3,n,H,W]``
    (``LAYOUT_CNHW``, the 3D-CNN clip layout)."""
    ctx = ctx or context()
    n = len(frames)
    if n == 0 or n > N.RVA_MAX_BATCH:
        raise ValueError(f"1..{N.RVA_MAX_BATCH} frames per call")
    H, W = int(dst_hw[0]), int(dst_hw[1])
    shape = (n, 3, H, W) if layout == N.LAYOUT_NCHW else (3, n, H, W)
    nv12 = isinstance(frames[0], Nv12Surface)
    dev = frames[0].y.device if nv12 else frames[0].device
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=dev)
    _require_cuda(out, "output tensor")
    assert out.is_contiguous() and out.dtype == dtype and tuple(out.shape) == shape
    L = N.lib()
    if nv12:
        w, h = frames[0].width, frames[0].height
        for s in frames:
            _require_cuda(s.y, "NV12 surface")
            if (s.width, s.height) != (w, h):
                raise ValueError("all surfaces of one launch must share one geometry")
        yp, _k1 = N.ptr_array([s.y.data_ptr() for s in frames])
        up, _k2 = N.ptr_array([s.uv.data_ptr() for s in frames])
        pp, _k3 = N.i32_array([s.pitch for s in frames])
        rc = L.rva_preprocess_frames_nv12_batch(ctx.handle, yp, up, pp, n, w, h, C.c_void_p(out.data_ptr()), _FRAME_DT[dtype],
                                                W, H, norm, layout, _stream_ptr())
    else:
        h, w = int(frames[0].shape[0]), int(frames[0].shape[1])
        for f in frames:
            _require_cuda(f, "BGR frame")
            assert f.dtype == torch.uint8 and f.dim() == 3 and f.shape[2] == 3 and f.stride(2) == 1 and f.stride(1) == 3
            if (int(f.shape[0]), int(f.shape[1])) != (h, w):
                raise ValueError("all frames of one launch must share one geometry")
        fp, _k1 = N.ptr_array([f.data_ptr() for f in frames])
        rb, _k2 = N.i32_array([int(f.stride(0)) for f in frames])
        rc = L.rva_preprocess_frames_bgr_batch(ctx.handle, fp, rb, n, w, h, C.c_void_p(out.data_ptr()), _FRAME_DT[dtype], W, H,
                                               norm, layout, _stream_ptr())
    ctx.check(rc, "rva_preprocess_frames_batch")
    return out


def resize_nv12_to_bgr(surfaces: Sequence[Nv12Surface], dst_wh: Tuple[int, int], out: Optional[torch.Tensor] = None,
                       ctx: Optional[N.Context] = None) -> torch.Tensor:
    """apply_roi (if a surface carries a mask) + downsample (utils/frame_filter.py:43-57): uint8 BGR images
    ``[n, dst_h, dst_w, 3]`` on the device, ready for :func:`preprocess_bgr`."""
    ctx = ctx or context()
    n, (dw, dh) = len(surfaces), dst_wh
    w, h = surfaces[0].width, surfaces[0].height
    if out is None:
        out = torch.empty((n, dh, dw, 3), dtype=torch.uint8, device=surfaces[0].y.device)
    yp, _k1 = N.ptr_array([s.y.data_ptr() for s in surfaces])
    up, _k2 = N.ptr_array([s.uv.data_ptr() for s in surfaces])
    pp, _k3 = N.i32_array([s.pitch for s in surfaces])
    mp, _k4 = N.ptr_array([s.mask.data_ptr() if s.mask is not None else 0 for s in surfaces])
    rc = N.lib().rva_resize_nv12_to_bgr_batch(ctx.handle, yp, up, pp, mp, n, w, h, C.c_void_p(out.data_ptr()), dw, dh,
                                              _stream_ptr())
    ctx.check(rc, "rva_resize_nv12_to_bgr_batch")
    return out


def preprocess_bgr(frames: Sequence[torch.Tensor], dst_hw=(640, 640), half: bool = True,
                   out: Optional[torch.Tensor] = None, clip: bool = False, ctx: Optional[N.Context] = None):
    """K1 over device copies of BGR uint8 [h,w,3] frames (the reference's FramePacket.frame)."""
    ctx = ctx or context()
    n = len(frames)
    h, w = int(frames[0].shape[0]), int(frames[0].shape[1])
    for f in frames:
        _require_cuda(f, "BGR frame")
        assert f.dtype == torch.uint8 and f.dim() == 3 and f.shape[2] == 3 and f.stride(2) == 1 and f.stride(1) == 3
        if (int(f.shape[0]), int(f.shape[1])) != (h, w):
            raise ValueError("all frames of one launch must share one geometry")
    dt = torch.float16 if half else torch.float32
    if out is None:
        out = torch.empty((n, 3, dst_hw[0], dst_hw[1]), dtype=dt, device=frames[0].device)
    assert out.is_contiguous() and out.dtype == dt
    L = N.lib()
    meta = N.Letterbox()
    for b0 in range(0, n, N.RVA_MAX_BATCH):
        chunk = frames[b0:b0 + N.RVA_MAX_BATCH]
        fp, _k1 = N.ptr_array([f.data_ptr() for f in chunk])
        rb, _k2 = N.i32_array([int(f.stride(0)) for f in chunk])
        optr = C.c_void_p(out[b0].data_ptr())
        if clip:
            rc = L.rva_preprocess_clip_bgr_batch(ctx.handle, fp, rb, len(chunk), w, h, optr,
                                                 N.RVA_F16 if half else N.RVA_F32, dst_hw[1], dst_hw[0], _stream_ptr())
        else:
            rc = L.rva_preprocess_bgr_batch(ctx.handle, fp, rb, len(chunk), w, h, optr,
                                            N.RVA_F16 if half else N.RVA_F32, dst_hw[1], dst_hw[0], C.byref(meta),
                                            _stream_ptr())
        ctx.check(rc, "rva_preprocess_bgr_batch")
    return (out, None) if clip else (out, meta)


# ------------------------------------------------------------------------------------------ K2+K3
@dataclass
class PostBuffers:
    """Device-resident result of the post-process for a batch (SoA, [B, max_det])."""
    boxes: torch.Tensor    # float32 [B, max_det, 4]
    scores: torch.Tensor   # float32 [B, max_det]
    cls: torch.Tensor      # int32
    anchor: torch.Tensor   # int32
    cand: torch.Tensor     # int32
    counts: torch.Tensor   # int32 [B]
    ncand: torch.Tensor    # int32 [B]
    max_det: int

    @staticmethod
    def allocate(batch: int, max_det: int, device) -> "PostBuffers":
        i32 = dict(dtype=torch.int32, device=device)
        return PostBuffers(torch.empty((batch, max_det, 4), dtype=torch.float32, device=device),
                           torch.empty((batch, max_det), dtype=torch.float32, device=device),
                           torch.empty((batch, max_det), **i32), torch.empty((batch, max_det), **i32),
                           torch.empty((batch, max_det), **i32), torch.zeros((batch,), **i32),
                           torch.zeros((batch,), **i32), max_det)

    def to_host(self) -> List[dict]:
        """One D2H sync; per-image dicts in NMS order."""
        counts = self.counts.cpu().numpy()
        m = int(counts.max()) if len(counts) else 0
        boxes = self.boxes[:, :m].cpu().numpy(); scores = self.scores[:, :m].cpu().numpy()
        cls = self.cls[:, :m].cpu().numpy(); anchor = self.anchor[:, :m].cpu().numpy()
        cand = self.cand[:, :m].cpu().numpy(); ncand = self.ncand.cpu().numpy()
        return [dict(n=int(c), boxes=boxes[b, :c], conf=scores[b, :c], cls=cls[b, :c], anchor=anchor[b, :c],
                     keep=cand[b, :c], n_cand=int(ncand[b])) for b, c in enumerate(counts)]


def postprocess(raw: torch.Tensor, conf_thr: float, iou_thr: float, classes: Optional[Sequence[int]],
                metas: Sequence[N.Letterbox], max_det: Optional[int] = None, out: Optional[PostBuffers] = None,
                ctx: Optional[N.Context] = None) -> PostBuffers:
    """K2+K3 over ``raw[B, d1, d2]`` (float16/float32, device)."""
    ctx = ctx or context()
    _require_cuda(raw, "head tensor")
    if raw.dim() == 2:
        raw = raw.unsqueeze(0)
    assert raw.dim() == 3 and raw.is_contiguous() and raw.dtype in (torch.float16, torch.float32)
    B, d1, d2 = (int(v) for v in raw.shape)
    A = d2 if d1 < d2 else d1
    max_det = max_det or A
    if out is None:
        out = PostBuffers.allocate(B, max_det, raw.device)
    assert out.max_det == max_det and out.counts.shape[0] >= B
    marr = (N.Letterbox * len(metas))(*metas)
    cls_p, _keep = (N.i32_array(list(classes)) if classes else (None, None))
    rc = N.lib().rva_postprocess_batch(
        ctx.handle, C.c_void_p(raw.data_ptr()), N.RVA_F16 if raw.dtype == torch.float16 else N.RVA_F32, B, d1, d2,
        float(conf_thr), float(iou_thr), cls_p, len(classes) if classes else 0, marr, len(metas), max_det,
        C.c_void_p(out.boxes.data_ptr()), C.c_void_p(out.scores.data_ptr()), C.c_void_p(out.cls.data_ptr()),
        C.c_void_p(out.anchor.data_ptr()), C.c_void_p(out.cand.data_ptr()), C.c_void_p(out.counts.data_ptr()),
        C.c_void_p(out.ncand.data_ptr()), _stream_ptr())
    ctx.check(rc, "rva_postprocess_batch")
    return out


def post_status(ctx: Optional[N.Context] = None) -> int:
    ctx = ctx or context()
    f = C.c_int()
    ctx.check(N.lib().rva_post_status(ctx.handle, _stream_ptr(), C.byref(f)), "rva_post_status")
    return f.value


def post_filter_stats(ctx: Optional[N.Context] = None) -> int:
    """Images since the last call whose NMS went through K3's centre-bin filter (diagnostic, ``rva_post_filter_stats``)."""
    ctx = ctx or context()
    f = C.c_int()
    ctx.check(N.lib().rva_post_filter_stats(ctx.handle, _stream_ptr(), C.byref(f)), "rva_post_filter_stats")
    return f.value


# ------------------------------------------------------------------------------------------ K4
class DeviceTracker:
    """All per-stream track tables of this process, resident in HBM (wraps ``rva_tracker``)."""

    def __init__(self, n_streams: int, max_age: int = 30, max_iou_distance: float = 0.7, min_hits: int = 3,
                 capacity: int = 1024, ctx: Optional[N.Context] = None):
        self.ctx = ctx or context()
        self.n_streams, self.capacity = n_streams, capacity
        h = C.c_void_p()
        self.ctx.check(N.lib().rva_tracker_create(self.ctx.handle, n_streams, capacity, int(max_age),
                                                  float(max_iou_distance), int(min_hits), C.byref(h)),
                       "rva_tracker_create")
        self.handle = h
        self._new_counts_ptr = N.lib().rva_tracker_new_counts(h)

    def close(self):
        if getattr(self, "handle", None):
            N.lib().rva_tracker_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def update_from_post(self, slot_of_stream: Sequence[int], post: Optional[PostBuffers], filter_thr: float,
                         gated: bool = False, motion: Optional[Tuple[torch.Tensor, Sequence[int]]] = None):
        """K4 over the streams of one launch.  ``slot_of_stream[s]``: batch row of ``post``, -1 no frame, -2 skipped frame,
        -3 owned by another launch of this tick.  ``gated``: decide the motion / adaptive-fps gates on the device
        (``set_gates``); ``motion = (counts int32 device tensor, row per stream or -1)`` feeds the motion gate."""
        if len(slot_of_stream) != self.n_streams:
            raise ValueError(f"slot_of_stream needs {self.n_streams} entries (one per tracker stream)")
        sp, _k = N.i32_array(list(slot_of_stream))
        nul = C.c_void_p(None)
        arrs = (nul, nul, nul, nul, 0) if post is None else \
            (C.c_void_p(post.boxes.data_ptr()), C.c_void_p(post.scores.data_ptr()), C.c_void_p(post.cls.data_ptr()),
             C.c_void_p(post.counts.data_ptr()), post.max_det)
        if gated:
            if motion is not None:
                cnt, rows = motion
                assert cnt.is_cuda and cnt.dtype == torch.int32 and len(rows) == self.n_streams
                mp, _m = N.i32_array(list(rows))
                cp = C.c_void_p(cnt.data_ptr())
            else:
                mp, cp = None, nul
            rc = N.lib().rva_tracker_update_gated_f32(self.handle, sp, *arrs, float(filter_thr), cp, mp, _stream_ptr())
        else:
            rc = N.lib().rva_tracker_update_f32(self.handle, sp, *arrs, float(filter_thr), _stream_ptr())
        self.ctx.check(rc, "rva_tracker_update_f32")

    def set_gates(self, adaptive_enabled: Sequence[int], max_process_every: Sequence[int], idle_tolerance: Sequence[int],
                  motion_min_count: Sequence[int], reset: bool = True) -> None:
        """Per-stream parameters of the device-side gates (host-synchronous; see rva_tracker_set_gates)."""
        arrs = []
        for v in (adaptive_enabled, max_process_every, idle_tolerance, motion_min_count):
            if len(v) != self.n_streams:
                raise ValueError(f"gate arrays need {self.n_streams} entries")
            arrs.append(N.i32_array([int(x) for x in v]))
        self.ctx.check(N.lib().rva_tracker_set_gates(self.handle, arrs[0][0], arrs[1][0], arrs[2][0], arrs[3][0], 1 if reset else 0),
                       "rva_tracker_set_gates")

    def snapshot_status(self, slot: int) -> Tuple[np.ndarray, np.ndarray, int]:
        """(emitted[S], processed[S], flags) that rode along in snapshot ``slot`` (valid once the slot's copy is done)."""
        em = np.empty(self.n_streams, np.int32); pr = np.empty(self.n_streams, np.int32)
        fl = C.c_int32()
        self.ctx.check(N.lib().rva_tracker_snapshot_status(self.handle, slot, C.c_void_p(em.ctypes.data), C.c_void_p(pr.ctypes.data),
                                                           C.byref(fl)), "rva_tracker_snapshot_status")
        return em, pr, int(fl.value)

    @staticmethod
    def raise_on_flags(flags: int) -> None:
        """The reference's NMS and tracker are unbounded; the device tables are not.  Any truncation is an error."""
        if flags & 0x1:
            raise RuntimeError("tracker table capacity exceeded (detections were dropped); construct IouTracker with a larger `capacity`")
        if flags & 0x100:
            raise RuntimeError("post-process kept more boxes than max_det for an image (survivors were dropped)")
        if flags & 0x200:
            raise RuntimeError("post-process: more thresholded anchors than the NMS sort holds (candidates were dropped)")

    def update_from_host(self, per_stream: dict, others_untouched: bool = False):
        """``per_stream[s] = (boxes f64 [D,4], conf f64 [D], cls i64 [D])`` for the streams to update.  The other streams
        count as "no update this tick" unless ``others_untouched`` (another launch of the same tick owns them)."""
        active = [2 if others_untouched else 0] * self.n_streams
        offs = [0] * (self.n_streams + 1)
        bl, cl, kl = [], [], []
        tot = 0
        for s in range(self.n_streams):
            offs[s] = tot
            if s in per_stream:
                b, c, k = per_stream[s]
                b = np.ascontiguousarray(b, np.float64).reshape(-1, 4)
                active[s] = 1
                bl.append(b); cl.append(np.ascontiguousarray(c, np.float64)); kl.append(np.ascontiguousarray(k, np.int64))
                tot += len(b)
        offs[self.n_streams] = tot
        dev = torch.device("cuda", self.ctx.device)
        if tot:
            db = torch.from_numpy(np.concatenate(bl)).to(dev)
            dc = torch.from_numpy(np.concatenate(cl)).to(dev)
            dk = torch.from_numpy(np.concatenate(kl)).to(dev)
        else:
            db = torch.zeros((1, 4), dtype=torch.float64, device=dev)
            dc = torch.zeros((1,), dtype=torch.float64, device=dev)
            dk = torch.zeros((1,), dtype=torch.int64, device=dev)
        ap, _k1 = N.i32_array(active)
        op, _k2 = N.i32_array(offs)
        rc = N.lib().rva_tracker_update_f64(self.handle, ap, op, C.c_void_p(db.data_ptr()), C.c_void_p(dc.data_ptr()),
                                            C.c_void_p(dk.data_ptr()), _stream_ptr())
        self.ctx.check(rc, "rva_tracker_update_f64")
        self._keepalive = (db, dc, dk)

    def set_box_scale(self, scales: Sequence[float]) -> None:
        """_rescale_detections (pipeline.py:224-240): per-stream float64 multiplier for boxes fed by update_from_post."""
        arr = (C.c_double * self.n_streams)(*[float(v) for v in scales])
        self.ctx.check(N.lib().rva_tracker_set_box_scale(self.handle, arr), "rva_tracker_set_box_scale")

    def new_counts_tensor(self) -> torch.Tensor:
        """Zero-copy int32[n_streams] view of the device new-track counts of the last update."""
        # built through the array interface so torch wraps, not copies, the librva-owned buffer
        class _Holder:
            pass
        hld = _Holder()
        hld.__cuda_array_interface__ = {"shape": (self.n_streams,), "typestr": "<i4",
                                        "data": (int(self._new_counts_ptr), False), "version": 3}
        return torch.as_tensor(hld, device=torch.device("cuda", self.ctx.device))

    def assign_ids(self, counts_all: Optional[torch.Tensor] = None, global_index: Optional[Sequence[int]] = None):
        if counts_all is None:
            rc = N.lib().rva_tracker_assign_ids(self.handle, C.c_void_p(None), 0, None, _stream_ptr())
        else:
            assert counts_all.is_cuda and counts_all.dtype == torch.int32 and counts_all.is_contiguous()
            if global_index is not None and len(global_index) != self.n_streams:   # the C side reads n_streams entries
                raise ValueError(f"global_index needs {self.n_streams} entries (pad unused tracker streams with 0)")
            gp, _k = N.i32_array(list(global_index)) if global_index is not None else (None, None)
            rc = N.lib().rva_tracker_assign_ids(self.handle, C.c_void_p(counts_all.data_ptr()), int(counts_all.numel()),
                                                gp, _stream_ptr())
        self.ctx.check(rc, "rva_tracker_assign_ids")

    def read(self, stream_id: int) -> dict:
        cap = self.capacity
        ids = np.empty(cap, np.int64); cls = np.empty(cap, np.int32); age = np.empty(cap, np.int32)
        hits = np.empty(cap, np.int32); conf = np.empty(cap, np.float64); box = np.empty((cap, 4), np.float64)
        ld = np.empty(cap, np.int32)
        n = C.c_int32()
        rc = N.lib().rva_tracker_read(self.handle, stream_id, cap, C.c_void_p(ids.ctypes.data), C.c_void_p(cls.ctypes.data),
                                      C.c_void_p(age.ctypes.data), C.c_void_p(hits.ctypes.data),
                                      C.c_void_p(conf.ctypes.data), C.c_void_p(box.ctypes.data),
                                      C.c_void_p(ld.ctypes.data), C.byref(n), _stream_ptr())
        self.ctx.check(rc, "rva_tracker_read")
        m = n.value
        return dict(n=m, id=ids[:m], cls=cls[:m], age=age[:m], hits=hits[:m], conf=conf[:m], boxes=box[:m],
                    last_det=ld[:m])

    def snapshot_async(self, slot: int) -> None:
        self.ctx.check(N.lib().rva_tracker_snapshot_async(self.handle, slot, _stream_ptr()), "rva_tracker_snapshot_async")

    def snapshot_fetch(self, slot: int, wait: bool = True) -> List[dict]:
        S, cap = self.n_streams, self.capacity
        ids = np.empty((S, cap), np.int64); cls = np.empty((S, cap), np.int32); age = np.empty((S, cap), np.int32)
        hits = np.empty((S, cap), np.int32); conf = np.empty((S, cap), np.float64); box = np.empty((S, cap, 4), np.float64)
        cnt = np.empty(S, np.int32); ld = np.empty((S, cap), np.int32)
        rc = N.lib().rva_tracker_snapshot_fetch(self.handle, slot, 1 if wait else 0, C.c_void_p(ids.ctypes.data), C.c_void_p(cls.ctypes.data),
                                                C.c_void_p(age.ctypes.data), C.c_void_p(hits.ctypes.data),
                                                C.c_void_p(conf.ctypes.data), C.c_void_p(box.ctypes.data),
                                                C.c_void_p(ld.ctypes.data), C.c_void_p(cnt.ctypes.data))
        self.ctx.check(rc, "rva_tracker_snapshot_fetch")
        return [dict(n=int(cnt[s]), id=ids[s, :cnt[s]], cls=cls[s, :cnt[s]], age=age[s, :cnt[s]], hits=hits[s, :cnt[s]],
                     conf=conf[s, :cnt[s]], boxes=box[s, :cnt[s]], last_det=ld[s, :cnt[s]]) for s in range(S)]

    def read_all(self) -> List[dict]:
        S, cap = self.n_streams, self.capacity
        ids = np.empty((S, cap), np.int64); cls = np.empty((S, cap), np.int32); age = np.empty((S, cap), np.int32)
        hits = np.empty((S, cap), np.int32); conf = np.empty((S, cap), np.float64); box = np.empty((S, cap, 4), np.float64)
        cnt = np.empty(S, np.int32); ld = np.empty((S, cap), np.int32)
        rc = N.lib().rva_tracker_read_all(self.handle, C.c_void_p(ids.ctypes.data), C.c_void_p(cls.ctypes.data),
                                          C.c_void_p(age.ctypes.data), C.c_void_p(hits.ctypes.data),
                                          C.c_void_p(conf.ctypes.data), C.c_void_p(box.ctypes.data),
                                          C.c_void_p(ld.ctypes.data), C.c_void_p(cnt.ctypes.data), _stream_ptr())
        self.ctx.check(rc, "rva_tracker_read_all")
        return [dict(n=int(cnt[s]), id=ids[s, :cnt[s]], cls=cls[s, :cnt[s]], age=age[s, :cnt[s]], hits=hits[s, :cnt[s]],
                     conf=conf[s, :cnt[s]], boxes=box[s, :cnt[s]], last_det=ld[s, :cnt[s]]) for s in range(S)]

    def state(self) -> Tuple[int, int]:
        nid = C.c_int64(); fl = C.c_int()
        self.ctx.check(N.lib().rva_tracker_state(self.handle, C.byref(nid), C.byref(fl), _stream_ptr()), "rva_tracker_state")
        return nid.value, fl.value

    def set_next_id(self, v: int):
        self.ctx.check(N.lib().rva_tracker_set_next_id(self.handle, int(v), _stream_ptr()), "rva_tracker_set_next_id")


def jpeg_encode_bgr(img: torch.Tensor, quality: int, ctx: Optional[N.Context] = None) -> bytes:
    """K7: a uint8 BGR image ``[h, w, 3]`` in HBM (what ``preview.render_nv12`` returns) -> baseline JFIF bytes, encoded on the
    device (``rva_jpeg_encode_bgr``: libjpeg's arithmetic, 4:2:0, Annex-K tables, one restart interval per MCU row).  Only the
    finished stream crosses PCIe.  Replaces ``cv2.imencode('.jpg', ...)`` of sinks/kafka_sink.py:260-284."""
    ctx = ctx or context()
    _require_cuda(img, "image")
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or img.stride(2) != 1 or img.stride(1) != 3:
        raise ValueError("jpeg_encode_bgr takes a uint8 [h, w, 3] device tensor with packed pixels")
    h, w = int(img.shape[0]), int(img.shape[1])
    L = N.lib()
    cap = int(L.rva_jpeg_max_bytes(w, h))
    bufs = ctx.__dict__.setdefault("_jpeg_bufs", {})
    if cap not in bufs:
        bufs[cap] = (torch.empty(cap, dtype=torch.uint8, device=img.device), torch.zeros(1, dtype=torch.int32, device=img.device))
    out, size = bufs[cap]
    s = _stream_ptr()
    ctx.check(L.rva_jpeg_encode_bgr(ctx.handle, C.c_void_p(img.data_ptr()), int(img.stride(0)), w, h, int(quality), C.c_void_p(out.data_ptr()),
                                    cap, C.c_void_p(size.data_ptr()), s), "rva_jpeg_encode_bgr")
    n = int(size.item())                                            # the one host sync
    flags = C.c_int(0)
    ctx.check(L.rva_jpeg_status(ctx.handle, s, C.byref(flags)), "rva_jpeg_status")
    if flags.value & 1 or not 0 < n <= cap:
        raise RuntimeError(f"device JPEG encoder: the stream did not fit {cap} bytes ({w}x{h}, quality {quality})")
    return out[:n].cpu().numpy().tobytes()

"""Annotated frame preview: what the reference attaches to a Kafka message as ``frame_jpeg``
(sinks/kafka_sink.py:134-146, 151-294) and what ``StreamWorker._maybe_save_snapshot`` writes every five minutes
(pipeline.py:264-290).

The reference copies every 1080p frame to the host, draws on it with OpenCV and encodes it on a worker thread.  Here
  * the POLICY -- send at most one frame per stream per 0.1 s, quality from the number of tracks, colour per class --
    and the DRAWING PLAN -- target size (downscale above 1920x1080), the outline / label-bar rectangles with their integer
    coordinates, the label text and origin, the encoder parameters -- are host logic, restated from the reference and
    pinned against a call-level recording of its own ``_render_frame`` (tests/golden/preview_plan.json);
  * the PIXELS are produced by one HIP launch on the NV12 surface in HBM (``rva_preview_nv12``: colour conversion, integer
    box downscale, filled rectangles, a built-in 5x7 font), and only the finished preview crosses PCIe;
  * the JPEG ENCODER runs on the device too since round 4 (``ops.jpeg_encode_bgr`` / ``rva_jpeg_encode_bgr``, K7: libjpeg's
    arithmetic step by step -- a decoder reconstructs exactly the picture it reconstructs from libjpeg's own file of that
    quality; baseline instead of the reference's progressive + optimised entropy coding of the same coefficients), so only the
    finished stream (100-300 KB instead of a 6.2 MB image) crosses PCIe; WebP (only when the reference's
    ``webp_available`` is set and quality >= 80) stays Pillow's libwebp on the host.  The result is a valid
    ``data:image/...;base64,`` URL of the kind the dashboard displays.
How OpenCV covers pixels for a 2-px outline, its Hershey glyphs, INTER_AREA at non-integer ratios and its encoder's exact
bits are OpenCV-internal and cannot be pinned here (cv2 is absent): the plan is exact, the raster is this module's own.
"""
from __future__ import annotations

import base64
import ctypes as C
import io
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

FRAME_SEND_INTERVAL = 0.1          # kafka_sink.py:49: at most 10 previews per second per stream
INTER_AREA, FONT_HERSHEY_SIMPLEX = 3, 0          # the cv2 constants the recorded plan carries
IMWRITE_JPEG_QUALITY, IMWRITE_JPEG_PROGRESSIVE, IMWRITE_JPEG_OPTIMIZE, IMWRITE_WEBP_QUALITY = 1, 2, 3, 64


def color_for(class_id: int) -> Tuple[int, int, int]:
    """kafka_sink.py:296-302: a stable BGR colour per class id."""
    seed = (hash(class_id) & 0xFFFFFF) or 0xFFAA33
    return int(seed & 0xFF), int((seed >> 8) & 0xFF), int((seed >> 16) & 0xFF)


class PreviewPolicy:
    """Rate limiting and adaptive quality of KafkaSink (kafka_sink.py:47-56, 151-198)."""

    def __init__(self, frame_quality: int = 75, webp_available: bool = False, clock: Callable[[], float] = time.time):
        self.base_quality = frame_quality
        self.use_adaptive_quality = True
        self.webp_available = webp_available
        self._last: dict = {}
        self._clock = clock

    def should_send_frame(self, stream_name: str) -> bool:
        now = self._clock()
        if now - self._last.get(stream_name, 0.0) >= FRAME_SEND_INTERVAL:
            self._last[stream_name] = now
            return True
        return False

    def adaptive_quality(self, detection_count: int) -> int:
        if not self.use_adaptive_quality:
            return self.base_quality
        boost = -10 if detection_count == 0 else (0 if detection_count <= 3 else (5 if detection_count <= 10 else 10))
        return max(50, min(95, self.base_quality + boost))


def builtin_text_size(label: str, glyph_scale: int = 2) -> Tuple[Tuple[int, int], int]:
    """Metrics of the built-in 5x7 font at ``glyph_scale`` (6 columns per character): this module's stand-in for
    ``cv2.getTextSize(label, FONT_HERSHEY_SIMPLEX, 0.5, 2)``."""
    return (6 * glyph_scale * len(label), 7 * glyph_scale), glyph_scale


def plan_render(frame_wh: Tuple[int, int], track_list: Sequence[dict], quality: int, webp_available: bool = False,
                text_size: Callable[[str], Tuple[Tuple[int, int], int]] = builtin_text_size) -> List[list]:
    """The sequence of drawing / encode operations ``KafkaSink._render_frame`` performs (kafka_sink.py:218-289), in the
    form the golden recording uses: ["resize", [w, h], INTER_AREA], ["rectangle", p0, p1, bgr, thickness], ["text", label,
    org, 0.5, [255, 255, 255], 2], ["encode", ext, params, [h, w]]."""
    w, h = frame_wh
    ops: List[list] = []
    scale = 1.0
    if w > 1920 or h > 1080:
        scale = min(1920 / w, 1080 / h)
        w, h = int(w * scale), int(h * scale)
        ops.append(["resize", [w, h], INTER_AREA])
    for t in track_list:
        x1, y1, x2, y2 = [int(v * scale) for v in t["bbox_xyxy"]]
        color = list(color_for(t["class_id"]))
        ops.append(["rectangle", [x1, y1], [x2, y2], color, 2])
        label = f'ID {t["track_id"]}'
        (lw, lh), baseline = text_size(label)
        ops.append(["rectangle", [x1, max(0, y1 - lh - baseline - 4)], [x1 + lw, max(0, y1)], color, -1])
        ops.append(["text", label, [x1, max(0, y1 - 4)], 0.5, [255, 255, 255], 2])
    if webp_available and quality >= 80:
        ops.append(["encode", ".webp", [IMWRITE_WEBP_QUALITY, quality], [h, w]])
    else:
        ops.append(["encode", ".jpg", [IMWRITE_JPEG_QUALITY, quality, IMWRITE_JPEG_PROGRESSIVE, 1, IMWRITE_JPEG_OPTIMIZE, 1], [h, w]])
    return ops


def raster_primitives(ops: Sequence[list], size_wh: Tuple[int, int], glyph_scale: int = 2):
    """Turn a plan into what K6 draws: inclusive filled rectangles (painter's order) + glyph cells.  An outline of
    thickness 2 becomes four bars covering coordinates c-1 .. c of its edge (this module's rule; OpenCV's is unpinned)."""
    w, h = size_wh
    rects, colors, glyphs = [], [], []

    def add(x0, y0, x1, y1, col):
        x0, y0, x1, y1 = max(0, min(x0, x1)), max(0, min(y0, y1)), min(w - 1, max(x0, x1)), min(h - 1, max(y0, y1))
        if x0 <= x1 and y0 <= y1:
            rects.append([x0, y0, x1, y1]); colors.append([col[0], col[1], col[2], 0])
    for op in ops:
        if op[0] == "rectangle":
            (x0, y0), (x1, y1), col, th = op[1], op[2], op[3], op[4]
            if th < 0:
                add(x0, y0, x1, y1, col)
            else:
                t0 = th // 2                                   # thickness 2: rows / columns c-1 .. c
                add(x0 - t0, y0 - t0, x1 + (th - 1 - t0), y0 + (th - 1 - t0), col)
                add(x0 - t0, y1 - t0, x1 + (th - 1 - t0), y1 + (th - 1 - t0), col)
                add(x0 - t0, y0 - t0, x0 + (th - 1 - t0), y1 + (th - 1 - t0), col)
                add(x1 - t0, y0 - t0, x1 + (th - 1 - t0), y1 + (th - 1 - t0), col)
        elif op[0] == "text":
            label, (ox, oy) = op[1], op[2]
            top = oy - 7 * glyph_scale                        # putText's origin is the bottom-left of the text
            for k, ch in enumerate(label):
                glyphs.append([ox + 6 * glyph_scale * k, top, ord(ch)])
    return (np.asarray(rects, np.int32).reshape(-1, 4), np.asarray(colors, np.uint8).reshape(-1, 4),
            np.asarray(glyphs, np.int32).reshape(-1, 3))


def render_bgr(frame, ops: Sequence[list], glyph_scale: int = 2, ctx=None):
    """A plan without a resize drawn over a COPY of a BGR frame -- ``[h, w, 3]`` uint8, a host ndarray as cv2.VideoCapture
    delivers it (uploaded once) or a device tensor: K6 with ratio 0 draws in place on the copy."""
    import torch
    from . import _native as N
    from . import ops as O
    ctx = ctx or O.context()
    if any(op[0] == "resize" for op in ops):
        raise ValueError("render_bgr draws at the frame's own size; a resizing plan needs the NV12 surface (render_nv12)")
    src = torch.from_numpy(np.ascontiguousarray(frame)) if isinstance(frame, np.ndarray) else frame
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
        raise ValueError("render_bgr expects a [h, w, 3] uint8 BGR image")
    out = src.cuda().contiguous().clone() if src.is_cuda else src.cuda().contiguous()
    th, tw = int(out.shape[0]), int(out.shape[1])
    rects, colors, glyphs = raster_primitives(ops, (tw, th), glyph_scale)
    dev = out.device
    d_r = torch.from_numpy(rects).to(dev) if len(rects) else None
    d_c = torch.from_numpy(colors).to(dev) if len(rects) else None
    d_g = torch.from_numpy(glyphs).to(dev) if len(glyphs) else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    rc = N.lib().rva_preview_nv12(ctx.handle, None, None, 0, tw, th, 0, p(out), tw, th, p(d_r), p(d_c), len(rects), p(d_g), len(glyphs),
                                  glyph_scale, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ctx.check(rc, "rva_preview_nv12")
    return out


def render_nv12(surface, ops: Sequence[list], glyph_scale: int = 2, ctx=None):
    """The preview image of a plan for an NV12 surface in HBM: uint8 BGR ``[h, w, 3]`` device tensor (one K6 launch; a
    non-integer downscale goes through the INTER_LINEAR resize stage first, a documented deviation from INTER_AREA)."""
    import torch
    from . import _native as N
    from . import ops as O
    ctx = ctx or O.context()
    w, h = surface.width, surface.height
    tw, th = w, h
    for op in ops:
        if op[0] == "resize":
            tw, th = op[1]
    rects, colors, glyphs = raster_primitives(ops, (tw, th), glyph_scale)
    dev = surface.y.device
    out = torch.empty((th, tw, 3), dtype=torch.uint8, device=dev)
    ratio = 1
    if (tw, th) != (w, h):
        if w % tw == 0 and h % th == 0 and w // tw == h // th:
            ratio = w // tw                                   # 4K -> 1080p: the 2x2 box mean INTER_AREA computes
        else:
            out = O.resize_nv12_to_bgr([surface], (tw, th), ctx=ctx)[0].contiguous()
            ratio = 0
    d_r = torch.from_numpy(rects).to(dev) if len(rects) else None
    d_c = torch.from_numpy(colors).to(dev) if len(rects) else None
    d_g = torch.from_numpy(glyphs).to(dev) if len(glyphs) else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    rc = N.lib().rva_preview_nv12(ctx.handle, p(surface.y), p(surface.uv), surface.pitch, w, h, ratio, p(out), tw, th,
                                  p(d_r), p(d_c), len(rects), p(d_g), len(glyphs), glyph_scale,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ctx.check(rc, "rva_preview_nv12")
    return out


def encode_image(bgr: np.ndarray, ext: str, params: Sequence[int]) -> Tuple[bytes, str]:
    """Host encoder standing in for ``cv2.imencode`` (kafka_sink.py:260-284): Pillow, same quality / progressive /
    optimise settings.  Raises ``RuntimeError`` like the reference when no encoder is available."""
    try:
        from PIL import Image
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError("preview encoding needs Pillow (no device encoder is present in this image)") from exc
    kv = dict(zip(params[0::2], params[1::2]))
    img = Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]))                  # BGR -> RGB
    buf = io.BytesIO()
    if ext == ".webp":
        img.save(buf, format="WEBP", quality=int(kv.get(IMWRITE_WEBP_QUALITY, 75)))
        return buf.getvalue(), "image/webp"
    img.save(buf, format="JPEG", quality=int(kv.get(IMWRITE_JPEG_QUALITY, 75)), progressive=bool(kv.get(IMWRITE_JPEG_PROGRESSIVE, 0)),
             optimize=bool(kv.get(IMWRITE_JPEG_OPTIMIZE, 0)))
    return buf.getvalue(), "image/jpeg"


def render_frame(surface, track_list: Sequence[dict], quality: Optional[int] = None, policy: Optional[PreviewPolicy] = None,
                 ctx=None) -> str:
    """``KafkaSink._render_frame`` for a device surface: the ``frame_jpeg`` value of the wire message."""
    policy = policy or PreviewPolicy()
    q = policy.base_quality if quality is None else quality
    ops = plan_render((surface.width, surface.height), track_list, q, policy.webp_available)
    img = render_nv12(surface, ops, ctx=ctx)
    ext, params = next((op[1], op[2]) for op in ops if op[0] == "encode")
    if ext == ".jpg":                                      # encoded where it was rendered: only the stream crosses PCIe
        from . import ops as O
        kv = dict(zip(params[0::2], params[1::2]))
        data, mime = O.jpeg_encode_bgr(img, int(kv.get(IMWRITE_JPEG_QUALITY, 75)), ctx=ctx), "image/jpeg"
    else:
        data, mime = encode_image(img.cpu().numpy(), ext, params)
    return f"data:{mime};base64,{base64.b64encode(data).decode('ascii')}"


def attach_preview(payload: dict, surface, policy: PreviewPolicy, ctx=None) -> dict:
    """kafka_sink.py:134-146: add ``frame_jpeg`` to a tracks payload when the stream's rate limit allows it."""
    if surface is not None and policy.should_send_frame(payload["stream"]):
        payload["frame_jpeg"] = render_frame(surface, payload["tracks"], policy.adaptive_quality(len(payload["tracks"])), policy, ctx)
    return payload


# ---------------------------------------------------------------------------------------------- five-minute snapshots
SNAPSHOT_INTERVAL = 300.0                     # pipeline.py:269
SNAPSHOT_ROOT = "/data/outputs"               # pipeline.py:282
SNAPSHOT_COLOR = (0, 204, 255)                # pipeline.py:278, BGR
SNAPSHOT_JPEG_QUALITY = 95                    # cv2.imwrite's default for ".jpg" (no parameters are passed, pipeline.py:287)


def _field(track, name, default=None):
    return track.get(name, default) if isinstance(track, dict) else getattr(track, name, default)


def plan_snapshot(stream_name: str, frame_id: int, now: float, frame_wh: Tuple[int, int], track_list: Sequence,
                  root: str = SNAPSHOT_ROOT) -> List[list]:
    """What ``StreamWorker._maybe_save_snapshot`` does once a snapshot is due (pipeline.py:275-288), in the form of the golden
    recording (tests/golden/snapshot_plan.json): the frame is copied, every track gets a 2-px outline in one fixed colour
    and the label ``ID<track_id> cls<class_id>`` (``ID cls<class_id>`` for an object without a track id) 6 px above its top-left
    corner, the stream's directory is made and the image is written as ``<int(now)>_frame<frame_id>.jpg``.  Tracks are Track /
    Detection objects or wire dicts."""
    w, h = frame_wh
    ops: List[list] = [["copy", [h, w]]]
    for t in track_list:
        x1, y1, x2, y2 = [int(v) for v in _field(t, "bbox_xyxy")]
        ops.append(["rectangle", [x1, y1], [x2, y2], list(SNAPSHOT_COLOR), 2])
        tid = _field(t, "track_id")
        ops.append(["text", f"ID{'' if tid is None else tid} cls{_field(t, 'class_id')}", [x1, max(0, y1 - 6)], 0.5, [255, 255, 255], 1])
    out_dir = f"{root.rstrip('/')}/{stream_name}"
    ops.append(["mkdir", out_dir, True, True])
    ops.append(["imwrite", f"{out_dir}/{int(now)}_frame{frame_id}.jpg", [h, w]])
    return ops


class SnapshotWriter:
    """``StreamWorker._maybe_save_snapshot`` for every stream of a pipeline (the reference keeps ``_last_snapshot_ts`` per
    worker, one worker per stream; pipeline.py:93, 264-290): at most one annotated frame per stream per 300 s, drawn (K6) and
    JPEG-encoded (K7) on the device -- the finished file is the only thing that crosses PCIe.  Like the reference, a failure to
    write is logged, not raised."""

    def __init__(self, root: str = SNAPSHOT_ROOT, clock: Callable[[], float] = time.time, interval: float = SNAPSHOT_INTERVAL, ctx=None):
        self.root, self._clock, self.interval, self.ctx = root, clock, interval, ctx
        self._last: dict = {}
        self.written: List[str] = []

    def due(self, stream_name: str) -> Optional[float]:
        """The timestamp to file the snapshot under if one is due now (and note it as taken), else None."""
        now = self._clock()
        if now - self._last.get(stream_name, 0.0) < self.interval:
            return None
        self._last[stream_name] = now
        return now

    def maybe_save(self, packet, track_list: Sequence) -> Optional[str]:
        """``packet``: a FramePacket whose frame is an Nv12Surface in HBM or a host BGR ndarray.  Returns the path written."""
        import logging
        name = packet.stream.name
        now = self.due(name)
        if now is None:
            return None
        frame = packet.frame
        is_bgr = isinstance(frame, np.ndarray) or hasattr(frame, "dim")
        wh = (int(frame.shape[1]), int(frame.shape[0])) if is_bgr else (frame.width, frame.height)
        ops = plan_snapshot(name, packet.frame_id, now, wh, track_list, self.root)
        from . import ops as O
        img = render_bgr(frame, ops, ctx=self.ctx) if is_bgr else render_nv12(frame, ops, ctx=self.ctx)
        data = O.jpeg_encode_bgr(img, SNAPSHOT_JPEG_QUALITY, ctx=self.ctx)
        path = next(op[1] for op in ops if op[0] == "imwrite")
        try:
            import os
            os.makedirs(next(op[1] for op in ops if op[0] == "mkdir"), exist_ok=True)
            with open(path, "wb") as f:
                f.write(data)
        except Exception as exc:  # noqa: BLE001  pipeline.py:289-290
            logging.getLogger(__name__).error("Failed to save snapshot for '%s': %s", name, exc)
            return None
        logging.getLogger(__name__).info("Saved snapshot for '%s' to %s", name, path)
        self.written.append(path)
        return path

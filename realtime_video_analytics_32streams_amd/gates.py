"""Pre-detector frame gates: which frames reach the detector.

Host logic restated from the reference's StreamWorker (pipeline.py:96-116, 156-170, 242-262) with the
pixel work on the GPU:
  * :class:`MotionGate` -- ``MotionFilter.should_process`` (utils/frame_filter.py:26-40) for all streams of
    a tick in one K5 launch; the blurred-gray history stays in HBM (two buffers per stream, ping-pong).
  * :class:`AdaptiveFps` -- the idle-frame counter that thins detection to every ``ratio``-th frame
    (pipeline.py:107-116, 165-170, 242-262).
A gated-out frame is a *skipped* frame: ``tracker.update(name, [])`` semantics (pipeline.py:214-222),
which :class:`~.pipeline.TickPipeline` expresses as ``process[i] = False``.
  * :func:`rasterize_polygons` -- the mask of ``apply_roi`` (utils/frame_filter.py:43-50); the mask is applied on
    the GPU inside K1 / K5 (``Nv12Surface.mask``).  cv2.fillPoly's exact edge rule is unpinned (OpenCV absent):
    interior by the even-odd rule at integer pixel coordinates plus every pixel on a polygon edge; exact for
    axis-aligned rectangles, which fillPoly fills inclusively.
  * ``downsample_ratio`` (utils/frame_filter.py:53-57) is ``ops.resize_nv12_to_bgr`` + the per-stream box scale of
    ``DeviceTracker.set_box_scale`` (``_rescale_detections``, pipeline.py:224-240); wired in ``TickPipeline``.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _native as N
from . import ops
from .config import StreamConfig


def rasterize_polygons(polygons, width: int, height: int):
    """uint8 [height, width] mask, 255 inside any polygon (see module docstring for the edge rule)."""
    import numpy as np
    mask = np.zeros((height, width), np.uint8)
    ys, xs = np.mgrid[0:height, 0:width]
    for poly in polygons or []:
        pts = np.asarray(poly, np.int64)
        n = len(pts)
        if n == 0:
            continue
        inside = np.zeros((height, width), bool)
        edge = np.zeros((height, width), bool)
        for i in range(n):
            (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
            if y0 != y1:      # even-odd crossing test, half-open in y
                cond = ((y0 <= ys) & (ys < y1)) | ((y1 <= ys) & (ys < y0))
                t = (xs - x0) * (y1 - y0) - (x1 - x0) * (ys - y0)
                inside ^= cond & ((t < 0) == (y1 > y0))
            # pixels on the edge itself (fillPoly draws the outline)
            cross = (xs - x0) * (y1 - y0) - (ys - y0) * (x1 - x0)
            on = (np.abs(cross) * 2 <= max(abs(x1 - x0), abs(y1 - y0))) & (xs >= min(x0, x1)) & (xs <= max(x0, x1)) & \
                 (ys >= min(y0, y1)) & (ys <= max(y0, y1))
            edge |= on
        mask[inside | edge] = 255
    return mask


def motion_min_count(threshold: float, n_pixels: int) -> int:
    """Smallest changed-pixel count ``c`` with ``float(c) / float(n_pixels) >= threshold`` -- the reference's decision
    (utils/frame_filter.py:37-39) as an integer compare, exact because the quotient is monotonic in ``c``."""
    n = float(n_pixels)
    c = max(0, min(n_pixels + 1, int(threshold * n)))
    while c > 0 and float(c - 1) / n >= threshold:
        c -= 1
    while c <= n_pixels and not (float(c) / n >= threshold):
        c += 1
    return c


class MotionGate:
    """``MotionFilter.should_process`` for all streams of a tick.  Like the reference (one MotionFilter per stream at that
    stream's own frame shape, pipeline.py:156-158) every stream keeps its own geometry: it is learnt from the stream's first
    frame, the blurred-gray history is sized for it, and a later frame of another size or kind is an error.  One K5 launch
    per distinct geometry of the tick."""

    def __init__(self, n_streams: int, width: Optional[int] = None, height: Optional[int] = None,
                 thresholds: Sequence[float] = (), ctx: Optional[N.Context] = None):
        self.ctx = ctx or ops.context()
        self.dev = torch.device("cuda", self.ctx.device)
        self.n = n_streams
        self.thresholds = list(thresholds) if len(thresholds) else [0.02] * n_streams
        self.geom: List[Optional[tuple]] = [None] * n_streams        # (w, h, "nv12" | "bgr") per stream
        self._blur: List[Optional[torch.Tensor]] = [None] * n_streams
        self._have_prev = [False] * n_streams
        self._flip = [0] * n_streams
        # eight count rows: a pipelined caller rotates them per tick (tick k's K4 may still read its counts on another HIP
        # stream while K5 of the following ticks writes)
        self.counts = torch.zeros((8, n_streams), dtype=torch.int32, device=self.dev)
        if width is not None and height is not None:                 # geometry known up front (all streams alike)
            for i in range(n_streams):
                self._blur[i] = torch.empty((2, height, width), dtype=torch.uint8, device=self.dev)

    @property
    def w(self):
        return next((g[0] for g in self.geom if g), None)

    @property
    def h(self):
        return next((g[1] for g in self.geom if g), None)

    @staticmethod
    def _geom_of(f) -> tuple:
        if isinstance(f, ops.Nv12Surface):
            return (int(f.width), int(f.height), "nv12")
        if not (isinstance(f, torch.Tensor) and f.is_cuda and f.dtype == torch.uint8 and f.dim() == 3 and f.shape[2] == 3):
            raise ValueError("motion gate frames must be Nv12Surface or device uint8 BGR [h, w, 3] tensors")
        return (int(f.shape[1]), int(f.shape[0]), "bgr")

    def min_count(self, i: int) -> int:
        """Device-gate form of stream i's threshold (needs the stream's geometry, i.e. its first frame)."""
        g = self.geom[i]
        return motion_min_count(self.thresholds[i], g[0] * g[1]) if g else 0

    def launch(self, surfaces: Sequence, slot: int = 0) -> List[int]:
        """Enqueue K5 for the streams that delivered a frame (no host sync).  Returns, per stream, the row of
        ``self.counts[slot]`` that will hold its changed-pixel count (-1 on the first frame of a stream), or -1 for no frame."""
        if len(surfaces) != self.n:
            raise ValueError(f"MotionGate was built for {self.n} streams, got {len(surfaces)} frames")
        rows = [-1] * self.n
        groups = {}
        for i, f in enumerate(surfaces):
            if f is None:
                continue
            g = self._geom_of(f)
            if self.geom[i] is None:
                if self._blur[i] is not None and tuple(self._blur[i].shape[1:]) != (g[1], g[0]):
                    raise ValueError(f"motion gate stream {i}: frame is {g[0]}x{g[1]}, the gate was built for "
                                     f"{self._blur[i].shape[2]}x{self._blur[i].shape[1]}")
                self.geom[i] = g
                if self._blur[i] is None:
                    self._blur[i] = torch.empty((2, g[1], g[0]), dtype=torch.uint8, device=self.dev)
            elif self.geom[i] != g:
                raise ValueError(f"motion gate stream {i}: frame geometry changed from {self.geom[i]} to {g} "
                                 "(the history buffer of a stream has one size, like the reference's MotionFilter)")
            groups.setdefault(g, []).append(i)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        row0 = 0
        for (w, h, kind), idx in groups.items():
            for b0 in range(0, len(idx), N.RVA_MAX_BATCH):
                part = idx[b0:b0 + N.RVA_MAX_BATCH]
                prev, _d = N.ptr_array([self._blur[i][self._flip[i] ^ 1].data_ptr() if self._have_prev[i] else 0 for i in part])
                cur, _e = N.ptr_array([self._blur[i][self._flip[i]].data_ptr() for i in part])
                cnt = C.c_void_p(self.counts[slot].data_ptr() + 4 * row0)
                if kind == "nv12":
                    yp, _a = N.ptr_array([surfaces[i].y.data_ptr() for i in part])
                    up, _b = N.ptr_array([surfaces[i].uv.data_ptr() for i in part])
                    pp, _c = N.i32_array([surfaces[i].pitch for i in part])
                    mp, _m = N.ptr_array([surfaces[i].mask.data_ptr() if surfaces[i].mask is not None else 0 for i in part])
                    rc = N.lib().rva_motion_nv12_masked_batch(self.ctx.handle, yp, up, pp, mp, prev, cur, len(part), w, h, cnt, stream)
                else:
                    fp, _a = N.ptr_array([surfaces[i].data_ptr() for i in part])
                    rb, _b = N.i32_array([int(surfaces[i].stride(0)) for i in part])
                    rc = N.lib().rva_motion_bgr_batch(self.ctx.handle, fp, rb, prev, cur, len(part), w, h, cnt, stream)
                self.ctx.check(rc, "rva_motion_*_batch")
                for k, i in enumerate(part):
                    rows[i] = row0 + k
                    self._have_prev[i] = True
                    self._flip[i] ^= 1
                row0 += len(part)
        return rows

    def step(self, surfaces: Sequence) -> List[bool]:
        """``surfaces[i]`` is stream i's frame_for_detection of this tick: an ``Nv12Surface`` (its ``mask`` is
        honoured), a uint8 BGR device tensor [h, w, 3] (downsampled frame), or None (no frame).  Returns
        should_process per stream (True for a stream's first frame, frame_filter.py:33-35).  One host sync."""
        rows = self.launch(surfaces)
        out = [True] * len(surfaces)
        if all(r < 0 for r in rows):
            return out
        cnt = self.counts[0].cpu().tolist()
        for i, r in enumerate(rows):
            if r >= 0 and cnt[r] >= 0:                                  # -1: first frame of the stream
                g = self.geom[i]
                out[i] = (float(cnt[r]) / float(g[0] * g[1])) >= self.thresholds[i]
        return out


class AdaptiveFps:
    """pipeline.py:104-116 (setup), 165-170 (gate), 242-262 (state update) for one stream."""

    def __init__(self, stream: StreamConfig):
        self.enabled = bool(stream.adaptive_fps)
        self.frame_index = 0
        self.idle_frames = 0
        self.process_every = 1
        if self.enabled:
            target = stream.target_fps or 30.0
            min_fps = max(stream.min_target_fps, 1.0)
            self.max_process_every = max(1, int(round(target / min_fps)))
            self.idle_tolerance = max(int(stream.idle_frame_tolerance), 1)
        else:
            self.max_process_every, self.idle_tolerance = 1, 0

    def should_process(self) -> bool:
        """Call once per frame, before detection (pipeline.py:144, 165-170)."""
        self.frame_index += 1
        if self.enabled and self.process_every > 1:
            return (self.frame_index - 1) % self.process_every == 0
        return True

    def update(self, detections_count: int, tracks_count: int) -> None:
        """pipeline.py:242-262, called for processed AND skipped frames (skipped: detections_count = 0)."""
        if not self.enabled:
            return
        if detections_count > 0 or tracks_count > 0:
            self.idle_frames = 0
            self.process_every = 1
        else:
            self.idle_frames += 1
            if self.idle_frames >= self.idle_tolerance:
                self.process_every = max(self.max_process_every, 1)

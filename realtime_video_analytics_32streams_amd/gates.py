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


class MotionGate:
    def __init__(self, n_streams: int, width: int, height: int, thresholds: Sequence[float], ctx: Optional[N.Context] = None):
        self.ctx = ctx or ops.context()
        dev = torch.device("cuda", self.ctx.device)
        self.w, self.h, self.n = width, height, n_streams
        self.thresholds = list(thresholds)
        self._blur = [torch.empty((2, height, width), dtype=torch.uint8, device=dev) for _ in range(n_streams)]
        self._have_prev = [False] * n_streams
        self._flip = [0] * n_streams
        self.counts = torch.zeros(n_streams, dtype=torch.int32, device=dev)

    def step(self, surfaces: Sequence) -> List[bool]:
        """``surfaces[i]`` is stream i's frame_for_detection of this tick: an ``Nv12Surface`` (its ``mask`` is
        honoured), a uint8 BGR device tensor [h, w, 3] (downsampled frame), or None (no frame).  Returns
        should_process per stream (True for a stream's first frame, frame_filter.py:33-35).  One host sync."""
        idx = [i for i, s in enumerate(surfaces) if s is not None]
        out = [True] * len(surfaces)
        if not idx:
            return out
        prev, _d = N.ptr_array([self._blur[i][self._flip[i] ^ 1].data_ptr() if self._have_prev[i] else 0 for i in idx])
        cur, _e = N.ptr_array([self._blur[i][self._flip[i]].data_ptr() for i in idx])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if isinstance(surfaces[idx[0]], ops.Nv12Surface):
            yp, _a = N.ptr_array([surfaces[i].y.data_ptr() for i in idx])
            up, _b = N.ptr_array([surfaces[i].uv.data_ptr() for i in idx])
            pp, _c = N.i32_array([surfaces[i].pitch for i in idx])
            mp, _m = N.ptr_array([surfaces[i].mask.data_ptr() if surfaces[i].mask is not None else 0 for i in idx])
            rc = N.lib().rva_motion_nv12_masked_batch(self.ctx.handle, yp, up, pp, mp, prev, cur, len(idx), self.w, self.h,
                                                      C.c_void_p(self.counts.data_ptr()), stream)
        else:
            fp, _a = N.ptr_array([surfaces[i].data_ptr() for i in idx])
            rb, _b = N.i32_array([int(surfaces[i].stride(0)) for i in idx])
            rc = N.lib().rva_motion_bgr_batch(self.ctx.handle, fp, rb, prev, cur, len(idx), self.w, self.h,
                                              C.c_void_p(self.counts.data_ptr()), stream)
        self.ctx.check(rc, "rva_motion_*_batch")
        cnt = self.counts[:len(idx)].cpu().tolist()
        for k, i in enumerate(idx):
            if self._have_prev[i]:
                out[i] = (float(cnt[k]) / float(self.w * self.h)) >= self.thresholds[i]
            self._have_prev[i] = True
            self._flip[i] ^= 1
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

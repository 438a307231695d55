"""Pre-detector frame gates: which frames reach the detector.

Host logic restated from the reference's StreamWorker (pipeline.py:96-116, 156-170, 242-262) with the
pixel work on the GPU:
  * :class:`MotionGate` -- ``MotionFilter.should_process`` (utils/frame_filter.py:26-40) for all streams of
    a tick in one K5 launch; the blurred-gray history stays in HBM (two buffers per stream, ping-pong).
  * :class:`AdaptiveFps` -- the idle-frame counter that thins detection to every ``ratio``-th frame
    (pipeline.py:107-116, 165-170, 242-262).
A gated-out frame is a *skipped* frame: ``tracker.update(name, [])`` semantics (pipeline.py:214-222),
which :class:`~.pipeline.TickPipeline` expresses as ``process[i] = False``.
Not built this round: ROI polygon masks and ``downsample_ratio`` (the other two gates of 8f-2).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _native as N
from . import ops
from .config import StreamConfig


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

    def step(self, surfaces: Sequence[Optional[ops.Nv12Surface]]) -> List[bool]:
        """``surfaces[i]`` is stream i's frame of this tick (None: no frame).  Returns should_process per stream
        (True for the first frame of a stream, frame_filter.py:33-35).  One host sync (n int32)."""
        idx = [i for i, s in enumerate(surfaces) if s is not None]
        out = [True] * len(surfaces)
        if not idx:
            return out
        yp, _a = N.ptr_array([surfaces[i].y.data_ptr() for i in idx])
        up, _b = N.ptr_array([surfaces[i].uv.data_ptr() for i in idx])
        pp, _c = N.i32_array([surfaces[i].pitch for i in idx])
        prev, _d = N.ptr_array([self._blur[i][self._flip[i] ^ 1].data_ptr() if self._have_prev[i] else 0 for i in idx])
        cur, _e = N.ptr_array([self._blur[i][self._flip[i]].data_ptr() for i in idx])
        rc = N.lib().rva_motion_nv12_batch(self.ctx.handle, yp, up, pp, prev, cur, len(idx), self.w, self.h,
                                           C.c_void_p(self.counts.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
        self.ctx.check(rc, "rva_motion_nv12_batch")
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

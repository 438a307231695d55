"""Tick-batched orchestrator for the hot path.

The reference runs ``_process_packet`` once per frame per stream on one asyncio thread
(pipeline.py:143-212): detector.predict -> filter_detections -> tracker.update.  Here one *tick*
takes at most one frame from every stream of this GPU and runs the same order of operations for
all of them at once, without a host round trip between the stages:

    K1 pre-process (1 launch) -> detector network (PyTorch-ROCm) -> K2 decode + K3 NMS
    -> K4 tracker update (filter_detections fused in) -> id assignment -> one read-back.

Canonical order (SURVEY.md hard part 2): tick-major, streams in config order.  Skipped frames
(``process=False``: the motion / adaptive-fps gates of pipeline.py:156-170) age the stream's tracks
exactly like ``tracker.update(name, [])`` (pipeline.py:214-222).  A stream that delivers no frame in
a tick is masked out and does not stall the others.

Multi-GPU: each rank owns a contiguous slice of the streams; the only exchange is the per-tick
all-gather of ``n_streams`` int32 new-track counts (RCCL) feeding ``assign_ids`` -- see
:mod:`.dist`.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .config import PipelineConfig, StreamConfig
from .detector import HipYoloDetector, create_detector
from .gates import AdaptiveFps, MotionGate, rasterize_polygons
from .tracker import IouTracker, Track
from .video_stream import FramePacket, open_stream


@dataclass
class TickResult:
    tick: int
    tracks: Dict[str, List[Track]]          # per stream, every surviving track (tracker.py:95)
    detections_emitted: Dict[str, int]      # len(filtered) of pipeline.py:187 (after F1)
    latency_s: float


class StreamCounters:
    """The three per-stream counters the reference publishes (telemetry/metrics.py:55-72) as plain
    numbers; exporting them is out of scope."""

    def __init__(self):
        self.frames_total: Dict[str, int] = {}
        self.detections_total: Dict[str, int] = {}
        self.active_tracks: Dict[str, int] = {}

    def update(self, stream: str, frames: int, dets: int, tracks: int):
        self.frames_total[stream] = self.frames_total.get(stream, 0) + frames
        self.detections_total[stream] = self.detections_total.get(stream, 0) + dets
        self.active_tracks[stream] = tracks


class TickPipeline:
    """All streams of one GPU, one detector, one tracker."""

    def __init__(self, streams: Sequence[StreamConfig], detector: HipYoloDetector, tracker: IouTracker,
                 sources: Optional[Sequence] = None, id_sync=None, first_global_index: int = 0,
                 n_global_streams: Optional[int] = None):
        self.streams = list(streams)
        self.detector, self.tracker = detector, tracker
        self.sources = list(sources) if sources is not None else [open_stream(s, i) for i, s in enumerate(self.streams)]
        self.names = [s.name for s in self.streams]
        self.slots = tracker.register_streams(self.names)
        assert self.slots == sorted(self.slots), "streams must be registered in canonical (config) order"
        self.counters = StreamCounters()
        self.id_sync = id_sync
        self.global_index = [first_global_index + i for i in range(len(self.streams))]
        self.n_global = n_global_streams or len(self.streams)
        self._tick = 0
        # pre-detector gates, configured by the reference's StreamConfig keys
        self.adaptive = [AdaptiveFps(s) for s in self.streams]
        self._motion: Optional[MotionGate] = None
        self._motion_on = [bool(s.motion_filter) for s in self.streams]
        self._roi_masks: Dict[int, torch.Tensor] = {}          # stream index -> device mask (built at the first frame)
        ratios = {float(s.downsample_ratio) for s in self.streams}
        if len(ratios) > 1 and any(r < 0.999 for r in ratios):
            raise NotImplementedError("TickPipeline batches one geometry per tick: use the same downsample_ratio on all streams")
        self.downsample_ratio = ratios.pop() if ratios else 1.0
        if self.downsample_ratio < 0.999:                       # _rescale_detections, pipeline.py:224-240
            scale = 1.0 / max(self.downsample_ratio, 1e-6)
            tracker.device_tracker.set_box_scale([scale] * tracker.device_tracker.n_streams)

    def _frames_for_detection(self, packets: Sequence[Optional[FramePacket]]) -> List[Optional[FramePacket]]:
        """apply_roi -> downsample (pipeline.py:148-154): returns packets whose ``frame`` is what the reference
        calls ``frame_for_detection`` (masked surface, or the downsampled BGR image)."""
        need_roi = any(s.roi_polygons for s in self.streams)
        if not need_roi and self.downsample_ratio >= 0.999:
            return list(packets)
        out: List[Optional[FramePacket]] = []
        live = []
        for i, p in enumerate(packets):
            if p is None or not isinstance(p.frame, ops.Nv12Surface):
                out.append(p)
                continue
            f = p.frame
            if self.streams[i].roi_polygons:
                if i not in self._roi_masks:
                    m = rasterize_polygons(self.streams[i].roi_polygons, f.width, f.height)
                    self._roi_masks[i] = torch.from_numpy(m).to(f.y.device)
                f = ops.Nv12Surface(f.y, f.uv, f.width, f.height, mask=self._roi_masks[i])
            out.append(FramePacket(stream=p.stream, frame=f, frame_id=p.frame_id, timestamp=p.timestamp))
            live.append(i)
        if self.downsample_ratio < 0.999 and live:
            f0 = out[live[0]].frame
            dw, dh = int(f0.width * self.downsample_ratio), int(f0.height * self.downsample_ratio)
            small = ops.resize_nv12_to_bgr([out[i].frame for i in live], (dw, dh), ctx=self.detector.ctx)
            for k, i in enumerate(live):
                p = out[i]
                out[i] = FramePacket(stream=p.stream, frame=small[k], frame_id=p.frame_id, timestamp=p.timestamp)
        return out

    def _gate(self, packets: Sequence[Optional[FramePacket]]) -> List[bool]:
        """should-process decision per stream, in the reference's order: motion gate (pipeline.py:156-163), then
        adaptive-fps gate (:165-170).  ``AdaptiveFps.should_process`` runs for every packet (the frame index of
        pipeline.py:144 advances even when the motion gate already dropped the frame)."""
        n = len(packets)
        motion_ok = [True] * n
        if any(self._motion_on):
            surf = [p.frame if (p is not None and self._motion_on[i]) else None for i, p in enumerate(packets)]
            first = next((f for f in surf if f is not None), None)
            if first is not None:
                if self._motion is None:
                    fw, fh = (first.width, first.height) if isinstance(first, ops.Nv12Surface) else (int(first.shape[1]), int(first.shape[0]))
                    self._motion = MotionGate(n, fw, fh, [s.motion_threshold for s in self.streams], ctx=self.detector.ctx)
                motion_ok = self._motion.step(surf)
        out = []
        for i, p in enumerate(packets):
            if p is None:
                out.append(True)
                continue
            adaptive_ok = self.adaptive[i].should_process()
            out.append(motion_ok[i] and adaptive_ok)
        return out

    @classmethod
    def from_config(cls, cfg: PipelineConfig, **kw) -> "TickPipeline":
        streams = [s for s in cfg.streams if s.enabled]
        det = create_detector(cfg.detector_for(streams[0]))
        trk = IouTracker(cfg.tracker, max_streams=max(len(streams), 1))
        return cls(streams, det, trk, **kw)

    # the device part of a tick: no host synchronisation inside --------------------------------------
    def enqueue(self, packets: Sequence[Optional[FramePacket]], process: Optional[Sequence[bool]] = None):
        live = [(i, p) for i, p in enumerate(packets) if p is not None and (process is None or process[i])]
        n_s = self.tracker.device_tracker.n_streams
        slot_of_stream = [-1] * n_s
        post = None
        if live:
            post = self.detector.predict_batch_device([p for _, p in live])
            for row, (i, _) in enumerate(live):
                slot_of_stream[self.slots[i]] = row
        for i, p in enumerate(packets):
            if p is not None and process is not None and not process[i]:
                slot_of_stream[self.slots[i]] = -2       # skipped frame: ages the tracks
        dev = self.tracker.device_tracker
        dev.update_from_post(slot_of_stream, post, self.detector.config.confidence_threshold)
        if self.id_sync is None:
            dev.assign_ids()
        else:
            counts_all = self.id_sync.all_gather_counts(dev.new_counts_tensor()[:len(self.streams)])
            dev.assign_ids(counts_all, self.global_index + [0] * (n_s - len(self.streams)))
        return post

    def tick(self, process: Optional[Sequence[bool]] = None) -> TickResult:
        t0 = time.perf_counter()
        packets = self._frames_for_detection([src.next_packet() for src in self.sources])
        if process is None and (any(self._motion_on) or any(a.enabled for a in self.adaptive)):
            process = self._gate(packets)
        post = self.enqueue(packets, process)
        tables = self.tracker.device_tracker.read_all()          # the one host sync of the tick
        names = [n for n, p in zip(self.names, packets) if p is not None]
        tabs = [tables[self.slots[i]] for i, p in enumerate(packets) if p is not None]
        tracks = dict(zip(names, self.tracker.tracks_from_tables(names, tabs)))
        emitted = {}
        if post is not None:
            # len(filtered): detections that survive filter_detections (float64 compare on widened scores)
            counts = post.counts.cpu().numpy()
            thr = self.detector.config.confidence_threshold
            scores = post.scores[:, :max(int(counts.max()), 1)].double().cpu().numpy()
            row = 0
            for i, p in enumerate(packets):
                if p is None or (process is not None and not process[i]):
                    continue
                emitted[self.names[i]] = int((scores[row, :counts[row]] >= thr).sum())
                row += 1
        for n in names:
            self.counters.update(n, 1, emitted.get(n, 0), len(tracks[n]))
        for i, p in enumerate(packets):                     # pipeline.py:197 / :222 _adjust_adaptive_state
            if p is not None:
                self.adaptive[i].update(emitted.get(self.names[i], 0), len(tracks[self.names[i]]))
        self._tick += 1
        return TickResult(self._tick - 1, tracks, emitted, time.perf_counter() - t0)

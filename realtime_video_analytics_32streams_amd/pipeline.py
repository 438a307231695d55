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


class PipelinedTicks:
    """Throughput mode of :class:`TickPipeline`: ``depth`` ticks in flight, no host round trip inside a tick.

    Two HIP streams with fixed roles.  Stream A: K1 (eager) + the detector network; stream B: K2/K3 -> K4 -> global ids
    -> D2H snapshot of the track tables (slot = tick parity).  With ``use_graph`` the network and the part on B are
    replayed from captured hipGraphs (one per head-tensor parity; the fused plan never allocates or synchronises).  With
    sharded streams (``pipe.id_sync``) the RCCL exchange of new-track counts, ``k4_assign_ids`` and the snapshot follow
    B's graph eagerly.  B's work for tick k is released once K1 of tick k+1 is through, so the latency-bound tail hides
    under the next network and K1 runs alone.  Same results as ``TickPipeline.tick`` (same kernels, same order per
    stream); the pre-detector gates are host decisions per tick and are not supported here.

    ``submit()`` enqueues one tick and returns its ticket; ``collect()`` returns ``(ticket, tables)`` of the oldest
    outstanding tick, ``tables[slot]`` being the arrays of ``DeviceTracker.snapshot_fetch``.
    """

    def __init__(self, pipe: TickPipeline, depth: int = 2, use_graph: bool = True, overlap: bool = True):
        if any(pipe._motion_on) or any(a.enabled for a in pipe.adaptive) or pipe.downsample_ratio < 0.999 or \
                any(s.roi_polygons for s in pipe.streams):
            raise NotImplementedError("PipelinedTicks runs the ungated path (motion / adaptive-fps / ROI / downsample off)")
        if depth not in (1, 2):
            raise ValueError("depth must be 1 or 2 (two snapshot slots, two head tensors)")
        self.pipe, self.depth = pipe, depth
        self.det, self.dt = pipe.detector, pipe.tracker.device_tracker
        self.world_sharded = pipe.id_sync is not None
        self.slot = [-1] * self.dt.n_streams
        for i in range(len(pipe.streams)):
            self.slot[pipe.slots[i]] = i
        self.use_graph = bool(use_graph) and self.det.engine == "fused" and self.det.half
        # ``overlap=False`` keeps everything on one stream (eager only): the per-stage timing pass of bench.py
        self.two_streams = (overlap or self.use_graph) and self.det.engine == "fused" and self.det.half   # needs the second head tensor
        self.sA = torch.cuda.current_stream()
        self.sB = torch.cuda.Stream(device=self.det.device) if self.two_streams else self.sA
        self._pending = [None, None]          # eager mode: (raw, meta, events) of the tick whose stream-B part is due
        self._next, self._oldest = 0, 0
        self._done = [torch.cuda.Event(), torch.cuda.Event()]
        self._net_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._k1_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._net_graphs, self._post_graphs = [None, None], [None, None]
        self._posted = -1                     # last tick whose stream-B part has been issued
        self.last_post = None
        self._captured = False

    # -- pieces of a tick -------------------------------------------------------------------------------------
    def _post_part(self, raw, meta, events=None):
        with torch.inference_mode():
            post = self.det._postprocess_device(raw, [meta])
        if events: events[3].record()
        self.dt.update_from_post(self.slot, post, self.det.config.confidence_threshold)       # K4 (+F1 filter)
        self.last_post = post

    def _ids_and_snapshot(self, k, events=None):
        p = self.pipe
        if p.id_sync is None:
            self.dt.assign_ids()
        else:
            n = len(p.streams)
            self.dt.assign_ids(p.id_sync.all_gather_counts(self.dt.new_counts_tensor()[:n]), p.global_index)
        if events: events[4].record()
        self.dt.snapshot_async(k & 1)

    def _capture(self, frames):
        det = self.det
        with torch.inference_mode():
            tensor0, meta0 = det._preprocess(frames)
            det._infer(tensor0)                                    # builds + autotunes the plan outside any capture
        plan = det._plans[(int(tensor0.shape[0]), int(tensor0.shape[2]), int(tensor0.shape[3]))]
        torch.cuda.synchronize()
        raws = [None, None]
        for par in (0, 1):
            plan.use_output(par)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                with torch.inference_mode():
                    raws[par] = det._infer(tensor0)                # network only, writes head tensor `par`
            self._net_graphs[par] = g
        torch.cuda.synchronize()
        for par in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._post_part(raws[par], meta0)
                if not self.world_sharded:
                    self._ids_and_snapshot(par)                    # single GPU: ids + snapshot ride in the graph
            self._post_graphs[par] = g
        torch.cuda.synchronize()
        self._captured = True

    def _issue_post(self, k, after):
        par = k & 1
        with torch.cuda.stream(self.sB):
            self.sB.wait_event(self._net_done[par])
            if after is not None:
                self.sB.wait_event(after)
            if self.use_graph:
                self._post_graphs[par].replay()
                if self.world_sharded:
                    self._ids_and_snapshot(k)                      # the id exchange (RCCL) stays outside the graph
            else:
                raw, meta, events = self._pending[par]
                self._post_part(raw, meta, events)
                self._ids_and_snapshot(k, events)
            self._done[par].record(self.sB)
        self._posted = k

    # -- API ----------------------------------------------------------------------------------------------------
    def submit(self, packets: Optional[Sequence[FramePacket]] = None, events=None, before_k1=None) -> int:
        """Enqueue one tick.  ``events``: optional list of 5 timing events (before K1, after K1, after the network,
        after K2/K3, after ids) -- the last three are only recorded in the non-graph path.  ``before_k1``: optional
        callable run right before the K1 launch (bench.py arms the dispatch-level profiling events with it)."""
        if self._next - self._oldest >= self.depth:
            raise RuntimeError("collect() the oldest tick first")
        k = self._next
        par = k & 1
        if packets is None:
            packets = [src.next_packet() for src in self.pipe.sources]
        frames = [p.frame for p in packets]
        if self.use_graph and not self._captured:
            self._capture(frames)
        with torch.inference_mode():
            if events: events[0].record()
            if before_k1: before_k1()
            tensor, meta = self.det._preprocess(frames)            # K1
            if events: events[1].record()
        if self.two_streams:
            self._k1_done[par].record(self.sA)
            if k >= 2:
                self.sA.wait_event(self._done[par])                # tick k-2 has finished reading head tensor `par`
            if self.use_graph:
                self._net_graphs[par].replay()
            else:
                with torch.inference_mode():
                    plan = self.det._plans.get((int(tensor.shape[0]), int(tensor.shape[2]), int(tensor.shape[3])))
                    if plan is None:
                        self.det._infer(tensor)                    # builds + autotunes the plan
                        plan = self.det._plans[(int(tensor.shape[0]), int(tensor.shape[2]), int(tensor.shape[3]))]
                    plan.use_output(par)
                    raw = self.det._infer(tensor)
                if events: events[2].record()
                self._pending[par] = (raw, meta, events)
            self._net_done[par].record(self.sA)
            if self.depth == 1:
                self._issue_post(k, None)
            elif k >= 1 and self._posted < k - 1:
                self._issue_post(k - 1, self._k1_done[par])        # K1 of this tick first, then the previous tail
        else:
            with torch.inference_mode():
                raw = self.det._infer(tensor)
            if events: events[2].record()
            self._post_part(raw, meta, events)
            self._ids_and_snapshot(k, events)
        self._next += 1
        return k

    def collect(self):
        if self._oldest >= self._next:
            raise RuntimeError("nothing in flight")
        k = self._oldest
        if self.two_streams:
            if self._posted < k:
                self._issue_post(k, None)                          # no younger tick was submitted: release the tail now
            self._done[k & 1].synchronize()
            tables = self.dt.snapshot_fetch(k & 1, wait=False)
        else:
            tables = self.dt.snapshot_fetch(k & 1)                 # tracks visible to the host
        self._oldest += 1
        return k, tables

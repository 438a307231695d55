"""Tick-batched orchestrator for the hot path.

The reference runs ``_process_packet`` once per frame per stream on one asyncio thread
(pipeline.py:143-212): [roi -> downsample -> motion gate -> adaptive-fps gate] -> detector.predict ->
_rescale_detections -> filter_detections -> tracker.update.  Here one *tick* takes at most one frame from every
stream of this GPU and runs the same order of operations for all of them at once, without a host round trip
between the stages:

    K5 motion counts -> per frame group: K1 pre-process (1 launch) -> detector network -> K2 decode + K3 NMS
    -> K4 tracker update (gates, _rescale_detections and filter_detections fused in) -> id assignment -> one read-back.

A *frame group* is the set of streams of a tick that share a detector object and a frame geometry: every stream is
routed to the detector its ``detector_id`` names (one detector object per id, shared by its streams, as in
pipeline.py:470-489) and a tick with several resolutions issues one K1 / network / K2-K3 / K4 sequence per group.
All groups update the ONE shared tracker; ids are handed out after the last group, so they do not depend on the
grouping.  Detectors without a batched device path (the temporal heads, which return host detections) are called
per packet like the reference does and feed the tracker through its float64 entry point.

Canonical order (SURVEY.md hard part 2): tick-major, streams in config order.  Skipped frames (the motion /
adaptive-fps gates of pipeline.py:156-170) age the stream's tracks exactly like ``tracker.update(name, [])``
(pipeline.py:214-222).  A stream that delivers no frame in a tick is masked out and does not stall the others.

Gates: ``TickPipeline.tick`` decides them on the host (one small read-back of the K5 counts; gated-out frames never
reach the detector), ``PipelinedTicks`` decides them on the device inside K4 (no host sync; the detector has then run
on every delivered frame).  Both give the same tables.

Multi-GPU: each rank owns a contiguous slice of the streams; the only exchange is the per-tick all-gather of
``n_streams`` int32 new-track counts (RCCL) feeding ``assign_ids`` -- see :mod:`.dist`.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .config import PipelineConfig, StreamConfig
from .detector import create_detector, filter_detections
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


@dataclass
class _Group:
    """Streams of one tick that share a detector object and a frame geometry (one K1 launch, one network batch)."""
    det: int                 # index into TickPipeline.detectors
    key: tuple               # (w, h, kind) of the frames
    idx: List[int]           # stream indices, ascending (batch row r holds stream idx[r])


@dataclass
class _TickPlan:
    groups: List[_Group]
    host_idx: List[int]      # live streams whose detector has no batched device path (predict(packet) per stream)
    base: List[int]          # per tracker slot: -1 no frame / -2 skipped by the caller / -3 handled by a group or host path
    signature: tuple         # what a captured hipGraph of this tick shape depends on


class TickPipeline:
    """All streams of one GPU, their detectors, one tracker."""

    def __init__(self, streams: Sequence[StreamConfig], detector, tracker: IouTracker,
                 sources: Optional[Sequence] = None, id_sync=None, first_global_index: int = 0,
                 n_global_streams: Optional[int] = None):
        """``detector``: one detector object for every stream, or a sequence with one entry per stream (entries may repeat:
        streams that name the same ``detector_id`` share one object, pipeline.py:470-489)."""
        self.streams = list(streams)
        per_stream = list(detector) if isinstance(detector, (list, tuple)) else [detector] * len(self.streams)
        if len(per_stream) != len(self.streams):
            raise ValueError("one detector per stream is required")
        self.detectors: List = []
        self.det_of: List[int] = []
        for d in per_stream:
            k = next((j for j, e in enumerate(self.detectors) if e is d), None)
            if k is None:
                self.detectors.append(d)
                k = len(self.detectors) - 1
            self.det_of.append(k)
        self.detector = self.detectors[0]          # the common single-detector case keeps its old attribute
        self.tracker = tracker
        self.ctx = getattr(self.detector, "ctx", None) or ops.context()
        self.sources = list(sources) if sources is not None else [open_stream(s, i) for i, s in enumerate(self.streams)]
        self.names = [s.name for s in self.streams]
        self.slots = tracker.register_streams(self.names)
        assert self.slots == sorted(self.slots), "streams must be registered in canonical (config) order"
        self.counters = StreamCounters()
        self.id_sync = id_sync
        dev = tracker.device_tracker
        # rva_tracker_assign_ids reads one entry per TRACKER stream: pad the streams this pipeline does not use
        self.global_index = [0] * dev.n_streams
        for i, sl in enumerate(self.slots):
            self.global_index[sl] = first_global_index + i
        self.n_global = n_global_streams or len(self.streams)
        self._tick = 0
        self.snapshots = None          # a preview.SnapshotWriter: StreamWorker._maybe_save_snapshot (pipeline.py:196, 264-290)
        # pre-detector gates, configured by the reference's StreamConfig keys
        self.adaptive = [AdaptiveFps(s) for s in self.streams]
        self._motion: Optional[MotionGate] = None
        self._motion_on = [bool(s.motion_filter) for s in self.streams]
        self.has_gates = any(self._motion_on) or any(a.enabled for a in self.adaptive)
        # Streams with a gate (adaptive fps / motion) on a STATEFUL head: the reference never shows a skipped frame to the detector
        # (pipeline.py:156-181), so its clip buffer holds processed frames only.  Device-decided gates run every delivered frame
        # through stage_pre -- harmless for the stateless YOLO head, wrong for a clip ring -- so that combination keeps the gates
        # on the host (TickPipeline.tick) and is refused by the device-gate modes.
        self.gated_stateful = [self.names[i] for i in range(len(self.streams))
                               if (self._motion_on[i] or self.adaptive[i].enabled) and getattr(self.detectors[self.det_of[i]], "two_chain_ok", False)]
        # the temporal heads size their frame ring for all their streams now (a ring that grows mid-run would be swapped under
        # ticks still in flight on other streams)
        for j, d in enumerate(self.detectors):
            if hasattr(d, "reserve_streams"):
                d.reserve_streams([self.names[i] for i in range(len(self.streams)) if self.det_of[i] == j])
        self._gates_uploaded: Optional[tuple] = None
        self._roi_masks: Dict[int, torch.Tensor] = {}          # stream index -> device mask (built at the first frame)
        self.ratios = [float(s.downsample_ratio) for s in self.streams]
        if any(r < 0.999 for r in self.ratios):                 # _rescale_detections, pipeline.py:224-240
            scales = [1.0] * dev.n_streams
            for i, r in enumerate(self.ratios):
                if r < 0.999:
                    scales[self.slots[i]] = 1.0 / max(r, 1e-6)
            dev.set_box_scale(scales)

    @classmethod
    def from_config(cls, cfg: PipelineConfig, **kw) -> "TickPipeline":
        """One detector object per detector id (``__default__`` + ``detectors``), each stream routed by its
        ``detector_id`` (pipeline.py:470-489); only the ids some enabled stream uses are built."""
        streams = [s for s in cfg.streams if s.enabled]
        if not streams:
            raise RuntimeError("No stream workers started")       # pipeline.py:512-513
        built: Dict[str, object] = {}
        per_stream = []
        for s in streams:
            key = s.detector_id or "__default__"
            if key not in built:
                built[key] = create_detector(cfg.detector_for(s))
            per_stream.append(built[key])
        trk = IouTracker(cfg.tracker, max_streams=max(len(streams), 1))
        return cls(streams, per_stream, trk, **kw)

    # -- what reaches the detector ----------------------------------------------------------------------------
    def _frames_for_detection(self, packets: Sequence[Optional[FramePacket]]) -> List[Optional[FramePacket]]:
        """apply_roi -> downsample (pipeline.py:148-154): returns packets whose ``frame`` is what the reference
        calls ``frame_for_detection`` (masked surface, or the downsampled BGR image)."""
        need_roi = any(s.roi_polygons for s in self.streams)
        if not need_roi and all(r >= 0.999 for r in self.ratios):
            return list(packets)
        out: List[Optional[FramePacket]] = list(packets)
        by_size: Dict[tuple, List[int]] = {}
        for i, p in enumerate(packets):
            if p is None or not isinstance(p.frame, ops.Nv12Surface):
                continue
            f = p.frame
            if self.streams[i].roi_polygons:
                if i not in self._roi_masks:
                    m = rasterize_polygons(self.streams[i].roi_polygons, f.width, f.height)
                    self._roi_masks[i] = torch.from_numpy(m).to(f.y.device)
                f = ops.Nv12Surface(f.y, f.uv, f.width, f.height, mask=self._roi_masks[i])
                out[i] = FramePacket(stream=p.stream, frame=f, frame_id=p.frame_id, timestamp=p.timestamp)
            if self.ratios[i] < 0.999:
                dw, dh = int(f.width * self.ratios[i]), int(f.height * self.ratios[i])
                by_size.setdefault((f.width, f.height, dw, dh), []).append(i)
        for (_, _, dw, dh), idx in by_size.items():               # one resize launch per (source, target) geometry
            small = ops.resize_nv12_to_bgr([out[i].frame for i in idx], (dw, dh), ctx=self.ctx)
            for k, i in enumerate(idx):
                p = out[i]
                out[i] = FramePacket(stream=p.stream, frame=small[k], frame_id=p.frame_id, timestamp=p.timestamp)
        return out

    def _motion_gate(self) -> MotionGate:
        if self._motion is None:
            self._motion = MotionGate(len(self.streams), thresholds=[s.motion_threshold for s in self.streams], ctx=self.ctx)
        return self._motion

    def _gate(self, packets: Sequence[Optional[FramePacket]]) -> List[bool]:
        """Host form of the gates: should-process per stream, in the reference's order: motion gate (pipeline.py:156-163),
        then adaptive-fps gate (:165-170).  ``AdaptiveFps.should_process`` runs for every packet (the frame index of
        pipeline.py:144 advances even when the motion gate already dropped the frame).  One small host sync."""
        n = len(packets)
        motion_ok = [True] * n
        if any(self._motion_on):
            surf = [p.frame if (p is not None and self._motion_on[i]) else None for i, p in enumerate(packets)]
            if any(f is not None for f in surf):
                motion_ok = self._motion_gate().step(surf)
        out = []
        for i, p in enumerate(packets):
            if p is None:
                out.append(True)
                continue
            adaptive_ok = self.adaptive[i].should_process()
            out.append(motion_ok[i] and adaptive_ok)
        return out

    def _device_gates(self, packets: Sequence[Optional[FramePacket]], slot: int = 0) -> Optional[Tuple[torch.Tensor, List[int]]]:
        """Device form: enqueue K5 (no sync; counts go to row ``slot`` of the gate's two count buffers) and make sure the
        tracker holds the gate parameters.  Returns the ``motion`` argument of ``DeviceTracker.update_from_post`` (None
        when no stream has a motion gate)."""
        dev = self.tracker.device_tracker
        motion = None
        mg = None
        if any(self._motion_on):
            surf = [p.frame if (p is not None and self._motion_on[i]) else None for i, p in enumerate(packets)]
            mg = self._motion_gate()
            rows = mg.launch(surf, slot)
            full = [-1] * dev.n_streams
            for i, r in enumerate(rows):
                full[self.slots[i]] = r
            motion = (mg.counts[slot], full)
        key = tuple(mg.min_count(i) if (mg and self._motion_on[i]) else 0 for i in range(len(self.streams)))
        if self._gates_uploaded != key:           # first tick, or a motion-gated stream just showed its geometry
            en, mx, tol, mc = ([0] * dev.n_streams for _ in range(4))
            for i, a in enumerate(self.adaptive):
                sl = self.slots[i]
                en[sl], mx[sl], tol[sl], mc[sl] = int(a.enabled), a.max_process_every, a.idle_tolerance, key[i]
            dev.set_gates(en, mx, tol, mc, reset=self._gates_uploaded is None)
            self._gates_uploaded = key
        return motion

    # -- the shape of a tick ----------------------------------------------------------------------------------
    def plan_tick(self, packets: Sequence[Optional[FramePacket]], process: Optional[Sequence[bool]] = None) -> _TickPlan:
        n_s = self.tracker.device_tracker.n_streams
        base = [-1] * n_s
        groups: Dict[tuple, _Group] = {}
        host_idx: List[int] = []
        for i, p in enumerate(packets):
            if p is None:
                continue
            if process is not None and not process[i]:
                base[self.slots[i]] = -2                           # skipped frame: ages the tracks
                continue
            base[self.slots[i]] = -3
            d = self.detectors[self.det_of[i]]
            if hasattr(d, "predict_batch_device"):
                key = (self.det_of[i],) + d.geometry_key(p.frame)
                g = groups.get(key)
                if g is None:
                    g = groups[key] = _Group(self.det_of[i], key[1:], [])
                g.idx.append(i)
            else:
                host_idx.append(i)
        gl = list(groups.values())
        sig = (tuple((g.det, g.key, tuple(g.idx)) for g in gl), tuple(host_idx), tuple(base))
        return _TickPlan(gl, host_idx, base, sig)

    def _kslot(self, plan: _TickPlan, g: Optional[_Group], first: bool) -> List[int]:
        """slot_of_stream of one K4 launch: the group's streams map to their batch rows, the FIRST launch of a tick also
        carries the no-frame (-1) and skipped (-2) streams, everything else belongs to another launch (-3)."""
        ks = list(plan.base) if first else [-3] * len(plan.base)
        if g is not None:
            for row, i in enumerate(g.idx):
                ks[self.slots[i]] = row
        return ks

    def _host_path(self, plan: _TickPlan, packets) -> None:
        """Detectors that return host detections (temporal heads): predict(packet) -> _rescale_detections ->
        filter_detections per stream, as the reference does (pipeline.py:179-182), then one float64 tracker launch."""
        if not plan.host_idx:
            return
        per = {}
        self._host_dets = getattr(self, "_host_dets", {})
        for i in plan.host_idx:
            det = self.detectors[self.det_of[i]]
            dets = det.predict(packets[i])
            if self.ratios[i] < 0.999:
                sc = 1.0 / max(self.ratios[i], 1e-6)
                for d in dets:
                    x1, y1, x2, y2 = d.bbox_xyxy
                    d.bbox_xyxy = (x1 * sc, y1 * sc, x2 * sc, y2 * sc)
            kept = filter_detections(dets, det.config.confidence_threshold)
            self._host_dets[i] = kept
            per[self.slots[i]] = IouTracker._arrays(kept)
        self.tracker.device_tracker.update_from_host(per, others_untouched=True)

    # the device part of a tick: no host synchronisation inside (batched detectors) ---------------------------
    def enqueue(self, packets: Sequence[Optional[FramePacket]], process: Optional[Sequence[bool]] = None,
                device_gates: bool = False):
        plan = self.plan_tick(packets, process)
        dev = self.tracker.device_tracker
        gated = device_gates and self.has_gates
        motion = self._device_gates(packets) if gated else None
        posts = []
        first = True
        for g in plan.groups:
            det = self.detectors[g.det]
            post = det.predict_batch_device([packets[i] for i in g.idx])
            self._note_clips(g, getattr(det, "last_clip_infos", None))
            dev.update_from_post(self._kslot(plan, g, first), post, det.config.confidence_threshold, gated=gated, motion=motion)
            posts.append(post)
            first = False
        if first:                                                  # no batched group: the -1 / -2 streams still need their launch
            dev.update_from_post(self._kslot(plan, None, True), None, 0.0, gated=gated, motion=motion)
        self._host_path(plan, packets)
        self.assign_ids()
        return plan, posts

    def _note_clips(self, g: _Group, infos) -> None:
        """Temporal heads on the device path: remember which clip each stream fired (``finish`` copies the temporal
        fields into the tracks the clip's detections touched, tracker.py:58-67)."""
        if infos:
            self._host_dets = getattr(self, "_host_dets", {})
            for i in g.idx:
                if self.names[i] in infos:
                    self._host_dets[i] = infos[self.names[i]]

    def assign_ids(self) -> None:
        dev = self.tracker.device_tracker
        if self.id_sync is None:
            dev.assign_ids()
        else:
            n = len(self.streams)
            dev.assign_ids(self.id_sync.all_gather_counts(dev.new_counts_tensor()[:n]), self.global_index)

    def finish(self, packets: Sequence[Optional[FramePacket]], tables: Sequence[dict], status, t0: float) -> TickResult:
        """Host bookkeeping once a tick's snapshot is on the host: Track objects, counters, overflow flags, and (host-gated
        mode) the adaptive-fps state."""
        emitted_all, processed_all, flags = status
        ops.DeviceTracker.raise_on_flags(flags)
        live = [i for i, p in enumerate(packets) if p is not None]
        names = [self.names[i] for i in live]
        tabs = [tables[self.slots[i]] for i in live]
        host_dets = getattr(self, "_host_dets", {})
        tracks = {}
        for i, n, t in zip(live, names, tabs):
            tracks[n] = self.tracker._materialise(n, t, host_dets.get(i, ()))
        host_dets.clear()
        emitted = {self.names[i]: int(emitted_all[self.slots[i]]) for i in live if processed_all[self.slots[i]] == 1}
        if self.snapshots is not None:                                # processed frames only, once per stream per five minutes; the
            for i, n in zip(live, names):                             # frame's surface must still be alive when its tick is collected
                if n in emitted:
                    self.snapshots.maybe_save(packets[i], tracks[n])
        for n in names:
            self.counters.update(n, 1, emitted.get(n, 0), len(tracks[n]))
        self._tick += 1
        return TickResult(self._tick - 1, tracks, emitted, time.perf_counter() - t0)

    def tick(self, process: Optional[Sequence[bool]] = None, device_gates: bool = False) -> TickResult:
        """One synchronous tick.  Gates: decided on the host by default (gated-out frames never reach the detector);
        ``device_gates=True`` decides them inside K4 like :class:`PipelinedTicks` does (every delivered frame runs through
        the detector; same decisions, and bit-identical detector batches with the pipelined mode).  Use one of the two
        modes for the life of a pipeline: each keeps its own adaptive-fps state."""
        t0 = time.perf_counter()
        if device_gates and self.gated_stateful:
            raise NotImplementedError(f"device-decided gates on temporal streams {self.gated_stateful}: a skipped frame must not enter "
                                      "the clip buffer (pipeline.py:156-181); use tick() with host-decided gates for them")
        packets = self._frames_for_detection([src.next_packet() for src in self.sources])
        if process is None and self.has_gates and not device_gates:
            process = self._gate(packets)
        self.enqueue(packets, process, device_gates=device_gates)
        dev = self.tracker.device_tracker
        tables = dev.read_all()                                    # the one host sync of the tick (snapshot slot 0)
        res = self.finish(packets, tables, dev.snapshot_status(0), t0)
        if not device_gates:
            for i, p in enumerate(packets):                        # pipeline.py:197 / :222 _adjust_adaptive_state
                if p is not None:
                    self.adaptive[i].update(res.detections_emitted.get(self.names[i], 0), len(res.tracks[self.names[i]]))
        return res


class PipelinedTicks:
    """Throughput mode of :class:`TickPipeline`: ``depth`` ticks in flight, no host round trip inside a tick.

    Default layout (``depth`` >= 2): a tick is ONE chain on one HIP stream --

        roi / downsample / K5 motion counts -> per frame group K1 -> detector network (launched eagerly) ->
        K2/K3 -> K4 (gates decided on the device) -> global ids -> D2H snapshot (slot = tick mod depth)

    -- and consecutive ticks rotate over ``depth`` streams (``ops.chain_streams``: one set per process, the same streams for
    every runner), so the forward pass of tick k+1 (stem, 160x160 / 80x80 layers) fills the CUs the tail of tick k's pass
    (20x20 layers, detect branches) leaves idle.  ``bench.py`` runs three chains (two: -3..7 % frames/s, four: slower than
    two).  Each slot owns an input tensor, a fused plan (activation buffers), a head tensor and a snapshot slot.  What
    orders the chains, by events: K1(k) after K1(k-1) (gate state, source rings); network(k) after the tail of tick k-depth
    (head tensor and snapshot slot of this slot); tail(k) after tail(k-1) (tracker tables and the shared K2/K3 scratch are
    touched in tick order); ``collect(k)`` on the tail's event.  With ``use_graph`` the tail (K2/K3 -> K4 [-> ids ->
    snapshot]) is replayed from a hipGraph captured per slot; a tick shape (which streams are live, their grouping) is
    captured after it has run eagerly once -- that tick sizes every buffer and sets every kernel attribute outside any
    capture -- and ticks of another shape run eagerly.  With sharded streams (``pipe.id_sync``) the exchange of new-track
    counts (RCCL), ``k4_assign_ids`` and the snapshot follow the graph eagerly on the same stream.

    ``net_streams=1`` (or ``RVA_NET_STREAMS=1``; two ticks in flight at most) is the older layout: stream A carries K1 + the
    network with the detect branches forked onto side streams, stream B the tails, released once the next tick's K1 is
    through.  ``depth=1`` runs
    strictly one tick at a time.  Same results as ``TickPipeline.tick`` in every layout (same kernels, same order per
    stream).  Detectors without a batched device path need the host in the loop and are not supported here.

    ``submit()`` enqueues one tick and returns its ticket; ``collect()`` returns ``(ticket, tables)`` of the oldest
    outstanding tick, ``tables[slot]`` being the arrays of ``DeviceTracker.snapshot_fetch``; ``collect_result()`` returns
    the :class:`TickResult` instead.  Both raise if the device reported an overflow (tracker capacity / NMS capacity).
    """

    def __init__(self, pipe: TickPipeline, depth: Optional[int] = None, use_graph: bool = True, overlap: bool = True,
                 net_graph: bool = False, net_streams: int = 2):
        if any(not hasattr(d, "stage_pre") for d in pipe.detectors):
            raise NotImplementedError("PipelinedTicks needs detectors with a batched device path (stage_pre / stage_net / "
                                      "stage_post); a host-only detector runs through TickPipeline.tick")
        if pipe.gated_stateful:
            raise NotImplementedError(f"PipelinedTicks decides gates on the device, after every delivered frame has been pre-processed: "
                                      f"streams {pipe.gated_stateful} combine a gate (adaptive_fps / motion_filter) with a temporal head, "
                                      "whose clip buffer must only see processed frames (pipeline.py:156-181) -- run them through "
                                      "TickPipeline.tick() (host-decided gates)")
        if depth is None:
            # Four chains on one GPU, on streams probed to have a hardware lane each (ops.chain_streams): +2.3 % frames/s over
            # three at 32 x YOLOv8s, +11 % at 4 x YOLOv8m (round 4; five are slower than three).  With sharded streams
            # (pipe.id_sync) two: a collective that lands on a stream of its own takes a lane too -- measured with a stand-in on
            # one GPU in round 3: -15 % with three chains, 0 % with two (profiles/r03_experiments_not_kept.txt #14) -- and no 8-GPU
            # record exists yet that shows where RCCL puts it; 3 and 4 stay available (tests/test_gpu_multirank.py holds 3 to the oracle).
            depth = 4 if pipe.id_sync is None else 2
        if depth not in range(1, 9):
            raise ValueError("depth must be 1 .. 8 (snapshot slots of the tracker, motion-count rows of the gate)")
        self.pipe, self.depth = pipe, depth
        self.nslots = max(depth, 2)           # a tick's slot ("parity") = k mod nslots: buffers, events, snapshot slot
        self.det, self.dt = pipe.detector, pipe.tracker.device_tracker
        self.world_sharded = pipe.id_sync is not None
        is_fused = lambda d: getattr(d, "engine", None) == "fused" and d.half and d._infer_fn is None      # noqa: E731
        fused = all(is_fused(d) for d in pipe.detectors)
        # temporal heads run their (torch) network eagerly; they may still run as two chains: every result buffer of theirs
        # exists per tick parity and their frame ring has two slots more than a clip needs
        chains_ok = all(is_fused(d) or getattr(d, "two_chain_ok", False) for d in pipe.detectors)
        self.use_graph = bool(use_graph) and fused
        self._fused = fused
        # The network itself is launched eagerly by default: its plan forks the detect branches onto side streams, which
        # run concurrently when launched eagerly but are serialised by the hipGraph executor of this ROCm (measured:
        # 17.2 k vs 16.0 k frames/s); ~75 launches per tick cost the host ~0.3 ms of a 1.9 ms tick.  The latency-bound
        # tail on stream B (K2/K3 -> K4 -> ids -> snapshot) is what the captured graph is for.
        self.net_graph = bool(net_graph) and self.use_graph
        # with net_graph the network of a captured tick shape exists both ways; `replay_net` picks per tick (a caller may time both:
        # the graph saves ~0.35 ms of host time per tick and costs the GPU 2-5 % -- it wins where the host is the limit, YOLOv8n x 4)
        self.replay_net = self.net_graph
        # ``overlap=False`` keeps everything on one stream (eager only): the per-stage timing pass of bench.py
        self.two_streams = (overlap or self.use_graph) and chains_ok         # needs per-parity result buffers
        self.sA = torch.cuda.current_stream()
        # The networks of consecutive ticks run on TWO streams (even ticks on A, odd ticks on A'), each with its own input
        # tensor and fused plan: the tail of a forward pass (20x20 layers, detect branches: half the CUs idle) overlaps the
        # head of the next one (stem, 160x160 / 80x80 layers).  Measured with the plan alone: 1.73 -> 1.54 ms per 32-frame
        # forward pass (tools/two_streams.py).  A tick's K1 still waits for the previous tick's K1 / K5 (gate state, source
        # rings), and its network for the tick two before it to have released the head tensors of its parity.
        import os
        net_streams = int(os.environ.get("RVA_NET_STREAMS", net_streams))      # A/B switch for measurements
        self.net_streams = depth if (net_streams >= 2 and self.two_streams and depth >= 2) else 1
        if depth > 2 and self.net_streams == 1:
            # more than two ticks in flight need one chain per tick (fused or two_chain_ok detectors, RVA_NET_STREAMS unset);
            # the one-network-stream layout has two head tensors per plan: two ticks in flight
            self.depth = depth = 2
            self.nslots = 2
        # Two explicit streams created back to back (the runtime spreads consecutive streams over its hardware queues; the
        # caller's stream may be the null stream, whose queue another stream can share -- then the two networks would
        # serialise: 17.9 k instead of 20.3 k frames/s).  In this mode the plans run their detect branches in line: with two
        # forward passes in flight the side streams add nothing and, spread over more hardware queues, cost up to 25 %
        # (round-2 A/B, DESIGN.md section 4 item 11: 1.53-1.55 ms per pass in line for 3-16 queues, 1.68-1.82 ms with side streams).
        # The streams come from the process-wide pool (ops.chain_streams): the same ones the kernel selection timed its
        # overlapping passes on, and the same ones for every runner of the process.
        if self.net_streams >= 2:
            self.sAs = ops.chain_streams(self.det.device, self.nslots)
            self.sB = self.sAs[0]                                  # (tails ride on their tick's own stream in this layout)
        else:
            self.sAs = [self.sA] * self.nslots
            self.sB = torch.cuda.Stream(device=self.det.device) if self.two_streams else self.sA
        # K1 of tick k+1 beside the 20x20 phase of tick k's forward pass (most CUs and most of the HBM bandwidth idle there)
        # instead of beside its stem / 80x80 layers: RVA_K1_GATE=1.  Off by default: see DESIGN.md (K1 in the pipeline).
        self.k1_gate = os.environ.get("RVA_K1_GATE", "0") == "1" and self.net_streams >= 2
        self.k1_prio = os.environ.get("RVA_K1_PRIO", "0") == "1" and self.net_streams >= 2
        if self.k1_prio:
            self._k1_stream = torch.cuda.Stream(device=self.det.device, priority=-1)
        # kernel selection objective of plans built from now on: launches timed `net_streams` at a time (see FusedYoloV8.autotune)
        for d in pipe.detectors:
            if is_fused(d) and not getattr(d, "_plans", None):
                d.tune_overlap = max(self.net_streams, 1)
        ns = self.nslots
        self._phase_ev = [torch.cuda.Event() for _ in range(ns)]
        self._pending = [None] * ns           # per slot: what the stream-B part of that tick needs
        self._meta = [None] * ns              # per slot: (packets, t0) for collect_result
        self._next, self._oldest = 0, 0
        self._done = [torch.cuda.Event() for _ in range(ns)]
        self._net_done = [torch.cuda.Event() for _ in range(ns)]
        self._k1_done = [torch.cuda.Event() for _ in range(ns)]
        self._posted = -1                     # last tick whose stream-B part has been issued
        self.last_post = None
        self._seen_sigs = set()               # tick shapes that have run eagerly once
        self._cap_sig = None                  # the captured shape
        self._net_graphs: List[List] = []     # [group][parity]
        self._post_graphs = [None] * ns
        self._captured = False

    # -- pieces of a tick -------------------------------------------------------------------------------------
    def _plan_of(self, det, tensor):
        plan = det.plan_for(tensor)                                # builds + autotunes the plan of det's current slot (outside any capture)
        # detect branches in line when two passes overlap (see __init__), on side streams when one pass runs at a time
        plan.concurrent_heads = os.environ.get("RVA_SERIAL_HEADS") != "1" and \
            (self.net_streams < 2 or os.environ.get("RVA_NET2_LANES", "0") == "1")
        return plan

    def _set_slot(self, par):
        for d in self.pipe.detectors:
            d._slot = par if self.net_streams >= 2 else 0

    def _post_part(self, plan, raws, pres, motion, events=None):
        """The tail of one tick for every group: K2/K3 (temporal heads: top-5) then K4 (+ filter, rescale, gates).
        ``pres[g]``: what the group's ``stage_pre`` returned (YOLO: input tensor + letterbox meta)."""
        p = self.pipe
        gated = p.has_gates
        first = True
        for gi, g in enumerate(plan.groups):
            det = p.detectors[g.det]
            with torch.inference_mode():
                post = det.stage_post(raws[gi], pres[gi])
            if events and first: events[3].record()
            self.dt.update_from_post(p._kslot(plan, g, first), post, det.config.confidence_threshold, gated=gated, motion=motion)
            self.last_post = post
            first = False
        if first:
            self.dt.update_from_post(p._kslot(plan, None, True), None, 0.0, gated=gated, motion=motion)

    def _ids_and_snapshot(self, k, events=None):
        self.pipe.assign_ids()
        if events: events[4].record()
        self.dt.snapshot_async(k % self.nslots)

    def _capture(self, plan, pres, motion):
        """Record the networks (per group and head-tensor parity) and the stream-B part (per parity) of a tick of this
        shape.  Nothing executes here; every kernel of the shape has already run eagerly once."""
        p = self.pipe
        torch.cuda.synchronize()
        self._net_graphs = []
        raws = [[None] * self.nslots for _ in plan.groups]
        for gi, g in enumerate(plan.groups):
            det = p.detectors[g.det]
            pair = []
            for par in range(self.nslots):
                self._set_slot(par)
                # the slot's own input tensor (tick chains: one per slot; it exists once the slot has run, else it is made here)
                pre_par = pres[gi]
                if self.net_streams >= 2 and self.net_graph:
                    with torch.inference_mode():
                        pre_par = (det.input_tensor(int(pres[gi][0].shape[0])),) + tuple(pres[gi][1:])
                fp = self._plan_of(det, pre_par[0])
                # head tensor (group, parity): a stable buffer of the plan (two network streams: of the parity's own plan)
                raws[gi][par] = fp.use_output(gi if self.net_streams >= 2 else 2 * gi + par)
                if self.net_graph:
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        with torch.inference_mode():
                            det.stage_net(pre_par)                 # network only, writes that head tensor
                    pair.append(gr)
            self._net_graphs.append(pair)
        torch.cuda.synchronize()
        for par in range(self.nslots):
            mo = None if motion is None else (p._motion_gate().counts[par], motion[1])   # the parity's K5 count row
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                self._post_part(plan, [r[par] for r in raws], pres, mo)
                if not self.world_sharded:
                    self._ids_and_snapshot(par)                    # single GPU: ids + snapshot ride in the graph
            self._post_graphs[par] = gr
        torch.cuda.synchronize()
        self._set_slot(0)
        self._cap_sig = plan.signature
        self._captured = True

    def _issue_post(self, k, after, stream=None):
        par, prev = k % self.nslots, (k - 1) % self.nslots
        sb = stream if stream is not None else self.sB
        with torch.cuda.stream(sb), ops.roctx(f"tick{k}:tail"):
            sb.wait_event(self._net_done[par])
            if stream is not None and k >= 1:
                sb.wait_event(self._done[prev])                    # tails on several streams: tracker state is touched in tick order
            if after is not None:
                sb.wait_event(after)
            mode, plan, raws, pres, motion, events = self._pending[par]
            if mode == "graph":
                self._post_graphs[par].replay()
                if self.world_sharded:
                    self._ids_and_snapshot(k)                      # the id exchange (RCCL) stays outside the graph
            else:
                self._post_part(plan, raws, pres, motion, events)
                self._ids_and_snapshot(k, events)
            self._done[par].record(sb)
        self._posted = k

    # -- API ----------------------------------------------------------------------------------------------------
    def submit(self, packets: Optional[Sequence[Optional[FramePacket]]] = None, events=None, before_k1=None,
               process: Optional[Sequence[bool]] = None) -> int:
        """Enqueue one tick.  ``process``: optional per-stream decisions of the caller (False = skipped frame) on top of
        the configured gates.  ``events``: optional list of 5 timing events (before K1, after K1, after the network, after
        K2/K3, after ids) -- the last three are only recorded in the non-graph path.  ``before_k1``: optional callable run
        right before the first K1 launch (bench.py arms the dispatch-level profiling events with it)."""
        if self._next - self._oldest >= self.depth:
            raise RuntimeError("collect() the oldest tick first")
        p = self.pipe
        k = self._next
        par, prev = k % self.nslots, (k - 1) % self.nslots
        t0 = time.perf_counter()
        if packets is None:
            packets = [src.next_packet() for src in p.sources]
        sa = self.sAs[par]
        sk = sa                                                    # roi / downsample / K5 / K1 ride on the tick's own stream (a separate,
                                                                   # even high-priority, stream for them cost 20 % of the throughput)
        if self.k1_prio and self.net_streams >= 2:                 # experiment (RVA_K1_PRIO=1): K1 on ONE high-priority stream of its own
            sk = self._k1_stream
        self._set_slot(par)
        if self.net_streams >= 2:
            sa.wait_stream(torch.cuda.current_stream())            # whatever the caller queued before this tick (frame sources)
        with torch.cuda.stream(sk):
            if self.net_streams >= 2:
                if k >= 1:
                    sk.wait_event(self._k1_done[prev])             # the previous tick's K5 / K1 (gate state, source rings) first
                if k >= self.nslots:
                    sk.wait_event(self._net_done[par])             # the slot's previous tick's network has read its input tensors
                if self.k1_gate and k >= 1:
                    sk.wait_event(self._phase_ev[prev])            # ... and tick k-1's forward pass has reached its 20x20 phase
            packets = p._frames_for_detection(packets)             # roi / downsample (device work)
            plan = p.plan_tick(packets, process)
            motion = p._device_gates(packets, par) if p.has_gates else None       # K5 + gate parameters (no sync)
        sig = plan.signature
        replay = self.use_graph and self.two_streams and self._cap_sig == sig
        capture_after = self.use_graph and self.two_streams and not replay and sig in self._seen_sigs
        pres, raws = [], []
        for gi, g in enumerate(plan.groups):
            det = p.detectors[g.det]
            plan_det = hasattr(det, "plan_for") and getattr(det, "engine", None) == "fused" and det._infer_fn is None
            with torch.cuda.stream(sk), torch.inference_mode():
                if events and gi == 0: events[0].record()
                if before_k1 and gi == 0: before_k1()
                with ops.roctx(f"tick{k}:K1"):
                    pre = det.stage_pre([packets[i] for i in g.idx])    # K1 (temporal heads: into the frame ring + clip schedule)
                p._note_clips(g, getattr(pre, "infos", None))
                if events and gi == 0: events[1].record()
                if self.two_streams and gi == len(plan.groups) - 1:
                    self._k1_done[par].record(sk)
                    if sk is not sa:
                        sa.wait_event(self._k1_done[par])          # the network of this tick follows its K1 on another stream
            with torch.cuda.stream(sa):
                if self.two_streams and gi == 0 and k >= self.nslots:
                    sa.wait_event(self._done[par])                 # the slot's previous tick has finished reading its head tensors
                pres.append(pre)
                if replay and self.net_graph and self.replay_net:
                    self._net_graphs[gi][par].replay()
                else:
                    with torch.inference_mode():
                        if plan_det and self.two_streams:
                            fp = self._plan_of(det, pre[0])
                            fp.use_output(gi if self.net_streams >= 2 else 2 * gi + par)
                            fp.phase_event = self._phase_ev[par] if (self.k1_gate and gi == 0) else None
                        elif plan_det:
                            self._plan_of(det, pre[0])             # sets the plan's branch mode for this runner's layout
                        with ops.roctx(f"tick{k}:network"):
                            raws.append(det.stage_net(pre))
        with torch.cuda.stream(sa):
            if events: events[2].record()
            self._pending[par] = ("graph" if replay else "eager", plan, None if replay else raws, pres, motion,
                                  None if replay else events)
            clips = dict(getattr(p, "_host_dets", {}))             # clips the temporal heads fired in THIS tick
            if clips:
                p._host_dets.clear()
            self._meta[par] = (packets, t0, clips)
            if self.two_streams:
                if not plan.groups:
                    self._k1_done[par].record(sk)
                self._net_done[par].record(sa)
        self._set_slot(0)
        if self.two_streams:
            if self.depth == 1:
                self._issue_post(k, None)
            elif self.net_streams >= 2 and os.environ.get("RVA_TAIL_INLINE", "1") == "1":
                # two network streams: the tail goes right behind its own network on the same stream -- a tick is one chain
                # K1 -> network -> K2/K3 -> K4 -> ids -> snapshot, even and odd ticks on two streams (a third stream for the
                # tails measured 3 % slower: 19.06 k against 19.70 k frames/s, and made the result depend on how the runtime
                # spreads streams over its hardware queues)
                self._issue_post(k, None, stream=sa)
            elif k >= 1 and self._posted < k - 1:
                self._issue_post(k - 1, self._k1_done[par])        # K1 of this tick first, then the previous tail
        else:
            _, pl, rw, prs, mo, ev = self._pending[par]
            self._post_part(pl, rw, prs, mo, ev)
            self._ids_and_snapshot(k, ev)
        if capture_after:     # second tick of this shape: everything is sized and warm -> record it for the ticks to come
            if self._posted < k:
                self._issue_post(k, None)
            self._capture(plan, pres, motion)
        self._seen_sigs.add(sig)
        self._next += 1
        return k

    def _collect(self):
        if self._oldest >= self._next:
            raise RuntimeError("nothing in flight")
        k = self._oldest
        if self.two_streams:
            if self._posted < k:
                self._issue_post(k, None)                          # no younger tick was submitted: release the tail now
            if self.world_sharded:
                from . import dist as rdist
                # a tick behind an id exchange whose peer died never completes: bounded wait, then a non-zero exit (dist.fail)
                rdist.wait_event(self._done[k % self.nslots], what=f"tick {k} (id exchange of the sharded run)")
            else:
                self._done[k % self.nslots].synchronize()
            tables = self.dt.snapshot_fetch(k % self.nslots, wait=False)
        else:
            tables = self.dt.snapshot_fetch(k % self.nslots)       # tracks visible to the host
        status = self.dt.snapshot_status(k % self.nslots)
        self._oldest += 1
        return k, tables, status

    def collect(self):
        k, tables, status = self._collect()
        ops.DeviceTracker.raise_on_flags(status[2])
        return k, tables

    def collect_result(self) -> TickResult:
        k, tables, status = self._collect()
        packets, t0, clips = self._meta[k % self.nslots]
        if clips:
            self.pipe._host_dets = dict(clips)
        return self.pipe.finish(packets, tables, status, t0)

"""Tracker plugin API of the reference (tracker.py:18-95) backed by the K4 HIP kernels.

``IouTracker(config).update(stream_name, detections) -> List[Track]`` has the reference's
signature and semantics (greedy, sequential, non-exclusive matching; one GLOBAL id counter for all
streams, tracker.py:47; returns every surviving track of the stream in insertion order, and the
returned ``Track`` objects stay aliased to tracker state across updates as in the reference).
``update_batch`` updates many streams of one tick concurrently on the GPU (one wavefront per
stream) and assigns ids in the canonical order "streams in the order given" -- identical to
calling ``update`` for each stream in that order.

State lives in HBM (``ops.DeviceTracker``); Python keeps only the ``Track`` facade objects and the
string-valued temporal fields (tracker.py:58-67), which are carried host-side through the
``last_det`` column the kernel reports.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import ops
from .config import TrackerConfig
from .detector import Detection

_TEMPORAL_KEYS = ("action_label", "temporal_score", "sequence_start_frame", "sequence_end_frame")


@dataclass(slots=True)
class Track:
    """Track state that we propagate across frames (tracker.py:18-33)."""

    track_id: int
    class_id: int
    confidence: float
    bbox_xyxy: tuple[float, float, float, float]
    age: int = 0
    hits: int = 0
    action_label: Optional[str] = None
    temporal_score: Optional[float] = None
    sequence_start_frame: Optional[int] = None
    sequence_end_frame: Optional[int] = None


class IouTracker:
    """IoU tracker with the reference's behaviour, tables resident on the GPU."""

    def __init__(self, config: TrackerConfig, max_streams: int = 32, capacity: int = 1024, device: Optional[int] = None):
        self.config = config
        self._dev = ops.DeviceTracker(max_streams, config.max_age, config.max_iou_distance, config.min_hits,
                                      capacity=capacity, ctx=ops.context(device))
        self._index: Dict[str, int] = {}
        self._tracks: Dict[str, Dict[int, Track]] = {}
        self._max_streams = max_streams

    # -- helpers --------------------------------------------------------------------------------
    def _slot(self, stream_name: str) -> int:
        if stream_name not in self._index:
            if len(self._index) >= self._max_streams:
                raise RuntimeError(f"IouTracker was sized for {self._max_streams} streams")
            self._index[stream_name] = len(self._index)
            self._tracks[stream_name] = {}
        return self._index[stream_name]

    @staticmethod
    def _arrays(dets: Sequence[Detection]):
        n = len(dets)
        b = np.empty((n, 4), np.float64); c = np.empty(n, np.float64); k = np.empty(n, np.int64)
        for i, d in enumerate(dets):
            b[i] = d.bbox_xyxy; c[i] = d.confidence; k[i] = d.class_id
        return b, c, k

    def _materialise(self, stream_name: str, tab: dict, dets: Sequence[Detection]) -> List[Track]:
        live = self._tracks[stream_name]
        fresh: Dict[int, Track] = {}
        clip = dets if hasattr(dets, "start_frame") else None      # temporal.ClipInfo instead of a detection list
        for i in range(tab["n"]):
            tid = int(tab["id"][i])
            t = live.get(tid)
            box = tuple(float(v) for v in tab["boxes"][i])
            if t is None:
                t = Track(track_id=tid, class_id=int(tab["cls"][i]), confidence=float(tab["conf"][i]), bbox_xyxy=box,
                          age=int(tab["age"][i]), hits=int(tab["hits"][i]))
            else:
                t.confidence = float(tab["conf"][i]); t.bbox_xyxy = box
                t.age = int(tab["age"][i]); t.hits = int(tab["hits"][i])
            ld = int(tab["last_det"][i])
            if clip is not None:          # device path of a temporal head: the detection itself never reached the host
                if 0 <= ld < clip.n_dets:
                    t.action_label = clip.label(t.class_id)
                    t.temporal_score = t.confidence
                    t.sequence_start_frame, t.sequence_end_frame = clip.start_frame, clip.end_frame
            elif 0 <= ld < len(dets):     # tracker.py:58-67, 88-90: copy temporal fields when present
                d = dets[ld]
                for key in _TEMPORAL_KEYS:
                    if hasattr(d, key):
                        setattr(t, key, getattr(d, key, None))
            fresh[tid] = t
        self._tracks[stream_name] = fresh
        return list(fresh.values())

    # -- API ------------------------------------------------------------------------------------
    def update(self, stream_name: str, detections: Iterable[Detection]) -> List[Track]:
        return self.update_batch([stream_name], [list(detections)])[0]

    def update_batch(self, stream_names: Sequence[str], detections: Sequence[Sequence[Detection]]) -> List[List[Track]]:
        """One tick: ``detections[i]`` belongs to ``stream_names[i]``; ids are assigned as if
        ``update`` had been called for the streams in the order given.  For that equivalence the
        streams must be listed in registration order (the pipeline registers streams in config
        order, which is the canonical order of SURVEY.md hard part 2)."""
        slots = [self._slot(n) for n in stream_names]
        if sorted(slots) != slots or len(set(slots)) != len(slots):
            # canonical order differs from table order: fall back to one launch per stream (still on the GPU)
            return [self.update_batch([n], [d])[0] for n, d in zip(stream_names, detections)]
        dets = [list(d) for d in detections]
        self._dev.update_from_host({s: self._arrays(d) for s, d in zip(slots, dets)})
        self._dev.assign_ids()
        if len(slots) == 1:
            tabs = {slots[0]: self._dev.read(slots[0])}
        else:
            allt = self._dev.read_all()
            tabs = {s: allt[s] for s in slots}
        _, flags = self._dev.state()
        if flags & 1:
            raise RuntimeError("IouTracker table capacity exceeded; construct it with a larger `capacity`")
        return [self._materialise(n, tabs[s], d) for n, s, d in zip(stream_names, slots, dets)]

    # device-resident fast path used by the tick pipeline ------------------------------------------
    @property
    def device_tracker(self) -> ops.DeviceTracker:
        return self._dev

    def register_streams(self, names: Sequence[str]) -> List[int]:
        return [self._slot(n) for n in names]

    def tracks_from_tables(self, stream_names: Sequence[str], tables: Sequence[dict]) -> List[List[Track]]:
        return [self._materialise(n, t, ()) for n, t in zip(stream_names, tables)]

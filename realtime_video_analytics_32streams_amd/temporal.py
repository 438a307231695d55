"""Clip (temporal) detection path -- BASELINE config 5: CNN-LSTM over 16-frame clips.

Mirrors the reference's duck-typed temporal detector (temporal_detector.py:35-120, 150-426):
  * ``TemporalDetection`` = ``Detection`` + action_label / temporal_score / sequence_start_frame /
    sequence_end_frame (temporal_detector.py:35-47);
  * ``predict(packet)`` buffers per stream and returns ``[]`` until ``L*stride`` frames are held;
    the clip is ``buffer[i*stride]``; afterwards the last ``L*stride - step`` frames are kept with
    ``step = max(1, int(L*(1-overlap)))`` (temporal_detector.py:66-68, 88-118);
  * top-5 of the RAW model output (no softmax anywhere), emitted when ``>= confidence_threshold``,
    full-frame box, ``frame_id`` of the last packet (temporal_detector.py:392-424).

MI355X design: the reference keeps up to 32 full BGR frames per stream on the host (0.8 GB per 4K
stream) and resizes all 16 at clip time; here every frame is pre-processed ON ARRIVAL by the K1 clip
kernel into a per-stream ring of ``float[3,H,W]`` slots in HBM (301 kB each at 224x224), so clip
time is a gather of 16 slots + the network.
"""
from __future__ import annotations

import logging
from collections import deque
from dataclasses import dataclass
from typing import Deque, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import DetectorConfig
from .detector import Detection
from .video_stream import FramePacket

LOGGER = logging.getLogger(__name__)


@dataclass(slots=True)
class TemporalDetection(Detection):
    action_label: Optional[str] = None
    temporal_score: float = 0.0
    sequence_start_frame: int = 0
    sequence_end_frame: int = 0


class ClipSchedule:
    """The buffering rule alone (frame ids only): shared by the detector and by the tests that pin it
    against the reference's recorded schedule (tests/golden/temporal_buffer.json)."""

    def __init__(self, sequence_length: int, stride: int, overlap: float):
        self.L, self.stride = sequence_length, stride
        self.step = max(1, int(sequence_length * (1.0 - overlap)))
        self.need = sequence_length * stride
        self.keep = max(0, self.need - self.step)

    def push(self, buf: Deque, item) -> Tuple[Optional[list], list]:
        """Append ``item``; returns (clip items or None, items that left the buffer)."""
        dropped = []
        if len(buf) == self.need:          # deque(maxlen=need) semantics
            dropped.append(buf.popleft())
        buf.append(item)
        if len(buf) < self.need:
            return None, dropped
        clip = [buf[i * self.stride] for i in range(self.L)]
        n_drop = len(buf) - self.keep
        for _ in range(n_drop):
            dropped.append(buf.popleft())
        return clip, dropped


class CnnLstmNet(nn.Module):
    """Per-frame conv stem -> 2-layer LSTM -> linear on the last step (the architecture exported by
    scripts/convert_temporal_model_to_onnx.py:34-88, re-expressed with all T frames in one conv batch)."""

    def __init__(self, num_classes: int = 400, hidden: int = 512):
        super().__init__()
        self.stem = nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1),
            nn.Conv2d(64, 128, 3, 1, 1), nn.BatchNorm2d(128), nn.ReLU(inplace=True))
        self.rnn = nn.LSTM(128, hidden, num_layers=2, batch_first=True, dropout=0.5)
        self.head = nn.Linear(hidden, num_classes)

    def forward(self, clips: torch.Tensor) -> torch.Tensor:  # [B, T, 3, H, W]
        b, t = clips.shape[:2]
        f = self.stem(clips.flatten(0, 1)).mean((2, 3)).view(b, t, -1)
        out, _ = self.rnn(f)
        return self.head(out[:, -1])


class HipCNNLSTMDetector:
    """Duck-typed like the reference's temporal detectors: ``.config`` and ``.predict(packet)``."""

    def __init__(self, config: DetectorConfig, net: Optional[nn.Module] = None, seed: int = 1,
                 device: Optional[int] = None):
        self.config = config
        self.ctx = ops.context(device)
        self.device = torch.device("cuda", self.ctx.device)
        self.input_hw = (int(config.input_size[0]), int(config.input_size[1])) if config.input_size else (224, 224)
        self.sched = ClipSchedule(config.sequence_length, config.sequence_stride, config.temporal_overlap)
        self.sequence_step = self.sched.step
        self.half = bool(config.half)
        if net is None:
            st = torch.random.get_rng_state()
            torch.manual_seed(seed)
            net = CnnLstmNet(config.num_action_classes)
            torch.random.set_rng_state(st)
        self.net = net.eval().to(self.device)
        self.net = self.net.half() if self.half else self.net.float()
        self._buf: Dict[str, Deque] = {}
        self._ring: Dict[str, torch.Tensor] = {}
        self._free: Dict[str, List[int]] = {}

    def _state(self, name: str):
        if name not in self._buf:
            dt = torch.float16 if self.half else torch.float32
            self._buf[name] = deque()
            self._ring[name] = torch.empty((self.sched.need, 3, *self.input_hw), dtype=dt, device=self.device)
            self._free[name] = list(range(self.sched.need))
        return self._buf[name], self._ring[name], self._free[name]

    def predict(self, packet: FramePacket) -> List[Detection]:
        name = packet.stream.name
        buf, ring, free = self._state(name)
        if len(buf) == self.sched.need:       # the oldest entry is about to fall out: recycle its slot first
            free.append(buf[0][1])
        slot = free.pop()
        frame = packet.frame
        if isinstance(frame, ops.Nv12Surface):   # pre-process on arrival, straight into the ring slot
            ops.preprocess_nv12([frame], self.input_hw, self.half, out=ring[slot:slot + 1], clip=True, ctx=self.ctx)
            hw = (frame.height, frame.width)
        else:
            t = torch.from_numpy(np.ascontiguousarray(frame)) if isinstance(frame, np.ndarray) else frame
            ops.preprocess_bgr([t.to(self.device).contiguous()], self.input_hw, self.half, out=ring[slot:slot + 1],
                               clip=True, ctx=self.ctx)
            hw = (int(frame.shape[0]), int(frame.shape[1]))
        full = len(buf) == self.sched.need
        clip, dropped = self.sched.push(buf, (packet.frame_id, slot, hw))
        for k, item in enumerate(dropped):
            if not (full and k == 0):
                free.append(item[1])
        if clip is None:
            return []
        return self._predict_sequence(name, ring, clip)

    def _predict_sequence(self, name: str, ring: torch.Tensor, clip) -> List[Detection]:
        idx = torch.tensor([c[1] for c in clip], device=self.device)
        with torch.inference_mode():
            out = self.net(ring.index_select(0, idx).unsqueeze(0)).float().flatten().cpu().numpy()
        top_k = min(5, len(out))
        order = np.argsort(out, kind="stable")[-top_k:][::-1]     # temporal_detector.py:396-398
        h, w = clip[0][2]
        dets: List[Detection] = []
        for cid in order:
            conf = out[cid]
            if conf >= self.config.confidence_threshold:
                label = None
                if self.config.action_classes and cid < len(self.config.action_classes):
                    label = self.config.action_classes[cid]
                dets.append(TemporalDetection(stream_name=name, frame_id=clip[-1][0], class_id=int(cid),
                                              confidence=float(conf), bbox_xyxy=(0.0, 0.0, float(w), float(h)),
                                              action_label=label, temporal_score=float(conf),
                                              sequence_start_frame=clip[0][0], sequence_end_frame=clip[-1][0]))
        return dets

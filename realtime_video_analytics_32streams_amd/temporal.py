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
from typing import Deque, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _native as N
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


class Cnn3dNet(nn.Module):
    """The 3D-CNN the reference exports (scripts/convert_temporal_model_to_onnx.py:91-121): three Conv3d-BN-ReLU
    stages with (1,2,2) / (2,2,2) max-pools, global average pool, linear head.  Input ``[B,3,T,H,W]``."""

    def __init__(self, num_classes: int = 400):
        super().__init__()
        self.conv3d = nn.Sequential(
            nn.Conv3d(3, 64, 3, padding=1), nn.BatchNorm3d(64), nn.ReLU(inplace=True), nn.MaxPool3d((1, 2, 2), (1, 2, 2)),
            nn.Conv3d(64, 128, 3, padding=1), nn.BatchNorm3d(128), nn.ReLU(inplace=True), nn.MaxPool3d(2, 2),
            nn.Conv3d(128, 256, 3, padding=1), nn.BatchNorm3d(256), nn.ReLU(inplace=True), nn.AdaptiveAvgPool3d(1))
        self.fc = nn.Linear(256, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc(self.conv3d(x).flatten(1))


def load_reference_state_dict(net: nn.Module, state_dict, kind: str = "cnn_lstm") -> nn.Module:
    """Load a state dict in the REFERENCE's parameter naming (``DummyCNNLSTM``: cnn.* / lstm.* / fc.*, ``Dummy3DCNN``:
    conv3d.* / fc.*, scripts/convert_temporal_model_to_onnx.py:34-121) into :class:`CnnLstmNet` / :class:`Cnn3dNet`."""
    from .synth import map_temporal_key
    net.load_state_dict({map_temporal_key(k, kind): v for k, v in state_dict.items()})
    return net


class _HipTemporalDetector:
    """Shared body of the temporal heads, duck-typed like the reference's (``.config`` and ``.predict(packet)``):
    clip buffering (temporal_detector.py:58-120), per-frame pre-process on arrival into an HBM ring, top-5 of the
    raw model output (:392-424, :596-641, :760-799).  Subclasses fix the pre-process constants (``NORM``), the clip
    layout the network takes and the default input size."""

    NORM = N.NORM_IMAGENET_F32
    CLIP_LAYOUT = "TCHW"                  # "TCHW": [1,T,3,H,W] (CNN-LSTM, ConvGRU); "CTHW": [1,3,T,H,W] (3D-CNN)
    DEFAULT_HW = (224, 224)

    def __init__(self, config: DetectorConfig, net: Optional[nn.Module] = None, seed: int = 1,
                 device: Optional[int] = None, infer_fn=None):
        self.config = config
        self.ctx = ops.context(device)
        self.device = torch.device("cuda", self.ctx.device)
        self.input_hw = (int(config.input_size[0]), int(config.input_size[1])) if config.input_size else self.DEFAULT_HW
        self.sched = ClipSchedule(config.sequence_length, config.sequence_stride, config.temporal_overlap)
        self.sequence_step = self.sched.step
        self.half = bool(config.half)
        self._infer_fn = infer_fn
        self.net = None
        if infer_fn is None:
            if net is None:
                st = torch.random.get_rng_state()
                torch.manual_seed(seed)
                net = self._default_net()
                torch.random.set_rng_state(st)
            self.net = net.eval().to(self.device)
            self.net = self.net.to(self._frame_dtype())
        self._buf: Dict[str, Deque] = {}
        self._ring: Dict[str, torch.Tensor] = {}
        self._free: Dict[str, List[int]] = {}

    # -- per-head hooks ---------------------------------------------------------------------------
    def _default_net(self) -> nn.Module:
        raise NotImplementedError

    def _frame_dtype(self) -> torch.dtype:
        return torch.float16 if self.half else torch.float32

    # -- reference-shaped pre-process of a whole clip (parity surface for _preprocess_sequence) -----
    def preprocess_sequence(self, frames: Sequence) -> torch.Tensor:
        """``_preprocess_sequence`` of the reference for device frames: ``[1,T,3,H,W]`` or ``[1,3,T,H,W]``."""
        layout = N.LAYOUT_CNHW if self.CLIP_LAYOUT == "CTHW" else N.LAYOUT_NCHW
        dev = [self._to_device(f) for f in frames]
        return ops.preprocess_frames(dev, self.input_hw, self.NORM, layout, self._frame_dtype(), ctx=self.ctx).unsqueeze(0)

    def _to_device(self, frame):
        if isinstance(frame, ops.Nv12Surface):
            return frame
        t = torch.from_numpy(np.ascontiguousarray(frame)) if isinstance(frame, np.ndarray) else frame
        return t.to(self.device).contiguous()

    def _state(self, name: str):
        if name not in self._buf:
            self._buf[name] = deque()
            self._ring[name] = torch.empty((self.sched.need, 3, *self.input_hw), dtype=self._frame_dtype(), device=self.device)
            self._free[name] = list(range(self.sched.need))
        return self._buf[name], self._ring[name], self._free[name]

    def predict(self, packet: FramePacket) -> List[Detection]:
        name = packet.stream.name
        buf, ring, free = self._state(name)
        if len(buf) == self.sched.need:       # the oldest entry is about to fall out: recycle its slot first
            free.append(buf[0][1])
        slot = free.pop()
        frame = packet.frame
        hw = (frame.height, frame.width) if isinstance(frame, ops.Nv12Surface) else (int(frame.shape[0]), int(frame.shape[1]))
        # pre-process on arrival, straight into the ring slot
        ops.preprocess_frames([self._to_device(frame)], self.input_hw, self.NORM, N.LAYOUT_NCHW, self._frame_dtype(),
                              out=ring[slot:slot + 1], ctx=self.ctx)
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
        x = ring.index_select(0, idx)                                   # [T,3,H,W]
        x = (x.permute(1, 0, 2, 3).contiguous() if self.CLIP_LAYOUT == "CTHW" else x).unsqueeze(0)
        with torch.inference_mode():
            raw = self._infer_fn(x) if self._infer_fn is not None else self.net(x)
        out = raw.float().flatten().cpu().numpy()
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


class HipCNNLSTMDetector(_HipTemporalDetector):
    """CNN-LSTM head (temporal_detector.py:150-426): float32 ImageNet normalisation, clips ``[1,T,3,H,W]``."""

    def _default_net(self) -> nn.Module:
        return CnnLstmNet(self.config.num_action_classes)


class HipCNN3DDetector(_HipTemporalDetector):
    """3D-CNN head (temporal_detector.py:429-641; also what the reference instantiates for ``slow_fast``,
    detector.py:70-74): mean 0.45 / std 0.225, clips ``[1,3,T,H,W]``, default input 112x112 (:544)."""

    NORM = N.NORM_VIDEO_F32
    CLIP_LAYOUT = "CTHW"
    DEFAULT_HW = (112, 112)

    def _default_net(self) -> nn.Module:
        return Cnn3dNet(self.config.num_action_classes)


class HipConvGRUDetector(_HipTemporalDetector):
    """ConvGRU head (temporal_detector.py:644-799).  Its pre-process holds the ImageNet constants in float64 arrays, so
    the normalisation runs in float64 and the clip is float64 unless ``half`` (:741-752).  The reference has no
    architecture for this head (it only loads an OpenVINO/ONNX file), so a network -- ``net`` taking ``[1,T,3,H,W]`` --
    or an ``infer_fn`` must be supplied; constructing the detector without one fails like a missing model file."""

    NORM = N.NORM_IMAGENET_F64

    def _frame_dtype(self) -> torch.dtype:
        return torch.float16 if self.half else torch.float64

    def _default_net(self) -> nn.Module:
        raise RuntimeError("ConvGRU: the reference defines no architecture for this head; pass net=... or infer_fn=... "
                           f"(model_path '{self.config.model_path}' cannot be loaded offline)")

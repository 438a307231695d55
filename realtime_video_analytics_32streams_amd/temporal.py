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


@dataclass
class ClipInfo:
    """Host-side facts of the clip a stream fired in a tick (device path): what ``Track`` copies from a
    ``TemporalDetection`` (tracker.py:58-67) that is not in the track table itself."""
    start_frame: int
    end_frame: int
    labels: Optional[Sequence[str]]
    n_dets: int

    def label(self, class_id: int) -> Optional[str]:
        return self.labels[class_id] if self.labels and class_id < len(self.labels) else None


@dataclass
class _ClipTick:
    """What ``stage_pre`` hands to ``stage_net`` / ``stage_post`` for one tick of one frame group."""
    rows: int                                   # streams of the group (batch rows of the PostBuffers)
    fired: List[Tuple[int, List[int], Tuple[int, int]]]   # (row, ring slots of the clip, (h, w) of its first frame)
    cols: List[int]                             # ring column of every row
    infos: Dict[str, ClipInfo]
    slot: int                                   # pipeline slot (tick parity) whose buffers this tick uses


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
        # batched device path (stage_pre / stage_net / stage_post): one ring for all streams, [slots, columns, 3, H, W]
        self._slot = 0                         # set by PipelinedTicks: tick parity -> which result buffers a tick uses
        self._col: Dict[str, int] = {}         # stream name -> ring column
        self._arrivals: Dict[str, int] = {}    # frames seen per stream (ring slot = arrivals % ring slots)
        self._bbuf: Dict[str, Deque] = {}      # per stream: (frame_id, ring slot, (h, w)) of the buffered frames
        self._bring: Optional[torch.Tensor] = None
        self._post: Dict[tuple, ops.PostBuffers] = {}
        self.two_chain_ok = True               # PipelinedTicks may run consecutive ticks as two chains on two streams

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

    # -- batched device path: a whole tick of this head without a host round trip ----------------------------------------
    RING_EXTRA = 8        # ring slots beyond the clip buffer: up to eight ticks may be in flight (K1 of tick k+7 beside the network of k)

    @staticmethod
    def geometry_key(frame) -> tuple:
        if isinstance(frame, ops.Nv12Surface):
            return (int(frame.width), int(frame.height), "nv12")
        return (int(frame.shape[1]), int(frame.shape[0]), "bgr")

    def reserve_streams(self, names: Sequence[str]) -> None:
        """Give every stream of the pipeline its ring column now (TickPipeline calls this once): the ring then never grows
        while ticks are in flight."""
        self._columns(list(names))

    def _columns(self, names: Sequence[str]) -> List[int]:
        new = [n for n in names if n not in self._col]
        if new:
            for n in new:
                self._col[n] = len(self._col)
                self._arrivals[n] = 0
                self._bbuf[n] = deque()
            slots = self.sched.need + self.RING_EXTRA
            if self._bring is not None:
                # a stream nobody announced shows up mid-run: ticks in flight on other chain streams may still read the old
                # ring, and the caching allocator could hand its block out again -- drain the device before the swap (rare,
                # and never on the reserved path)
                torch.cuda.synchronize(self.device)
            ring = torch.empty((slots, len(self._col), 3, *self.input_hw), dtype=self._frame_dtype(), device=self.device)
            if self._bring is not None:
                ring[:, :self._bring.shape[1]] = self._bring
                torch.cuda.synchronize(self.device)
            self._bring = ring
        return [self._col[n] for n in names]

    def stage_pre(self, packets: Sequence[FramePacket]) -> _ClipTick:
        """K1 of the tick: every frame is pre-processed on arrival straight into its ring slot -- ONE launch when the
        streams of the group sit in adjacent columns and have seen the same number of frames (the steady state), one
        launch per stream otherwise -- and the clip schedule (host logic, temporal_detector.py:88-118) advances."""
        names = [p.stream.name for p in packets]
        cols = self._columns(names)
        R = self._bring.shape[0]
        slots = [self._arrivals[n] % R for n in names]
        frames = [self._to_device(p.frame) for p in packets]
        same_geo = len({self.geometry_key(f) for f in frames}) == 1
        if same_geo and len(set(slots)) == 1 and cols == list(range(cols[0], cols[0] + len(cols))):
            ops.preprocess_frames(frames, self.input_hw, self.NORM, N.LAYOUT_NCHW, self._frame_dtype(),
                                  out=self._bring[slots[0], cols[0]:cols[0] + len(cols)], ctx=self.ctx)
        else:
            for f, sl, c in zip(frames, slots, cols):
                ops.preprocess_frames([f], self.input_hw, self.NORM, N.LAYOUT_NCHW, self._frame_dtype(),
                                      out=self._bring[sl, c:c + 1], ctx=self.ctx)
        fired, infos = [], {}
        for row, (p, n, sl) in enumerate(zip(packets, names, slots)):
            f = p.frame
            hw = (f.height, f.width) if isinstance(f, ops.Nv12Surface) else (int(f.shape[0]), int(f.shape[1]))
            self._arrivals[n] += 1
            clip, _ = self.sched.push(self._bbuf[n], (p.frame_id, sl, hw))
            if clip is not None:
                fired.append((row, [c[1] for c in clip], clip[0][2]))
                infos[n] = ClipInfo(clip[0][0], clip[-1][0], self.config.action_classes, min(5, self.config.num_action_classes))
        return _ClipTick(len(packets), fired, cols, infos, self._slot)

    def stage_net(self, pre: _ClipTick) -> Optional[torch.Tensor]:
        """The clips that fired this tick as ONE network batch: ``[n_fired, classes]`` raw outputs (no softmax)."""
        if not pre.fired:
            return None
        C_ = self._bring.shape[1]
        flat = self._bring.view(-1, 3, *self.input_hw)
        idx = torch.tensor([[sl * C_ + pre.cols[row] for sl in slots] for row, slots, _ in pre.fired], device=self.device)
        x = flat.index_select(0, idx.flatten()).view(len(pre.fired), idx.shape[1], 3, *self.input_hw)      # [B,T,3,H,W]
        if self.CLIP_LAYOUT == "CTHW":
            x = x.permute(0, 2, 1, 3, 4).contiguous()
        with torch.inference_mode():
            return (self._infer_fn(x) if self._infer_fn is not None else self.net(x)).float()

    def stage_post(self, raw: Optional[torch.Tensor], pre: _ClipTick) -> ops.PostBuffers:
        """Top-5 of the raw output per clip in the reference's order (ascending stable sort, last five reversed:
        temporal_detector.py:396-398), full-frame boxes (:418); the ``>= confidence_threshold`` test of :402 is the
        tracker kernel's filter (same float64 comparison).  Streams without a clip this tick get an empty row: their
        tracks age like ``tracker.update(name, [])``."""
        key = (pre.rows, pre.slot)
        post = self._post.get(key)
        if post is None:
            post = self._post[key] = ops.PostBuffers.allocate(pre.rows, 8, self.device)
        post.counts.zero_()
        if raw is not None:
            k = min(5, raw.shape[1])
            order = torch.sort(raw, dim=1, stable=True).indices[:, -k:].flip(1)            # [n_fired, k]
            rows = torch.tensor([row for row, _, _ in pre.fired], device=self.device)
            post.scores[rows, :k] = torch.gather(raw, 1, order)
            post.cls[rows, :k] = order.to(torch.int32)
            wh = torch.tensor([[0.0, 0.0, float(hw[1]), float(hw[0])] for _, _, hw in pre.fired], device=self.device)
            post.boxes[rows, :k] = wh[:, None, :]
            post.counts[rows] = k
        return post

    def predict_batch_device(self, packets: Sequence[FramePacket]) -> ops.PostBuffers:
        pre = self.stage_pre(packets)
        self.last_clip_infos = pre.infos
        return self.stage_post(self.stage_net(pre), pre)

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

"""YAML configuration for the hot path: the same keys the reference reads (config.py:54-351),
restricted to the sections the detect/track path consumes.

Compatibility rules kept from the reference:
  * unknown keys are dropped silently (config.py:304-307);
  * ``streams`` must be a list, ``detectors`` a mapping id -> detector section (config.py:321-336);
  * the validation messages/limits of StreamConfig / DetectorConfig / TrackerConfig / PipelineConfig;
  * fields that the reference declares but never reads (``batch_size``, ``tracker.type``, ...) are
    accepted and carried, so a reference YAML loads unchanged.
Additions: backend ``"hip"`` (this library) is a valid detector backend, and the out-of-scope
sections (kafka / prometheus / ffmpeg_simulator) are kept as opaque dicts.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

import yaml

HIP_BACKENDS = ("hip", "rocm", "mi355x")
REFERENCE_BACKENDS = ("ultralytics", "tensorrt", "onnx", "onnxruntime", "openvino", "rknn", "rk3588")
TEMPORAL_MODELS = ("cnn_lstm", "3d_cnn", "conv_gru", "slow_fast")
MODEL_TYPES = ("yolov5", "yolov8", "resnet") + TEMPORAL_MODELS


class ConfigError(RuntimeError):
    """Raised when the supplied configuration is invalid."""


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise ConfigError(msg)


@dataclass(slots=True)
class StreamConfig:
    name: str
    url: str
    enabled: bool = True
    target_fps: Optional[float] = None
    batch_size: int = 1
    warmup_seconds: float = 2.0
    reconnect_backoff: float = 5.0
    max_retries: Optional[int] = None
    detector_id: Optional[str] = None
    roi_polygons: Optional[list] = None
    motion_filter: bool = False
    motion_threshold: float = 0.02
    downsample_ratio: float = 1.0
    adaptive_fps: bool = False
    min_target_fps: float = 5.0
    idle_frame_tolerance: int = 60
    ffmpeg_simulator: Optional[dict] = None   # out of scope: carried, never interpreted

    def validate(self) -> None:
        n = self.name
        _need(bool(n), "Stream name must not be empty")
        _need(bool(self.url), f"Stream '{n}' must define a non-empty url")
        _need(self.batch_size >= 1, f"Stream '{n}' batch_size must be >= 1")
        _need(self.target_fps is None or self.target_fps > 0, f"Stream '{n}' target_fps must be > 0 if provided")
        _need(self.warmup_seconds >= 0, f"Stream '{n}' warmup_seconds must be >= 0")
        _need(self.reconnect_backoff >= 0, f"Stream '{n}' reconnect_backoff must be >= 0")
        _need(self.max_retries is None or self.max_retries >= 0, f"Stream '{n}' max_retries must be >= 0")
        _need(self.motion_threshold >= 0, f"Stream '{n}' motion_threshold must be >= 0")
        _need(0.1 <= self.downsample_ratio <= 1.0, f"Stream '{n}' downsample_ratio must be between 0.1 and 1.0")
        if self.adaptive_fps:
            _need(0 < self.min_target_fps <= (self.target_fps or 30),
                  f"Stream '{n}' min_target_fps must be > 0 and <= target_fps when adaptive_fps is enabled")


@dataclass(slots=True)
class DetectorConfig:
    model_path: str = "yolov8n.pt"
    device: str = "auto"
    backend: str = "ultralytics"
    model_type: str = "yolov8"
    confidence_threshold: float = 0.5
    iou_threshold: float = 0.45
    classes: Optional[List[int]] = None
    half: bool = False
    warmup: bool = True
    input_size: Optional[List[int]] = None  # H, W
    tensorrt_max_workspace_size: int = 1 << 30
    tensorrt_use_fp16: bool = False
    resnet_num_classes: int = 1000
    resnet_top_k: int = 5
    sequence_length: int = 16
    sequence_stride: int = 1
    temporal_overlap: float = 0.5
    temporal_pooling: str = "avg"
    action_classes: Optional[List[str]] = None
    num_action_classes: int = 400

    def validate(self) -> None:
        _need(bool(self.model_path), "Detector model_path must not be empty")
        _need(self.backend in REFERENCE_BACKENDS + HIP_BACKENDS,
              f"Detector backend must be one of {set(REFERENCE_BACKENDS + HIP_BACKENDS)}")
        _need(self.model_type in MODEL_TYPES, f"Model type must be one of {set(MODEL_TYPES)}")
        _need(0.0 < self.confidence_threshold <= 1.0, "confidence_threshold must be in (0, 1]")
        _need(0.0 < self.iou_threshold <= 1.0, "iou_threshold must be in (0, 1]")
        _need(not self.input_size or len(self.input_size) == 2, "input_size must be [height, width]")
        if self.model_type in TEMPORAL_MODELS:
            _need(self.backend in ("onnx", "onnxruntime", "openvino") + HIP_BACKENDS,
                  "Temporal models currently only supported with ONNX Runtime, OpenVINO or HIP backends")
            _need(self.sequence_length > 0, "sequence_length must be > 0 for temporal models")
            _need(self.sequence_stride > 0, "sequence_stride must be > 0 for temporal models")
            _need(0.0 <= self.temporal_overlap < 1.0, "temporal_overlap must be in [0, 1) for temporal models")
            _need(self.temporal_pooling in ("avg", "max", "last"), "temporal_pooling must be one of: avg, max, last")
            _need(self.num_action_classes > 0, "num_action_classes must be > 0 for temporal models")


@dataclass(slots=True)
class TrackerConfig:
    type: str = "byte_track"      # a label nothing reads: the tracker is always the IoU tracker
    max_age: int = 30
    max_iou_distance: float = 0.7  # despite the name: the MINIMUM IoU for a match (tracker.py:106)
    min_hits: int = 3

    def validate(self) -> None:
        _need(self.max_age >= 1, "Tracker max_age must be >= 1")
        _need(self.max_iou_distance > 0, "Tracker max_iou_distance must be > 0")
        _need(self.min_hits >= 0, "Tracker min_hits must be >= 0")


@dataclass(slots=True)
class PipelineConfig:
    streams: List[StreamConfig] = field(default_factory=list)
    detector: DetectorConfig = field(default_factory=DetectorConfig)
    detectors: Dict[str, DetectorConfig] = field(default_factory=dict)
    tracker: TrackerConfig = field(default_factory=TrackerConfig)
    kafka: Dict[str, Any] = field(default_factory=dict)        # out of scope: opaque
    prometheus: Dict[str, Any] = field(default_factory=dict)   # out of scope: opaque
    max_concurrent_streams: int = 32
    stats_interval_seconds: float = 15.0

    def validate(self) -> None:
        _need(bool(self.streams), "At least one stream must be configured")
        _need(self.max_concurrent_streams >= 1, "max_concurrent_streams must be >= 1")
        _need(len(self.streams) <= self.max_concurrent_streams,
              f"Configured {len(self.streams)} streams but max_concurrent_streams={self.max_concurrent_streams}")
        _need(self.stats_interval_seconds > 0, "stats_interval_seconds must be > 0")
        for s in self.streams:
            _need(not s.detector_id or s.detector_id in self.detectors,
                  f"Stream '{s.name}' references unknown detector_id='{s.detector_id}'")
            s.validate()
        self.detector.validate()
        for d in self.detectors.values():
            d.validate()
        self.tracker.validate()

    def detector_for(self, stream: StreamConfig) -> DetectorConfig:
        return self.detectors[stream.detector_id] if stream.detector_id else self.detector


def _build(cls, data: Optional[dict]):
    names = {f.name for f in dataclasses.fields(cls)}
    return cls(**{k: v for k, v in (data or {}).items() if k in names})


def load_config(path) -> PipelineConfig:
    p = Path(path)
    if not p.exists():
        raise ConfigError(f"Configuration file not found: {p}")
    raw = yaml.safe_load(p.read_text(encoding="utf-8"))
    return config_from_dict(raw)


def config_from_dict(raw) -> PipelineConfig:
    _need(isinstance(raw, dict), "Top level configuration must be a mapping/dictionary")
    _need(isinstance(raw.get("streams"), list), "'streams' must be a list in the configuration")
    dets = raw.get("detectors") or {}
    _need(isinstance(dets, dict), "'detectors' section must be a mapping of id -> config")
    cfg = PipelineConfig(
        streams=[_build(StreamConfig, s) for s in raw["streams"]],
        detector=_build(DetectorConfig, raw.get("detector")),
        detectors={k: _build(DetectorConfig, v) for k, v in dets.items()},
        tracker=_build(TrackerConfig, raw.get("tracker")),
        kafka=dict(raw.get("kafka") or {}),
        prometheus=dict(raw.get("prometheus") or {}),
        max_concurrent_streams=raw.get("max_concurrent_streams", 32),
        stats_interval_seconds=raw.get("stats_interval_seconds", 15.0),
    )
    cfg.validate()
    return cfg

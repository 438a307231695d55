"""ResNet classification head behind the detector API (reference detector.py:870-1001).

The reference loads a ResNet from an OpenVINO / ONNX file and returns the top-K classes of the RAW output vector as
full-frame ``Detection`` objects (no softmax; ``np.argsort(output)[-k:][::-1]``, kept when ``>= confidence_threshold``).
Pre-process = the float32 ImageNet normalisation of the clip kernel on one frame (``rva_preprocess_frames_*``).  The
network here is a from-scratch torch ResNet-18 with seeded weights (no model files exist offline); any module or
``infer_fn`` mapping ``[B,3,H,W] -> [B,num_classes]`` can replace it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .config import DetectorConfig
from .detector import BaseDetector, Detection
from .video_stream import FramePacket


class _BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.c1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False); self.b1 = nn.BatchNorm2d(cout)
        self.c2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False); self.b2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = torch.relu(self.b1(self.c1(x)))
        y = self.b2(self.c2(y))
        return torch.relu(y + (x if self.down is None else self.down(x)))


class ResNet18(nn.Module):
    def __init__(self, num_classes: int = 1000):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, 2, 1))
        cfg, blocks, cin = [(64, 1), (128, 2), (256, 2), (512, 2)], [], 64
        for cout, stride in cfg:
            blocks += [_BasicBlock(cin, cout, stride), _BasicBlock(cout, cout, 1)]
            cin = cout
        self.layers = nn.Sequential(*blocks)
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        return self.fc(self.layers(self.stem(x)).mean((2, 3)))


class HipResNetDetector(BaseDetector):
    """``predict(packet)`` / ``predict_batch(packets)`` with the reference's top-K rule (detector.py:945-977)."""

    def __init__(self, config: DetectorConfig, net: Optional[nn.Module] = None, infer_fn=None, seed: int = 2,
                 device: Optional[int] = None):
        super().__init__(config)
        self.ctx = ops.context(device)
        self.device = torch.device("cuda", self.ctx.device)
        self.input_hw = (int(config.input_size[0]), int(config.input_size[1])) if config.input_size else (224, 224)
        self._infer_fn = infer_fn
        self.net = None
        if infer_fn is None:
            if net is None:
                st = torch.random.get_rng_state()
                torch.manual_seed(seed)
                net = ResNet18(config.resnet_num_classes)
                torch.random.set_rng_state(st)
            self.net = net.eval().float().to(self.device).to(memory_format=torch.channels_last)

    def _preprocess(self, frames: Sequence) -> torch.Tensor:
        dev = []
        for f in frames:
            if isinstance(f, ops.Nv12Surface):
                dev.append(f)
            else:
                t = torch.from_numpy(np.ascontiguousarray(f)) if isinstance(f, np.ndarray) else f
                dev.append(t.to(self.device).contiguous())
        return ops.preprocess_frames(dev, self.input_hw, N.NORM_IMAGENET_F32, N.LAYOUT_NCHW, torch.float32, ctx=self.ctx)

    def predict_batch(self, packets: Sequence[FramePacket]) -> List[List[Detection]]:
        groups = {}
        for i, p in enumerate(packets):
            f = p.frame
            key = (f.width, f.height, "nv12") if isinstance(f, ops.Nv12Surface) else (int(f.shape[1]), int(f.shape[0]), "bgr")
            groups.setdefault(key, []).append(i)
        out: List[Optional[List[Detection]]] = [None] * len(packets)
        for (w, h, _), idxs in groups.items():
            with torch.inference_mode():
                x = self._preprocess([packets[i].frame for i in idxs])
                raw = self._infer_fn(x) if self._infer_fn is not None else self.net(x)
            scores = raw.float().cpu().numpy()
            for row, i in enumerate(idxs):
                p, o = packets[i], scores[row].flatten()
                order = np.argsort(o, kind="stable")[-self.config.resnet_top_k:][::-1]      # detector.py:956
                out[i] = [Detection(stream_name=p.stream.name, frame_id=p.frame_id, class_id=int(c), confidence=float(o[c]),
                                    bbox_xyxy=(0.0, 0.0, float(w), float(h)))
                          for c in order if o[c] >= self.config.confidence_threshold]
        return out  # type: ignore[return-value]

    def predict(self, packet: FramePacket) -> List[Detection]:
        return self.predict_batch([packet])[0]

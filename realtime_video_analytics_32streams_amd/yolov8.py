"""YOLOv8 n/s/m detector network, written from the published architecture.

The reference never holds the network itself: it runs an exported ONNX graph through ONNX Runtime
(detector.py:564-609) whose single output is ``[B, 84, 8400]`` = (cx, cy, w, h in input pixels,
80 post-sigmoid class scores).  This module produces a tensor of exactly that contract on
PyTorch-ROCm (fp16, channels-last => MIOpen/hipBLASLt MFMA kernels); everything after it is the
hand-written HIP post-process.  There are no weights offline (models/ is stripped), so
weights are seeded random unless a local state-dict is given; "parity" for this stage is
self-parity (GPU fp16 vs CPU fp32 of the same module), said so wherever it is reported.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# depth multiple, width multiple, max channels  (Ultralytics yolov8.yaml scales)
SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}


def _divisible(x: float, d: int = 8) -> int:
    return int(math.ceil(x / d) * d)


class ConvBnAct(nn.Module):
    """Conv2d(bias=False) + BatchNorm + SiLU; ``fuse()`` folds the norm into the conv for inference."""

    def __init__(self, c1, c2, k=1, s=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = act
        self.fused = False

    def forward(self, x):
        x = self.conv(x)
        if not self.fused:
            x = self.bn(x)
        return F.silu(x, inplace=True) if self.act else x

    @torch.no_grad()
    def fuse(self):
        if self.fused:
            return
        w = self.conv.weight
        scale = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
        conv = nn.Conv2d(self.conv.in_channels, self.conv.out_channels, self.conv.kernel_size, self.conv.stride,
                         self.conv.padding, bias=True).to(w.device, w.dtype)
        conv.weight.copy_(w * scale.view(-1, 1, 1, 1))
        conv.bias.copy_(self.bn.bias - self.bn.running_mean * scale)
        self.conv = conv
        self.bn = nn.Identity()
        self.fused = True


class Bottleneck(nn.Module):
    def __init__(self, c, shortcut=True):
        super().__init__()
        self.cv1 = ConvBnAct(c, c, 3)
        self.cv2 = ConvBnAct(c, c, 3)
        self.add = shortcut

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = ConvBnAct(c1, 2 * self.c, 1)
        self.cv2 = ConvBnAct((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = ConvBnAct(c1, c_, 1)
        self.cv2 = ConvBnAct(c_ * 4, c2, 1)
        self.k = k

    def forward(self, x):
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], self.k, 1, self.k // 2))
        return self.cv2(torch.cat(y, 1))


class DetectHead(nn.Module):
    """Anchor-free head with DFL; inference output ``[B, 4 + nc, A]`` (xywh * stride, sigmoid scores)."""

    def __init__(self, nc: int, ch: Sequence[int], strides=(8, 16, 32), reg_max: int = 16):
        super().__init__()
        self.nc, self.reg_max, self.strides = nc, reg_max, strides
        c2 = max(16, ch[0] // 4, reg_max * 4)
        c3 = max(ch[0], min(nc, 100))
        self.box = nn.ModuleList(nn.Sequential(ConvBnAct(c, c2, 3), ConvBnAct(c2, c2, 3), nn.Conv2d(c2, 4 * reg_max, 1)) for c in ch)
        self.cls = nn.ModuleList(nn.Sequential(ConvBnAct(c, c3, 3), ConvBnAct(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for c in ch)
        self.register_buffer("proj", torch.arange(reg_max, dtype=torch.float32), persistent=False)
        self._anchor_cache = {}
        for b, c, s in zip(self.box, self.cls, strides):   # standard prior initialisation of the architecture
            b[-1].bias.data[:] = 1.0
            c[-1].bias.data[:nc] = math.log(5 / nc / (640 / s) ** 2)

    def _anchors(self, feats: List[torch.Tensor]):
        key = tuple((f.shape[2], f.shape[3]) for f in feats) + (feats[0].device, feats[0].dtype)
        if key not in self._anchor_cache:
            pts, strs = [], []
            for f, s in zip(feats, self.strides):
                h, w = f.shape[2], f.shape[3]
                sy, sx = torch.meshgrid(torch.arange(h, device=f.device, dtype=torch.float32) + 0.5,
                                        torch.arange(w, device=f.device, dtype=torch.float32) + 0.5, indexing="ij")
                pts.append(torch.stack((sx, sy), -1).view(-1, 2))
                strs.append(torch.full((h * w, 1), float(s), device=f.device))
            self._anchor_cache[key] = (torch.cat(pts).t().contiguous().to(feats[0].dtype),   # [2, A]
                                       torch.cat(strs).t().contiguous().to(feats[0].dtype))  # [1, A]
        return self._anchor_cache[key]

    def forward(self, feats: List[torch.Tensor], raw: bool = False):
        b = feats[0].shape[0]
        box = torch.cat([m(f).reshape(b, 4 * self.reg_max, -1) for m, f in zip(self.box, feats)], 2)
        cls = torch.cat([m(f).reshape(b, self.nc, -1) for m, f in zip(self.cls, feats)], 2)
        if raw:
            return box, cls
        anchors, strides = self._anchors(feats)
        a = box.shape[2]
        dist = box.view(b, 4, self.reg_max, a).float().softmax(2)
        dist = (dist * self.proj.view(1, 1, -1, 1)).sum(2).to(box.dtype)      # DFL expectation, [b, 4, A]
        lt, rb = dist[:, :2], dist[:, 2:]
        x1y1, x2y2 = anchors.unsqueeze(0) - lt, anchors.unsqueeze(0) + rb
        xywh = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides.unsqueeze(0)
        return torch.cat((xywh, cls.sigmoid()), 1)                          # [b, 4 + nc, A]


class YoloV8(nn.Module):
    def __init__(self, scale: str = "s", nc: int = 80):
        super().__init__()
        d, w, mc = SCALES[scale]
        ch = lambda c: _divisible(min(c, mc) * w)      # noqa: E731
        rep = lambda n: max(round(n * d), 1)           # noqa: E731
        c1, c2, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
        self.scale, self.nc = scale, nc
        self.b0 = ConvBnAct(3, c1, 3, 2)
        self.b1 = ConvBnAct(c1, c2, 3, 2)
        self.b2 = C2f(c2, c2, rep(3), True)
        self.b3 = ConvBnAct(c2, c3, 3, 2)
        self.b4 = C2f(c3, c3, rep(6), True)
        self.b5 = ConvBnAct(c3, c4, 3, 2)
        self.b6 = C2f(c4, c4, rep(6), True)
        self.b7 = ConvBnAct(c4, c5, 3, 2)
        self.b8 = C2f(c5, c5, rep(3), True)
        self.b9 = SPPF(c5, c5, 5)
        self.h12 = C2f(c5 + c4, c4, rep(3))
        self.h15 = C2f(c4 + c3, c3, rep(3))
        self.h16 = ConvBnAct(c3, c3, 3, 2)
        self.h18 = C2f(c3 + c4, c4, rep(3))
        self.h19 = ConvBnAct(c4, c4, 3, 2)
        self.h21 = C2f(c4 + c5, c5, rep(3))
        self.detect = DetectHead(nc, (c3, c4, c5))

    def features(self, x):
        x = self.b2(self.b1(self.b0(x)))
        p3 = self.b4(self.b3(x))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        n4 = self.h12(torch.cat((F.interpolate(p5, scale_factor=2.0, mode="nearest"), p4), 1))
        n3 = self.h15(torch.cat((F.interpolate(n4, scale_factor=2.0, mode="nearest"), p3), 1))
        m4 = self.h18(torch.cat((self.h16(n3), n4), 1))
        m5 = self.h21(torch.cat((self.h19(m4), p5), 1))
        return [n3, m4, m5]

    def forward(self, x, raw: bool = False):
        return self.detect(self.features(x), raw=raw)

    @torch.no_grad()
    def fuse(self):
        for m in self.modules():
            if isinstance(m, ConvBnAct):
                m.fuse()
        return self


def count_macs(model: nn.Module, hw=(640, 640)) -> int:
    """Multiply-accumulates of every Conv2d for one image (the FLOP figure quoted in DESIGN.md = 2x)."""
    macs = 0
    hooks = []

    def hook(m, inp, out):
        nonlocal macs
        macs += out.numel() // out.shape[0] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]

    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    was = model.training
    model.eval()
    with torch.no_grad():
        p = next(model.parameters())
        model(torch.zeros(1, 3, *hw, device=p.device, dtype=p.dtype))
    model.train(was)
    for h in hooks:
        h.remove()
    return macs


def variant_from_path(model_path: str) -> str:
    name = Path(model_path).stem.lower()
    for s in "nsmlx":
        if name.startswith(f"yolov8{s}"):
            return s
    return "n"


# Ultralytics layer index (yolov8.yaml: backbone 0-9, head 10-22; 10/11/13/14/17/20 are Upsample / Concat without
# parameters) -> attribute of YoloV8.  Inside a layer the two code bases use the same names (conv / bn, cv1 / cv2 / m.N).
_ULTRALYTICS_LAYERS = {0: "b0", 1: "b1", 2: "b2", 3: "b3", 4: "b4", 5: "b5", 6: "b6", 7: "b7", 8: "b8", 9: "b9",
                       12: "h12", 15: "h15", 16: "h16", 18: "h18", 19: "h19", 21: "h21"}


def map_ultralytics_state_dict(sd, net: "YoloV8") -> dict:
    """Rename an Ultralytics-style YOLOv8 state dict (``model.N...`` keys, ``model.22.cv2`` = box branch, ``.cv3`` = class
    branch, ``.dfl`` = the fixed 0..15 projection) to this module's names, with a shape check against ``net``.  Accepts
    the dict of ``YOLO(...).model.state_dict()`` (``model.N.*``) or of its inner ``nn.Sequential`` (``N.*``).  A ``.pt``
    checkpoint of that project is a pickled module, which ``torch.load(weights_only=True)`` rightly refuses: export the
    state dict once with the original tooling and hand THAT file to ``model_path``."""
    own = net.state_dict()
    out = {}
    for key, val in sd.items():
        k = key
        while k.startswith("model."):
            k = k[len("model."):]
        idx, _, rest = k.partition(".")
        if not idx.isdigit():
            raise KeyError(f"'{key}' is not an Ultralytics YOLOv8 parameter name (expected model.<layer>....)")
        i = int(idx)
        if i == 22:
            branch, _, tail = rest.partition(".")
            if branch == "dfl":
                if tuple(val.shape) != (1, 16, 1, 1) or not torch.equal(val.flatten().float(), torch.arange(16.0)):
                    raise ValueError("model.22.dfl is not the fixed 0..15 projection this head computes in closed form")
                continue
            if branch not in ("cv2", "cv3"):
                raise KeyError(f"unknown detect-head entry '{key}'")
            new = ("detect.box." if branch == "cv2" else "detect.cls.") + tail
        elif i in _ULTRALYTICS_LAYERS:
            new = _ULTRALYTICS_LAYERS[i] + "." + rest
        else:
            raise KeyError(f"'{key}': layer {i} has no parameters in YOLOv8 (Upsample / Concat)")
        if new.endswith("num_batches_tracked") and new not in own:
            continue
        if new not in own:
            raise KeyError(f"'{key}' -> '{new}' does not exist in YOLOv8{net.scale} (wrong model scale?)")
        if tuple(own[new].shape) != tuple(val.shape):
            raise ValueError(f"'{key}': shape {tuple(val.shape)} does not fit YOLOv8{net.scale}'s {new} {tuple(own[new].shape)}")
        out[new] = val
    missing = [k for k in own if k not in out and not k.endswith("num_batches_tracked")]
    if missing:
        raise KeyError(f"state dict lacks {len(missing)} parameters of YOLOv8{net.scale}, e.g. {missing[:3]}")
    return out


def load_detector_state_dict(net: "YoloV8", sd) -> "YoloV8":
    """This module's own naming, or Ultralytics naming (detected by its ``model.N`` / ``N`` layer-index keys)."""
    first = next(iter(sd))
    k = first
    while k.startswith("model."):
        k = k[len("model."):]
    if k.split(".", 1)[0].isdigit():
        sd = map_ultralytics_state_dict(sd, net)
        net.load_state_dict(sd, strict=False)        # only num_batches_tracked may be absent
    else:
        net.load_state_dict(sd)
    return net


def build_detector_net(scale: str = "s", seed: int = 0, weights: Optional[str] = None, nc: int = 80) -> YoloV8:
    """Seeded random-init network (torch.manual_seed(seed), BN statistics randomised so that the
    folded convs are not trivial), or a local state-dict when ``weights`` names an existing file
    (loaded with ``weights_only=True``; nothing is ever fetched by name)."""
    g = torch.Generator().manual_seed(seed)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        net = YoloV8(scale, nc)
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    finally:
        torch.random.set_rng_state(state)
    if weights and Path(weights).is_file():
        sd = torch.load(weights, map_location="cpu", weights_only=True)     # never unpickles code
        load_detector_state_dict(net, sd)
    net.eval()
    return net


@torch.no_grad()
def density_shifts(class_logits: torch.Tensor, conf_thr: float, target_per_image: float, iters: int = 24,
                   objectness_quantile: float = 0.95) -> Tuple[float, float]:
    """The two class-bias shifts (class 0, classes >= 1) under which about ``target_per_image`` anchors per image clear
    ``conf_thr`` with the reference's score rule (score = p[class0] * max_k p[class k>=1]).  Class 0
    acts as "objectness": the anchors above ``objectness_quantile`` of its logit are made confident (p = 0.9), the other
    shift is bisected.  Nothing is modified."""
    z = class_logits.float()
    z0, zr = z[:, 0], z[:, 1:].max(1).values
    want = target_per_image * z.shape[0]

    def count(s0, sr):
        return int(((torch.sigmoid(z0 + s0) * torch.sigmoid(zr + sr)) >= conf_thr).sum())

    q = torch.quantile(z0.flatten()[:2_000_000], objectness_quantile).item()
    s0 = math.log(0.9 / 0.1) - q
    lo, hi = -40.0, 40.0
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        if count(s0, mid) > want:
            hi = mid
        else:
            lo = mid
    return s0, 0.5 * (lo + hi)


def apply_class_shifts(net: "YoloV8", s0: float, sr: float) -> None:
    for seq in net.detect.cls:
        seq[-1].bias.data[0] += s0
        seq[-1].bias.data[1:] += sr


@torch.no_grad()
def calibrate_detection_density(net: YoloV8, sample: Optional[torch.Tensor], conf_thr: float, target_per_image: int = 120,
                                iters: int = 24, class_logits: Optional[torch.Tensor] = None) -> Tuple[float, float]:
    """Synthetic-weight helper (bench / smoke only): shift the class-branch biases so that about
    ``target_per_image`` anchors per image clear ``conf_thr`` under the reference's score rule
    (score = p[class0] * max_k p[class k>=1]).  A randomly initialised head
    otherwise emits either nothing or all 8400 anchors, neither of which exercises NMS/tracking
    like a trained detector does.  Returns the two shifts applied (class 0, classes >= 1)."""
    if class_logits is None:
        _, cls = net(sample, raw=True)
    else:
        cls = class_logits           # [B, nc, A] class logits obtained elsewhere (e.g. logit of the fused plan's probabilities)
    s0, sr = density_shifts(cls, conf_thr, target_per_image, iters)
    apply_class_shifts(net, s0, sr)
    return s0, sr

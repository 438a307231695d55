"""Fused YOLOv8 execution plan on the librva detector primitives.

Takes the seeded/loaded ``yolov8.YoloV8`` module (BatchNorm folded) as the weight container and
runs the same graph as a static list of HIP launches on NHWC fp16 buffers:

  * every Conv-BN-SiLU is ONE launch (MFMA implicit GEMM, bias + SiLU [+ residual] epilogue);
  * ``torch.cat`` / ``chunk`` never materialise: producers write channel slices of pre-allocated
    concat buffers and consumers read slices through a row stride (C2f, SPPF, FPN joins);
  * SPPF max-pools, FPN upsampling and the DFL/sigmoid decode are small NHWC kernels;
  * output is the same ``[B, 4+nc, A]`` fp16 tensor the torch module returns, so everything
    downstream (K2/K3/K4) and every parity test is unchanged.

The plan has no host synchronisation and allocates nothing after construction, so a whole tick can
be captured into a hipGraph.  Self-parity against the torch module is a test (fp16 tolerance).
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _native as N
from . import ops
from .yolov8 import C2f, ConvBnAct, SPPF, YoloV8


def _p(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


_LIB_DIGEST: Optional[str] = None


def _lib_digest() -> str:
    """sha256 of librva.so: a rebuilt library never inherits another build's kernel selection."""
    global _LIB_DIGEST
    if _LIB_DIGEST is None:
        import hashlib
        _LIB_DIGEST = hashlib.sha256(N.LIB_PATH.read_bytes()).hexdigest()
    return _LIB_DIGEST


def _tuning_dir():
    import os
    from pathlib import Path
    d = os.environ.get("RVA_TUNE_CACHE_DIR")
    return Path(d) if d else Path(os.environ.get("XDG_CACHE_HOME", str(Path.home() / ".cache"))) / "rva_amd" / "autotune"


class _View:
    """A channel slice of an NHWC buffer: (tensor [M, ld], channel offset, channels)."""

    __slots__ = ("buf", "off", "ch")

    def __init__(self, buf: torch.Tensor, off: int, ch: int):
        self.buf, self.off, self.ch = buf, off, ch

    @property
    def ld(self) -> int:
        return int(self.buf.shape[-1])

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.buf.data_ptr() + 2 * self.off)

    def sub(self, off: int, ch: int) -> "_View":
        assert off + ch <= self.ch
        return _View(self.buf, self.off + off, ch)


class FusedYoloV8:
    def __init__(self, net: YoloV8, batch: int, hw: Tuple[int, int] = (640, 640), device: Optional[torch.device] = None,
                 ctx: Optional[N.Context] = None, autotune: bool = True, tune_overlap: int = 1):
        self.tune_overlap = int(tune_overlap)
        self.ctx = ctx or ops.context()
        self.dev = device or torch.device("cuda", self.ctx.device)
        self.B, self.H, self.W = batch, hw[0], hw[1]
        assert hw[0] % 32 == 0 and hw[1] % 32 == 0
        net = net.fuse()
        self._net = net                     # kept for the twin plan of the in-plan kernel selection
        self.nc = net.nc
        self.L = N.lib()
        self._keep: List[torch.Tensor] = []
        self._steps: List[Callable[[C.c_void_p], None]] = []
        self._lane_of: List[int] = []       # per step: 0 = main stream, k > 0 = side stream k (detect-head branches)
        self._lane = 0
        import os
        self._split_branches = os.environ.get("RVA_HEAD_SPLIT", "0") == "1"     # measured: no gain over one lane per level (profiles/r02_head_lanes_ab.txt)
        self._forks: Dict[int, Tuple[int, int]] = {}   # side lane -> (parent lane, step index at which it forks off the parent)
        self._tunable = []
        self._build(net)
        self._lane_of += [self._lane] * (len(self._steps) - len(self._lane_of))
        self._side = {}                     # lane -> torch.cuda.Stream, created on first use
        import os
        self.concurrent_heads = os.environ.get("RVA_SERIAL_HEADS") != "1"      # A/B switch for measurements
        if autotune:
            self.autotune()

    # -- weight preparation ---------------------------------------------------------------------------
    def _conv_params(self, conv):
        """``conv``: one Conv2d, or a list of Conv2d with equal Cin / kernel / stride whose outputs are concatenated
        along the channel axis (one launch for sibling convolutions that read the same tensor)."""
        convs = list(conv) if isinstance(conv, (list, tuple)) else [conv]
        conv = convs[0]
        assert all(c.kernel_size == conv.kernel_size and c.stride == conv.stride and c.in_channels == conv.in_channels
                   for c in convs)
        w = torch.cat([c.weight.detach().float() for c in convs], 0)        # [Cout, Cin, k, k]
        cout, cin, k, _ = w.shape
        cpad = self.L.rva_conv_cout_pad(cout)
        cinp = (cin + 31) // 32 * 32
        wp = torch.zeros((cpad, k * k, cinp), dtype=torch.float16)
        wp[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, k * k, cin).half()
        bp = torch.zeros((cpad,), dtype=torch.float32)
        o = 0
        for c in convs:
            if c.bias is not None:
                bp[o:o + c.out_channels] = c.bias.detach().float()
            o += c.out_channels
        wp, bp = wp.to(self.dev).contiguous(), bp.to(self.dev).contiguous()
        self._keep += [wp, bp]
        return wp, bp, cin, cout, k, conv.stride[0]

    def _buf(self, m: int, ch: int) -> torch.Tensor:
        t = torch.zeros((m, ch), dtype=torch.float16, device=self.dev)
        self._keep.append(t)
        return t

    # -- step emitters --------------------------------------------------------------------------------
    def _conv(self, mod, src: _View, dst: _View, h: int, w: int, res: Optional[_View] = None):
        mods = list(mod) if isinstance(mod, (list, tuple)) else [mod]
        acts = {1 if isinstance(m, ConvBnAct) and m.act else 0 for m in mods}
        assert len(acts) == 1, "fused sibling convolutions must share the activation"
        act = acts.pop()
        wp, bp, cin, cout, k, stride = self._conv_params([m.conv if isinstance(m, ConvBnAct) else m for m in mods])
        assert cin == src.ch and cout == dst.ch, (cin, src.ch, cout, dst.ch)
        B, L, ctx = self.B, self.L, self.ctx

        state = {"variant": 0}

        def launch(stream, variant):
            return L.rva_conv2d_nhwc_f16_v(ctx.handle, src.ptr, src.ld, _p(wp), _p(bp), dst.ptr, dst.ld,
                                           res.ptr if res else None, res.ld if res else 0, B, h, w, cin, cout, k, stride, act,
                                           variant, stream)

        def run(stream):
            ctx.check(launch(stream, state["variant"]), "rva_conv2d_nhwc_f16")
        self._steps.append(run)
        self._tunable.append((launch, state, f"{cin}->{cout} k{k}s{stride} {h}x{w}"))
        return (h - 1) // stride + 1 if k == 3 else h // stride, (w - 1) // stride + 1 if k == 3 else w // stride

    def _conv_upcat(self, mod, low: _View, skip: _View, dst: _View, h: int, w: int):
        """1x1 convolution of cat([upsample2x(low), skip]) in one launch (rva_conv1x1_upcat_f16): neither the upsampled
        tensor nor the concatenation exists in memory."""
        conv = mod.conv if isinstance(mod, ConvBnAct) else mod
        act = 1 if isinstance(mod, ConvBnAct) and mod.act else 0
        wp, bp, cin, cout, k, stride = self._conv_params(conv)
        assert k == 1 and stride == 1 and cin == low.ch + skip.ch and cout == dst.ch and cin % 32 == 0
        B, L, ctx = self.B, self.L, self.ctx
        state = {"variant": 0}

        def launch(stream, variant):
            if variant and not 33 <= variant <= 39:
                return N.RVA_ERR_ARG
            return L.rva_conv1x1_upcat_f16(ctx.handle, low.ptr, low.ld, low.ch, skip.ptr, skip.ld, skip.ch, _p(wp), _p(bp),
                                           dst.ptr, dst.ld, B, h, w, cout, act, variant, stream)

        def run(stream):
            ctx.check(launch(stream, state["variant"]), "rva_conv1x1_upcat_f16")
        self._steps.append(run)
        self._tunable.append((launch, state, f"up{low.ch}+{skip.ch}->{cout} k1s1 {h}x{w}"))

    def _conv_head(self, conv, src: _View, mode: int, h: int, w: int, a0: int, stride: float):
        """Last 1x1 convolution of a detect-head branch with the head decode as its epilogue (rva_conv1x1_head_f16):
        mode 1 = box branch, mode 2 = class branch; writes rows of ``self.out`` at anchor offset ``a0``."""
        wp, bp, cin, cout, k, st = self._conv_params(conv)
        assert k == 1 and st == 1 and cin == src.ch
        B, L, ctx = self.B, self.L, self.ctx
        state = {"variant": 0}

        def launch(stream, variant):
            if variant and not 33 <= variant <= 39:
                return N.RVA_ERR_ARG
            return L.rva_conv1x1_head_f16(ctx.handle, src.ptr, src.ld, _p(wp), _p(bp), B, h, w, cin, cout, mode, _p(self.out), self.nc,
                                          self.A, a0, C.c_float(stride), variant, stream)

        def run(stream):
            ctx.check(launch(stream, state["variant"]), "rva_conv1x1_head_f16")
        self._steps.append(run)
        self._tunable.append((launch, state, f"head{mode}:{cin}->{cout} k1s1 {h}x{w}"))

    def _c2f(self, mod: C2f, src, dst: _View, h: int, w: int):
        """``src``: a view, or a pair (low, skip) standing for cat([upsample2x(low), skip])."""
        c, n = mod.c, len(mod.m)
        m = self.B * h * w
        cat = _View(self._buf(m, (2 + n) * c), 0, (2 + n) * c)
        if isinstance(src, tuple):
            self._conv_upcat(mod.cv1, src[0], src[1], cat.sub(0, 2 * c), h, w)
        else:
            self._conv(mod.cv1, src, cat.sub(0, 2 * c), h, w)
        tmp = _View(self._buf(m, c), 0, c)
        for i, b in enumerate(mod.m):
            x = cat.sub((1 + i) * c, c)
            self._conv(b.cv1, x, tmp, h, w)
            self._conv(b.cv2, tmp, cat.sub((2 + i) * c, c), h, w, res=x if b.add else None)
        self._conv(mod.cv2, cat, dst, h, w)

    def _sppf(self, mod: SPPF, src: _View, dst: _View, h: int, w: int):
        c_ = mod.cv1.conv.out_channels
        m = self.B * h * w
        cat = _View(self._buf(m, 4 * c_), 0, 4 * c_)
        self._conv(mod.cv1, src, cat.sub(0, c_), h, w)
        B, L, ctx = self.B, self.L, self.ctx
        if h * w <= 2400:            # the three chained pools as one launch (pool9 / pool13 of the same LDS tile)
            x, y1, y2, y3 = (cat.sub(i * c_, c_) for i in range(4))

            def run3(stream):
                ctx.check(L.rva_sppf_pool3_nhwc_f16(ctx.handle, x.ptr, x.ld, y1.ptr, y2.ptr, y3.ptr, y1.ld, B, h, w, c_, stream),
                          "sppf_pool3")
            self._steps.append(run3)
        else:
            for i in range(3):
                s, d = cat.sub(i * c_, c_), cat.sub((i + 1) * c_, c_)

                def run(stream, s=s, d=d):
                    ctx.check(L.rva_maxpool5_nhwc_f16(ctx.handle, s.ptr, s.ld, d.ptr, d.ld, B, h, w, c_, stream), "maxpool5")
                self._steps.append(run)
        self._conv(mod.cv2, cat, dst, h, w)

    def _upsample(self, src: _View, dst: _View, h: int, w: int):
        B, L, ctx = self.B, self.L, self.ctx

        def run(stream):
            ctx.check(L.rva_upsample2x_nhwc_f16(ctx.handle, src.ptr, src.ld, dst.ptr, dst.ld, B, h, w, src.ch, stream), "upsample")
        self._steps.append(run)

    def _set_lane(self, lane: int) -> None:
        """Steps appended from now on run on side stream ``lane`` (0 = main).  A side lane forks off the main stream at the
        point of this call (everything appended before it on the main lane is its dependency) and joins at the end."""
        self._lane_of += [self._lane] * (len(self._steps) - len(self._lane_of))
        self._lane = lane

    # -- the graph ------------------------------------------------------------------------------------
    def _build(self, net: YoloV8):
        B, H, W = self.B, self.H, self.W
        c1 = net.b0.conv.out_channels; c2 = net.b1.conv.out_channels; c3 = net.b3.conv.out_channels
        c4 = net.b5.conv.out_channels; c5 = net.b7.conv.out_channels
        h1, w1, h2, w2 = H // 2, W // 2, H // 4, W // 4
        h3, w3, h4, w4, h5, w5 = H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
        # stem (planar input from K1)
        sw = torch.zeros((64, 32), dtype=torch.float16)
        w0 = net.b0.conv.weight.detach().float().cpu().reshape(c1, 9, 3)        # [co][j = c*3+ky][kx]
        sw[:c1, 0:18] = w0[:, :, 0:2].reshape(c1, 18).half()                     # k = 2*j + kx, kx in {0,1}
        sw[:c1, 18:27] = w0[:, :, 2].half()                                      # k = 18 + j, kx = 2
        sb = torch.zeros((64,), dtype=torch.float32)
        sb[:c1] = net.b0.conv.bias.detach().float().cpu()
        sw, sb = sw.to(self.dev), sb.to(self.dev)
        self._keep += [sw, sb]
        L, ctx = self.L, self.ctx
        self._in_ptr = None
        x1 = _View(self._buf(B * h2 * w2, c2), 0, c2)
        import os
        self.fused_stem = (c1, c2) == (32, 64) and isinstance(net.b1, ConvBnAct) and net.b1.act \
            and os.environ.get("RVA_NO_STEM2", "0") != "1"
        if self.fused_stem:
            # YOLOv8s widths: stem + first downsampling convolution in one launch, the 32-channel half-resolution tensor
            # (210 MB at batch 32) stays in LDS
            wp1, bp1, _, _, _, _ = self._conv_params([net.b1.conv])

            def stem2(stream):
                ctx.check(L.rva_stem2_f16(ctx.handle, self._in_ptr, _p(sw), _p(sb), _p(wp1), _p(bp1),
                                          x1.ptr, x1.ld, B, H, W, stream), "stem2")
            self._steps.append(stem2)
        else:
            x0 = _View(self._buf(B * h1 * w1, c1), 0, c1)

            def stem(stream):
                ctx.check(L.rva_stem_conv_f16(ctx.handle, self._in_ptr, _p(sw), _p(sb),
                                              x0.ptr, x0.ld, B, H, W, c1, stream), "stem")
            self._steps.append(stem)
            self._conv(net.b1, x0, x1, h1, w1)
        x2 = _View(self._buf(B * h2 * w2, c2), 0, c2)
        self._c2f(net.b2, x1, x2, h2, w2)
        # concat buffers of the neck: producers write their slice directly
        cat15 = _View(self._buf(B * h3 * w3, c4 + c3), 0, c4 + c3)     # [up(n4) | p3]
        cat12 = _View(self._buf(B * h4 * w4, c5 + c4), 0, c5 + c4)     # [up(p5) | p4]
        cat18 = _View(self._buf(B * h4 * w4, c3 + c4), 0, c3 + c4)     # [h16(n3) | n4]
        cat21 = _View(self._buf(B * h5 * w5, c4 + c5), 0, c4 + c5)     # [h19(m4) | p5]
        p3, p4, n4, p5 = cat15.sub(c4, c3), cat12.sub(c5, c4), cat18.sub(c3, c4), cat21.sub(c4, c5)
        t3 = _View(self._buf(B * h3 * w3, c3), 0, c3)
        self._conv(net.b3, x2, t3, h2, w2)
        self._c2f(net.b4, t3, p3, h3, w3)
        t4 = _View(self._buf(B * h4 * w4, c4), 0, c4)
        self._conv(net.b5, p3, t4, h3, w3)
        self._c2f(net.b6, t4, p4, h4, w4)
        t5 = _View(self._buf(B * h5 * w5, c5), 0, c5)
        # from here on the backbone works at 20x20 (b7, b8, SPPF): a quarter of a millisecond in which most CUs and most of
        # the HBM bandwidth are idle -- a pipelined runner may start the NEXT tick's K1 here (``phase_event``)
        self.quiet_step = len(self._steps)
        self._conv(net.b7, p4, t5, h4, w4)
        t5b = _View(self._buf(B * h5 * w5, c5), 0, c5)
        self._c2f(net.b8, t5, t5b, h5, w5)
        self._sppf(net.b9, t5b, p5, h5, w5)
        fuse_up = c5 % 64 == 0 and c4 % 64 == 0 and c3 % 64 == 0 and h4 == 2 * h5 and w4 == 2 * w5 and h3 == 2 * h4 and w3 == 2 * w4
        n3 = _View(self._buf(B * h3 * w3, c3), 0, c3)
        if fuse_up:      # FPN top-down path: upsample + concat folded into the consuming 1x1 convolutions
            self._c2f(net.h12, (p5, p4), n4, h4, w4)
            self._c2f(net.h15, (n4, p3), n3, h3, w3)
        else:
            self._upsample(p5, cat12.sub(0, c5), h5, w5)
            self._c2f(net.h12, cat12, n4, h4, w4)
            self._upsample(n4, cat15.sub(0, c4), h4, w4)
            self._c2f(net.h15, cat15, n3, h3, w3)
        fork_n3 = len(self._steps)                    # n3 is complete here: the stride-8 detect branch may start
        self._conv(net.h16, n3, cat18.sub(0, c3), h3, w3)
        m4 = _View(self._buf(B * h4 * w4, c4), 0, c4)
        self._c2f(net.h18, cat18, m4, h4, w4)
        fork_m4 = len(self._steps)                    # m4 is complete: the stride-16 detect branch may start
        self._conv(net.h19, m4, cat21.sub(0, c4), h4, w4)
        m5 = _View(self._buf(B * h5 * w5, c5), 0, c5)
        self._c2f(net.h21, cat21, m5, h5, w5)
        # detect head
        A = h3 * w3 + h4 * w4 + h5 * w5
        self.A = A
        self.out = torch.empty((B, 4 + self.nc, A), dtype=torch.float16, device=self.dev)
        a0 = 0
        levels = []
        # the last 1x1 convolution of each branch can decode straight into the result tensor (no logits in HBM, no head
        # kernel) when its input width suits the LDS-DMA gather kernel
        fuse_head = all(net.detect.box[l][2].in_channels % 64 == 0 and net.detect.cls[l][2].in_channels % 64 == 0 and
                        net.detect.box[l][2].out_channels == 64 and net.detect.cls[l][2].out_channels == self.nc for l in range(3))
        for lvl, (feat, hh, ww, stride) in enumerate(((n3, h3, w3, 8.0), (m4, h4, w4, 16.0), (m5, h5, w5, 32.0))):
            # The three detect branches only depend on their own feature map and write disjoint anchor ranges of the
            # result: the stride-8 branch (the big one) runs on a side stream beside the rest of the neck -- h16 ... h21
            # are 40x40 / 20x20 layers that leave most CUs idle -- and the stride-16 branch beside h19 / h21; they join
            # the main stream at the end of the plan.  Only when the decode is fused into the branches (no k_head3).
            primary = lvl + 1 if (fuse_head and lvl < 2) else 0
            self._set_lane(primary)
            if primary:
                self._forks[primary] = (0, fork_n3 if lvl == 0 else fork_m4)
            box, cls = net.detect.box[lvl], net.detect.cls[lvl]
            m = B * hh * ww
            cb = box[0].conv.out_channels
            cc = cls[0].conv.out_channels
            # the first convolution of the box and of the class branch read the same feature map: one launch with
            # Cout = cb + cc writing one buffer, the branches continue on its channel slices
            first = _View(self._buf(m, cb + cc), 0, cb + cc)
            b1, k1 = first.sub(0, cb), first.sub(cb, cc)
            b2 = _View(self._buf(m, cb), 0, cb)
            bo = _View(self._buf(m, 64), 0, 64)
            k2 = _View(self._buf(m, cc), 0, cc)
            ko = _View(self._buf(m, self.nc), 0, self.nc)
            self._conv([box[0], cls[0]], feat, first, hh, ww)
            if fuse_head:
                # box and class sub-branches are independent after the shared first convolution: the class one gets a
                # lane of its own
                fork_cls = len(self._steps)
                self._conv(box[1], b1, b2, hh, ww); self._conv_head(box[2], b2, 1, hh, ww, a0, stride)
                if self._split_branches:
                    self._set_lane(4 + lvl)
                    self._forks[4 + lvl] = (primary, fork_cls)
                self._conv(cls[1], k1, k2, hh, ww); self._conv_head(cls[2], k2, 2, hh, ww, a0, stride)
                self._set_lane(primary)
            else:
                self._conv(box[1], b1, b2, hh, ww); self._conv(box[2], b2, bo, hh, ww)
                self._conv(cls[1], k1, k2, hh, ww); self._conv(cls[2], k2, ko, hh, ww)
            levels.append((bo, ko, hh, ww, stride))
            a0 += hh * ww
        self._set_lane(0)
        if fuse_head:
            return
        # DFL + dist2bbox + sigmoid of the three levels in one launch
        bp, _k1 = N.ptr_array([lv[0].ptr.value for lv in levels]); kp, _k2 = N.ptr_array([lv[1].ptr.value for lv in levels])
        lb, _k3 = N.i32_array([lv[0].ld for lv in levels]); lc, _k4 = N.i32_array([lv[1].ld for lv in levels])
        hs, _k5 = N.i32_array([lv[2] for lv in levels]); ws, _k6 = N.i32_array([lv[3] for lv in levels])
        st = (C.c_float * 3)(*[lv[4] for lv in levels])
        self._keep += [_k1, _k2, _k3, _k4, _k5, _k6, st]
        B_, nc = B, self.nc

        def head(stream):
            ctx.check(L.rva_yolo_head3_f16(ctx.handle, bp, lb, kp, lc, _p(self.out), B_, hs, ws, nc, A, st, stream), "yolo_head3")
        self._steps.append(head)

    # -- per-layer kernel selection ---------------------------------------------------------------------
    # -- persisted kernel selection -----------------------------------------------------------------------
    def _tuning_key(self) -> str:
        """What a kernel selection depends on: the device, the library build, the layer shapes of the plan (batch included)
        and the switches that change how the selection is made."""
        import hashlib
        import os
        h = hashlib.sha256()
        h.update(torch.cuda.get_device_name(self.dev).encode())
        h.update(_lib_digest().encode())
        h.update(repr((self.B, self.H, self.W, [d for _, _, d in self._tunable])).encode())
        h.update(b"per-layer times; in-plan pass opt-in, three overlapping passes")
        h.update(repr([os.environ.get(k, "") for k in ("RVA_SKIP_VARIANTS", "RVA_TUNE_IN_PLAN", "RVA_TUNE_OVERLAP", "RVA_TUNE_TOP",
                                                       "RVA_TUNE_WITHIN", "RVA_NO_STEM2", "RVA_HEAD_SPLIT")]).encode())
        h.update(repr(int(os.environ.get("RVA_TUNE_LAYER_OVERLAP", getattr(self, "tune_overlap", 1)))).encode())
        return h.hexdigest()[:24]

    def _load_tuning(self) -> bool:
        import json
        f = _tuning_dir() / f"{self._tuning_key()}.json"
        try:
            rec = json.loads(f.read_text())
            picks = rec["variants"]
            if len(picks) != len(self._tunable) or any(d != pd for (_, _, d), (pd, _) in zip(self._tunable, picks)):
                return False
        except (OSError, ValueError, KeyError, TypeError):
            return False
        for (_, state, _), (_, v) in zip(self._tunable, picks):
            state["variant"] = int(v)
        self.tuning = [tuple(t) for t in rec.get("tuning", [])]
        self.tuning_source = str(f)
        return True

    def _store_tuning(self) -> None:
        import json
        import os
        d = _tuning_dir()
        try:
            d.mkdir(parents=True, exist_ok=True)
            tmp = d / f".{self._tuning_key()}.{os.getpid()}.tmp"
            tmp.write_text(json.dumps({"device": torch.cuda.get_device_name(self.dev), "batch": self.B, "hw": [self.H, self.W],
                                       "variants": [[desc, st["variant"]] for _, st, desc in self._tunable],
                                       "tuning": [list(t) for t in self.tuning]}))
            os.replace(tmp, d / f"{self._tuning_key()}.json")            # atomic: concurrent ranks write the same content
        except OSError:
            pass                                                        # a read-only cache directory costs start-up time only

    def autotune(self, reps: int = 5) -> None:
        """Time every applicable conv kernel variant on each layer's real shape (buffers hold whatever they
        hold: timing only) and keep the fastest.  All variants compute the same sums in fp32 with the same
        per-chunk order of K, so the choice does not change results beyond fp32 summation order.  The selection is
        persisted per (device, library build, plan shape): a later process takes it over instead of timing again
        (``RVA_TUNE_CACHE=0`` disables, ``RVA_TUNE_CACHE_DIR`` moves it)."""
        import os
        self.tuning_source = "measured"
        use_cache = os.environ.get("RVA_TUNE_CACHE", "1") == "1"
        if use_cache and self._load_tuning():
            return
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.tuning = []
        self._candidates = {}
        self._n_variants = int(N.lib().rva_conv_num_variants())
        skip = {int(v) for v in os.environ.get("RVA_SKIP_VARIANTS", "").replace(",", " ").split()}      # tuning aid: same-box A/B of kernel families
        cache = {}
        # What a launch costs a pipeline with several ticks in flight is not its time alone on the chip but the share of the chip it
        # holds for that time: a one-workgroup-per-CU kernel (120-150 KB of LDS) that is 15 % faster alone than a two-per-CU one
        # leaves no room for the other chains' kernels while it runs.  tune_overlap = n >= 2 times every candidate as n launches
        # side by side on the chains' own streams (behind a spin kernel, so that the host's launch rate stays out of the
        # measurement) and keeps the variant with the lowest time PER LAUNCH; 1 = the launch alone (lowest latency: one tick at a
        # time, the paced operating point).  PipelinedTicks sets it to its depth; RVA_TUNE_LAYER_OVERLAP overrides.
        n_over = int(os.environ.get("RVA_TUNE_LAYER_OVERLAP", getattr(self, "tune_overlap", 1)))
        lanes = ops.chain_streams(self.dev, n_over) if n_over >= 2 else []
        spin = int(os.environ.get("RVA_TUNE_SPIN_CYCLES", "400000"))       # ~0.2 ms: every stream's launches are queued before any starts

        def time_variant(variant) -> float:
            torch.cuda.synchronize()
            if not lanes:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    launch(stream, variant)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps * 1e3
            evs = []
            for st in lanes:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    sp = C.c_void_p(st.cuda_stream)
                    for _ in range(reps):
                        launch(sp, variant)
                    b.record()
                    evs.append((a, b))
            torch.cuda.synchronize()
            # from the first stream's start to the last stream's end (the spins end within microseconds of each other)
            first = evs[0][0]
            return max(first.elapsed_time(b) for _, b in evs) / (reps * len(lanes)) * 1e3
        for launch, state, desc in self._tunable:
            if desc in cache:
                state["variant"] = cache[desc][0]
                continue
            best = (0, float("inf"))
            timed = []
            for variant in range(1, self._n_variants + 1):
                if variant in skip or launch(stream, variant) != N.RVA_OK:
                    continue
                us = time_variant(variant)
                timed.append((us, variant))
                if us < best[1]:
                    best = (variant, us)
            state["variant"] = best[0]
            cache[desc] = best
            self._candidates[desc] = sorted(timed)[:int(os.environ.get('RVA_TUNE_TOP', '5'))]
            self.tuning.append((desc, best[0], round(best[1], 1)))
        # The second pass (whole forward passes, three in flight, candidates within 25 % swapped in one by one) is opt-in since
        # round 3: with an honest base timing it moves the three-chain pipeline by -0.8 % +- 1 % (profiles/r03_experiments_not_kept.txt #8)
        if os.environ.get("RVA_TUNE_IN_PLAN", "0") == "1":
            self._refine_in_plan()
        if use_cache:
            self._store_tuning()

    def copy_tuning(self, other: "FusedYoloV8") -> None:
        """Take over the kernel selection of a plan built from the same network and batch shape."""
        assert len(self._tunable) == len(other._tunable)
        for (_, mine, d1), (_, theirs, d2) in zip(self._tunable, other._tunable):
            assert d1 == d2
            mine["variant"] = theirs["variant"]
        self.tuning = list(getattr(other, "tuning", []))
        self.tuning_source = "copied from the first slot's plan"

    def _refine_in_plan(self, reps: int = 12, within: float = 1.25) -> None:
        import os
        within = float(os.environ.get('RVA_TUNE_WITHIN', within))
        """Second pass of the kernel selection, on the real objective: a layer's launch time in isolation is not its cost inside
        the plan -- the detect branches run on side streams beside the neck, and a kernel that wants every CU for itself (one
        128-KB workgroup per CU) shares worse than a two-workgroups-per-CU one that is a few per cent slower alone.  For the
        layers with runners-up within 25 % the whole forward pass is timed with each candidate (coordinate descent, most
        expensive layers first) and a candidate is kept when the pass gets faster by more than the timing noise."""
        x = torch.zeros((self.B, 3, self.H, self.W), dtype=torch.float16, device=self.dev)
        # The objective is what the pipeline does with the plan: forward passes of consecutive ticks rotate over three streams
        # and overlap (PipelinedTicks, depth 3), so the pass is timed as a group -- this plan on one stream, twins with the same
        # kernel selection on the others.  RVA_TUNE_OVERLAP=n: n passes in flight (0 / 1: a single pass alone).
        twins: list = []
        lanes = self.concurrent_heads
        n_over = int(os.environ.get("RVA_TUNE_OVERLAP", "3"))
        if n_over >= 2:
            twins = [FusedYoloV8(self._net, self.B, (self.H, self.W), device=self.dev, ctx=self.ctx, autotune=False) for _ in range(n_over - 1)]
            from .ops import chain_streams
            streams = chain_streams(self.dev, n_over)                  # the streams the pipeline's tick chains will run on
            self.concurrent_heads = False                              # as PipelinedTicks runs overlapping passes: branches in line
            for t in twins:
                t.concurrent_heads = False

        def forward_us() -> float:
            best = float("inf")
            for t in twins:
                t.copy_tuning(self)
            group = [self] + twins
            rounds = max(reps // len(group), 2)
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if not twins:
                    for _ in range(reps):
                        self(x)
                else:
                    for _ in range(rounds):
                        for pl, st in zip(group, streams):
                            with torch.cuda.stream(st):
                                pl(x)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (rounds * len(group) if twins else reps) * 1e6)
            return best
        by_desc: Dict[str, list] = {}
        for _, state, desc in self._tunable:
            by_desc.setdefault(desc, []).append(state)
        cost = {d: v[0][0] * len(by_desc[d]) for d, v in self._candidates.items() if v}
        order = [d for d in sorted(cost, key=cost.get, reverse=True)
                 if len(self._candidates[d]) > 1 and self._candidates[d][1][0] <= within * self._candidates[d][0][0]][:24]
        if not order:
            self.concurrent_heads = lanes
            return
        self(x); self(x)
        forward_us()                                               # the twins' first passes (lazy buffers, cold caches) stay out of the base
        base = forward_us()
        self.refined = []
        for _sweep in range(2):
            changed = False
            for d in order:
                cur = by_desc[d][0]["variant"]
                for us, v in self._candidates[d]:
                    if us > within * self._candidates[d][0][0]:
                        break
                    if v == cur:
                        continue
                    for st in by_desc[d]:
                        st["variant"] = v
                    t = forward_us()
                    if t < base * 0.997:
                        self.refined.append((d, cur, v, round(base, 1), round(t, 1)))
                        base, cur, changed = t, v, True
                    else:
                        for st in by_desc[d]:
                            st["variant"] = cur
            if not changed:
                break
        self.concurrent_heads = lanes
        iso = {d: {v: us for us, v in c} for d, c in self._candidates.items()}
        self.tuning = [(d, by_desc[d][0]["variant"], round(iso[d].get(by_desc[d][0]["variant"], us), 1)) for d, _, us in self.tuning]

    # -- run ------------------------------------------------------------------------------------------
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """``x``: fp16 planar ``[B,3,H,W]`` contiguous (what K1 writes).  Returns ``[B, 4+nc, A]`` fp16."""
        assert x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and tuple(x.shape) == (self.B, 3, self.H, self.W)
        self._in_ptr = C.c_void_p(x.data_ptr())
        main = torch.cuda.current_stream()
        stream = C.c_void_p(main.cuda_stream)
        if not (self.concurrent_heads and self._forks):
            ev = getattr(self, "phase_event", None)
            for i, step in enumerate(self._steps):
                if ev is not None and i == self.quiet_step:
                    ev.record(main)                        # the pass enters its 20x20 phase
                step(stream)
            return self.out
        # fork / join over events: works eagerly and inside a stream capture (the side streams join the capture through
        # the fork events and leave it through the join events)
        for lane in self._forks:
            if lane not in self._side:
                self._side[lane] = (torch.cuda.Stream(device=self.dev), torch.cuda.Event(), torch.cuda.Event())
        started = set()

        def stream_of(lane):
            return main if lane == 0 else self._side[lane][0]
        for i, step in enumerate(self._steps):
            for lane, (parent, pt) in self._forks.items():
                if pt == i:
                    self._side[lane][1].record(stream_of(parent))  # everything the parent lane has been given so far is done
            lane = self._lane_of[i]
            if lane == 0:
                step(stream)
            else:
                side, fork_ev, _ = self._side[lane]
                if lane not in started:
                    side.wait_event(fork_ev)
                    started.add(lane)
                step(C.c_void_p(side.cuda_stream))
        for lane in started:
            side, _, join_ev = self._side[lane]
            join_ev.record(side)
            main.wait_event(join_ev)
        return self.out

    def use_output(self, index: int) -> torch.Tensor:
        """Select which result tensor the head kernels write (``index`` 0 is the one allocated at construction, others are
        allocated on first use).  A pipelined caller alternates two per tick (and uses a pair per frame group), so the
        post-process of tick k can still read its head tensor on another HIP stream while the network of tick k+1 runs."""
        if not hasattr(self, "_outs"):
            self._outs = {0: self.out}
        if index not in self._outs:
            self._outs[index] = torch.empty_like(self._outs[0])
        self.out = self._outs[index]
        return self.out

    @property
    def n_launches(self) -> int:
        return len(self._steps)

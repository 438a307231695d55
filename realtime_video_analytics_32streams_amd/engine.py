"""Fused YOLOv8 execution plan: the tuner and a thin caller of ``rva_yolov8_plan_*`` (include/rva.h).

The plan itself -- weight packing, the NHWC fp16 activation buffers, the static list of launches (one per Conv-BN-SiLU;
``torch.cat`` / ``chunk`` / upsample never materialise), the fork / join of the detect branches -- is built and replayed in C
(``csrc/rva_plan.hip``): one ABI call per forward pass.  What stays here:

  * handing the seeded / loaded ``yolov8.YoloV8`` module (BatchNorm folded) to ``rva_yolov8_plan_create`` as a list of
    convolutions in module order;
  * the per-layer kernel selection: every applicable variant of every convolution step is timed on the plan's own buffers
    through ``rva_yolov8_plan_launch_tunable`` and the choice fixed with ``rva_yolov8_plan_set_variant``; the selection is
    persisted per (device, library build, plan shape);
  * the result tensors (a pipelined caller alternates several).

Output: the same ``[B, 4+nc, A]`` fp16 tensor the torch module returns, so everything downstream (K2/K3/K4) and every parity
test is unchanged.  No host synchronisation and no allocation after construction: a whole tick can be captured into a hipGraph.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from . import ops
from .yolov8 import C2f, ConvBnAct, YoloV8


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


def module_order_convs(net: YoloV8) -> list:
    """The convolutions of a fused ``YoloV8`` in the module order ``rva_yolov8_plan_create`` consumes (include/rva.h)."""
    def c2f(m: C2f):
        out = [m.cv1, m.cv2]
        for b in m.m:
            out += [b.cv1, b.cv2]
        return out
    mods = [net.b0, net.b1, *c2f(net.b2), net.b3, *c2f(net.b4), net.b5, *c2f(net.b6), net.b7, *c2f(net.b8), net.b9.cv1, net.b9.cv2,
            *c2f(net.h12), *c2f(net.h15), net.h16, *c2f(net.h18), net.h19, *c2f(net.h21)]
    for branch in (net.detect.box, net.detect.cls):
        for seq in branch:
            mods += [seq[0], seq[1], seq[2]]
    return [m.conv if isinstance(m, ConvBnAct) else m for m in mods]


class _VariantCell:
    """``state["variant"]`` of one tunable step, stored in the C plan."""

    __slots__ = ("eng", "idx")

    def __init__(self, eng, idx):
        self.eng, self.idx = eng, idx

    def __getitem__(self, key):
        assert key == "variant"
        return int(self.eng.L.rva_yolov8_plan_get_variant(self.eng.handle, self.idx))

    def __setitem__(self, key, value):
        assert key == "variant"
        self.eng.ctx.check(self.eng.L.rva_yolov8_plan_set_variant(self.eng.handle, self.idx, int(value)), "rva_yolov8_plan_set_variant")


class FusedYoloV8:
    def __init__(self, net: YoloV8, batch: int, hw: Tuple[int, int] = (640, 640), device: Optional[torch.device] = None,
                 ctx: Optional[N.Context] = None, autotune: bool = True, tune_overlap: int = 1):
        import os
        self.tune_overlap = int(tune_overlap)
        self.ctx = ctx or ops.context()
        self.dev = device or torch.device("cuda", self.ctx.device)
        self.B, self.H, self.W = batch, hw[0], hw[1]
        assert hw[0] % 32 == 0 and hw[1] % 32 == 0
        net = net.fuse()
        self._net = net                     # kept for the twin plans of the in-plan kernel selection
        self.nc = net.nc
        self.L = N.lib()
        convs = module_order_convs(net)
        keep, arr = [], (N.ConvWeights * len(convs))()
        for i, c in enumerate(convs):
            w = np.ascontiguousarray(c.weight.detach().float().cpu().numpy())
            b = None if c.bias is None else np.ascontiguousarray(c.bias.detach().float().cpu().numpy())
            keep += [w, b]
            arr[i].weight = w.ctypes.data_as(C.POINTER(C.c_float))
            arr[i].bias = b.ctypes.data_as(C.POINTER(C.c_float)) if b is not None else None
            arr[i].cout, arr[i].cin, arr[i].k, arr[i].stride = int(w.shape[0]), int(w.shape[1]), int(w.shape[2]), int(c.stride[0])
        d = N.YoloV8Desc()
        d.batch, d.height, d.width = batch, hw[0], hw[1]
        d.widths[:] = [net.b0.conv.out_channels, net.b1.conv.out_channels, net.b3.conv.out_channels, net.b5.conv.out_channels,
                       net.b7.conv.out_channels]
        d.depth_backbone[:] = [len(net.b2.m), len(net.b4.m), len(net.b6.m), len(net.b8.m)]
        d.depth_head = len(net.h12.m)
        assert len(net.h15.m) == len(net.h18.m) == len(net.h21.m) == d.depth_head
        d.nc, d.reg_max, d.n_convs = net.nc, net.detect.reg_max, len(convs)
        d.flags = (N.RVA_PLAN_NO_STEM2 if os.environ.get("RVA_NO_STEM2", "0") == "1" else 0) | \
                  (N.RVA_PLAN_NO_CIN_PAD if os.environ.get("RVA_NO_CIN_PAD", "0") == "1" else 0) | \
                  (0 if self._use_pair32() else N.RVA_PLAN_NO_PAIR32)
        h = C.c_void_p()
        with torch.cuda.device(self.dev):
            self.ctx.check(self.L.rva_yolov8_plan_create(self.ctx.handle, C.byref(d), arr, C.byref(h)), "rva_yolov8_plan_create")
        self.handle = h
        del keep
        info = [C.c_int32() for _ in range(5)]
        self.ctx.check(self.L.rva_yolov8_plan_info(h, *[C.byref(v) for v in info]), "rva_yolov8_plan_info")
        self.A, rows, self._n_steps, n_tun, self.quiet_step = (int(v.value) for v in info)
        assert rows == 4 + self.nc
        self.out = torch.empty((batch, rows, self.A), dtype=torch.float16, device=self.dev)
        self._outs = {0: self.out}
        self.fused_stem = not (d.flags & N.RVA_PLAN_NO_STEM2) and tuple(d.widths[:2]) == (32, 64)
        # (launch(stream, variant) -> rc, state["variant"], "Cin->Cout kKsS HxW") per convolution step, as the tuner and the tools use them
        self._tunable = []
        buf = C.create_string_buffer(96)
        for i in range(n_tun):
            self.ctx.check(self.L.rva_yolov8_plan_tunable_desc(h, i, buf, 96), "rva_yolov8_plan_tunable_desc")

            def launch(stream, variant, i=i):
                return self.L.rva_yolov8_plan_launch_tunable(self.handle, i, variant, C.c_void_p(self.out.data_ptr()), stream)
            self._tunable.append((launch, _VariantCell(self, i), buf.value.decode()))
        self._side = None                   # two side streams for the detect branches, created on first use
        self._forks = bool(self.fused_head_lanes())
        self.concurrent_heads = os.environ.get("RVA_SERIAL_HEADS") != "1"      # A/B switch for measurements
        if autotune:
            self.autotune()

    def fused_head_lanes(self) -> bool:
        """Whether the plan has detect branches that may run on side streams (head decode fused into the branches)."""
        return any(d.startswith("head1:") for _, _, d in self._tunable)

    def __del__(self):  # best effort
        try:
            if getattr(self, "handle", None):
                self.L.rva_yolov8_plan_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001
            pass

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
                                                       "RVA_TUNE_WITHIN", "RVA_NO_STEM2", "RVA_HEAD_SPLIT", "RVA_NO_CIN_PAD")]).encode())
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
        try:
            for (_, state, _), (_, v) in zip(self._tunable, picks):
                state["variant"] = int(v)          # the plan checks the variant number against this build (a stale or edited file)
        except RuntimeError:
            for _, state, _ in self._tunable:
                state["variant"] = 0
            return False
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
        """``x``: fp16 planar ``[B,3,H,W]`` contiguous (what K1 writes).  Returns ``[B, 4+nc, A]`` fp16.  One ABI call."""
        assert x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and tuple(x.shape) == (self.B, 3, self.H, self.W)
        main = torch.cuda.current_stream()
        stream = C.c_void_p(main.cuda_stream)
        xin, out = C.c_void_p(x.data_ptr()), C.c_void_p(self.out.data_ptr())
        if self.concurrent_heads and self._forks:
            # fork / join over events inside the plan: works eagerly and inside a stream capture
            if self._side is None:
                self._side = (torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev))
            self.ctx.check(self.L.rva_yolov8_plan_run_lanes(self.handle, xin, out, stream, C.c_void_p(self._side[0].cuda_stream),
                                                            C.c_void_p(self._side[1].cuda_stream)), "rva_yolov8_plan_run_lanes")
            return self.out
        ev = getattr(self, "phase_event", None)
        if ev is not None:
            self.ctx.check(self.L.rva_yolov8_plan_run_range(self.handle, xin, out, 0, self.quiet_step, stream), "rva_yolov8_plan_run_range")
            ev.record(main)                                # the pass enters its 20x20 phase
            self.ctx.check(self.L.rva_yolov8_plan_run_range(self.handle, xin, out, self.quiet_step, self._n_steps, stream),
                           "rva_yolov8_plan_run_range")
        else:
            self.ctx.check(self.L.rva_yolov8_plan_run(self.handle, xin, out, stream), "rva_yolov8_plan_run")
        return self.out

    def use_output(self, index: int) -> torch.Tensor:
        """Select which result tensor the head kernels write (``index`` 0 is the one allocated at construction, others are
        allocated on first use).  A pipelined caller alternates two per tick (and uses a pair per frame group), so the
        post-process of tick k can still read its head tensor on another HIP stream while the network of tick k+1 runs."""
        if index not in self._outs:
            self._outs[index] = torch.empty_like(self._outs[0])
        self.out = self._outs[index]
        return self.out

    def _use_pair32(self) -> bool:
        """The 32-channel C2f bottleneck (YOLOv8s at 160 x 160) as ONE launch with the intermediate in LDS (rva_c2f_pair32_f16) or as
        two convolution launches.  Fused, a forward pass alone is 2.2 % shorter (1.749 against 1.787 ms, bit-identical output) -- less
        memory traffic in a bandwidth-bound stage -- but it recomputes the intermediate's halo (+30 % SiLUs and MFMAs in that stage),
        and with four forward passes side by side, where instruction issue and not bandwidth is what is short, the pipeline loses
        0.5 % (22.36-22.41 k against 22.47-22.55 k frames/s, same box, three alternating pairs).  So: fused for a plan that serves one
        tick at a time (tune_overlap <= 1: `TickPipeline.tick()`, the per-frame API, depth-1 runners), two launches for a plan that
        serves overlapping ticks.  ``RVA_PAIR32=1 / 0`` forces either."""
        env = os.environ.get("RVA_PAIR32")
        if env in ("0", "1"):
            return env == "1"
        return int(os.environ.get("RVA_TUNE_LAYER_OVERLAP", self.tune_overlap)) <= 1

    @property
    def n_launches(self) -> int:
        return self._n_steps

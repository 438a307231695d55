"""ctypes binding of the CPU oracle (oracle/rva_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB: Optional[C.CDLL] = None


def build(force: bool = False) -> Path:
    """liborc.so, or -- RVA_ORACLE_LIB=liborc_asan.so, with libasan / libubsan preloaded -- the AddressSanitizer + UBSan build
    of the same source (``make asan``; tests/test_oracle_golden.py replays the goldens under it in a child process)."""
    name = os.environ.get("RVA_ORACLE_LIB", "liborc.so")
    so = _HERE / name
    src = _HERE / "rva_oracle.c"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.check_call(["make", "-s", "-C", str(_HERE), "asan" if name == "liborc_asan.so" else "liborc.so"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(str(build()))
        L.orc_tracker_new.restype = C.c_void_p
        L.orc_tracker_new.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_tracker_free.argtypes = [C.c_void_p]
        L.orc_tracker_next_id.restype = C.c_int64
        L.orc_tracker_next_id.argtypes = [C.c_void_p]
        L.orc_tracker_set_next_id.argtypes = [C.c_void_p, C.c_int64]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def letterbox(w, h, tw=640, th=640):
    s = C.c_double(); nw = C.c_int(); nh = C.c_int(); l = C.c_int(); t = C.c_int()
    lib().orc_letterbox(w, h, tw, th, C.byref(s), C.byref(nw), C.byref(nh), C.byref(l), C.byref(t))
    return dict(scale=s.value, new=(nw.value, nh.value), pad=(l.value, t.value))


def postprocess(raw2d: np.ndarray, conf_thr: float, iou_thr: float, classes: Optional[Sequence[int]],
                orig_wh, input_wh=(640, 640)):
    """raw2d: float32 [d1, d2]; orientation decided like detector.py:282-283."""
    raw2d = np.ascontiguousarray(raw2d, dtype=np.float32)
    d1, d2 = raw2d.shape
    if lib().orc_head_is_channel_major(d1, d2):
        Cn, A, sa, sc = d1, d2, 1, d2
    else:
        A, Cn, sa, sc = d1, d2, d2, 1
    lb = letterbox(orig_wh[0], orig_wh[1], input_wh[0], input_wh[1])
    cls_arr = np.asarray(classes if classes else [], dtype=np.int32)
    anchor = np.empty(A, np.int32); keep = np.empty(A, np.int32); cls = np.empty(A, np.int32)
    conf = np.empty(A, np.float32); box = np.empty((A, 4), np.float32); ncand = C.c_int()
    L = lib()
    L.orc_postprocess.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_double, C.c_double,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orc_postprocess(raw2d.ctypes.data, A, Cn, sa, sc, conf_thr, iou_thr,
                          cls_arr.ctypes.data if len(cls_arr) else None, len(cls_arr),
                          orig_wh[0], orig_wh[1], lb["scale"], lb["pad"][0], lb["pad"][1],
                          anchor.ctypes.data, keep.ctypes.data, cls.ctypes.data, conf.ctypes.data,
                          box.ctypes.data, C.byref(ncand))
    return dict(n=n, anchor=anchor[:n].copy(), keep=keep[:n].copy(), cls=cls[:n].copy(),
                conf=conf[:n].copy(), boxes=box[:n].copy(), n_cand=ncand.value)


class Tracker:
    """Oracle twin of tracker.py IouTracker (streams addressed by index)."""

    def __init__(self, n_streams, max_age=30, min_iou=0.7, min_hits=3, cap=4096):
        self._h = lib().orc_tracker_new(n_streams, max_age, float(min_iou), min_hits)
        self.cap = cap
        L = lib()
        L.orc_tracker_update.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7
        self.last_new = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_tracker_free(self._h)
            self._h = None

    @property
    def next_id(self):
        return lib().orc_tracker_next_id(self._h)

    def update(self, stream: int, boxes, conf, cls):
        boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
        conf = np.ascontiguousarray(conf, np.float64)
        cls = np.ascontiguousarray(cls, np.int64)
        D = len(conf)
        cap = self.cap
        oid = np.empty(cap, np.int64); ocl = np.empty(cap, np.int32); oag = np.empty(cap, np.int32)
        ohi = np.empty(cap, np.int32); oco = np.empty(cap, np.float64); obx = np.empty((cap, 4), np.float64)
        nn = C.c_int()
        n = lib().orc_tracker_update(self._h, stream, D, boxes.ctypes.data, conf.ctypes.data, cls.ctypes.data,
                                     cap, oid.ctypes.data, ocl.ctypes.data, oag.ctypes.data, ohi.ctypes.data,
                                     oco.ctypes.data, obx.ctypes.data, C.byref(nn))
        assert n <= cap
        self.last_new = nn.value
        return dict(n=n, id=oid[:n].copy(), cls=ocl[:n].copy(), age=oag[:n].copy(), hits=ohi[:n].copy(),
                    conf=oco[:n].copy(), boxes=obx[:n].copy())


def table_of(res):
    """Same serialisation as oracle/gen_golden.py:_table."""
    return [[int(res["id"][i]), int(res["cls"][i]), int(res["age"][i]), int(res["hits"][i]),
             float(res["conf"][i]), [float(v) for v in res["boxes"][i]]] for i in range(res["n"])]


def nv12_to_bgr(y, uv, w, h):
    y = np.ascontiguousarray(y); uv = np.ascontiguousarray(uv)
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_nv12_to_bgr(C.c_void_p(y.ctypes.data), C.c_void_p(uv.ctypes.data), C.c_int(y.shape[1]),
                          C.c_int(w), C.c_int(h), C.c_void_p(out.ctypes.data))
    return out


def resize_linear(bgr, dw, dh):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((dh, dw, 3), np.uint8)
    lib().orc_resize_linear_u8c3(C.c_void_p(bgr.ctypes.data), w, h, C.c_void_p(out.ctypes.data), dw, dh)
    return out


def preprocess_bgr(bgr, tw=640, th=640, half=True):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((3, th, tw), np.float16 if half else np.float32)
    s = C.c_double(); l = C.c_int(); t = C.c_int()
    lib().orc_preprocess_bgr(C.c_void_p(bgr.ctypes.data), w, h, tw, th, int(half), C.c_void_p(out.ctypes.data),
                             C.byref(s), C.byref(l), C.byref(t))
    return out, dict(scale=s.value, pad=(l.value, t.value), orig_shape=(h, w))


def preprocess_nv12(y, uv, w, h, tw=640, th=640, half=True):
    y = np.ascontiguousarray(y); uv = np.ascontiguousarray(uv)
    out = np.empty((3, th, tw), np.float16 if half else np.float32)
    s = C.c_double(); l = C.c_int(); t = C.c_int()
    lib().orc_preprocess_nv12(C.c_void_p(y.ctypes.data), C.c_void_p(uv.ctypes.data), C.c_int(y.shape[1]), w, h,
                              tw, th, int(half), C.c_void_p(out.ctypes.data), C.byref(s), C.byref(l), C.byref(t))
    return out, dict(scale=s.value, pad=(l.value, t.value), orig_shape=(h, w))


def preprocess_clip_frame(bgr=None, nv12=None, wh=None, tw=224, th=224, half=False):
    out = np.empty((3, th, tw), np.float16 if half else np.float32)
    if bgr is not None:
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w = bgr.shape[:2]
        lib().orc_preprocess_clip_frame_bgr(C.c_void_p(bgr.ctypes.data), w, h, tw, th, int(half),
                                            C.c_void_p(out.ctypes.data))
    else:
        y, uv = nv12
        y = np.ascontiguousarray(y); uv = np.ascontiguousarray(uv)
        w, h = wh
        lib().orc_preprocess_clip_frame_nv12(C.c_void_p(y.ctypes.data), C.c_void_p(uv.ctypes.data),
                                             C.c_int(y.shape[1]), w, h, tw, th, int(half),
                                             C.c_void_p(out.ctypes.data))
    return out


_NORM_DT = {0: np.float16, 1: np.float32, 2: np.float64}


def preprocess_norm_frames(frames, tw, th, norm, out_dtype, layout=0, nv12_wh=None):
    """SURVEY 8f-4 pre-process of a list of frames (BGR uint8 arrays, or (y, uv) NV12 pairs with ``nv12_wh``):
    returns ``[n,3,th,tw]`` (layout 0) or ``[3,n,th,tw]`` (layout 1) in float16/32/64 (``out_dtype`` 0/1/2)."""
    n = len(frames)
    shape = (n, 3, th, tw) if layout == 0 else (3, n, th, tw)
    out = np.empty(shape, _NORM_DT[out_dtype])
    plane = th * tw
    fstride, cstride = (3 * plane, plane) if layout == 0 else (plane, n * plane)
    base = out.ctypes.data
    for i, f in enumerate(frames):
        dst = C.c_void_p(base + i * fstride * out.itemsize)
        if nv12_wh is None:
            bgr = np.ascontiguousarray(f, np.uint8)
            h, w = bgr.shape[:2]
            lib().orc_preprocess_norm_frame_bgr(C.c_void_p(bgr.ctypes.data), w, h, tw, th, norm, out_dtype, dst,
                                                C.c_long(cstride))
        else:
            y, uv = np.ascontiguousarray(f[0]), np.ascontiguousarray(f[1])
            w, h = nv12_wh
            lib().orc_preprocess_norm_frame_nv12(C.c_void_p(y.ctypes.data), C.c_void_p(uv.ctypes.data),
                                                 C.c_int(y.shape[1]), w, h, tw, th, norm, out_dtype, dst, C.c_long(cstride))
    return out


def resize_bgr(bgr, tw, th):
    """cv2.resize(bgr, (tw, th)) restated (INTER_LINEAR 11-bit fixed point): the uint8 image the normalisers see."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((th, tw, 3), np.uint8)
    lib().orc_resize_linear_u8c3(C.c_void_p(bgr.ctypes.data), w, h, C.c_void_p(out.ctypes.data), tw, th)
    return out


def clip_schedule(L, stride, overlap, n_frames):
    cap = n_frames
    fired = np.empty(cap, np.int32); ids = np.empty((cap, L), np.int32)
    n = lib().orc_clip_schedule(L, stride, C.c_double(overlap), n_frames, cap, C.c_void_p(fired.ctypes.data),
                                C.c_void_p(ids.ctypes.data))
    return fired[:n].tolist(), ids[:n].tolist()


def motion_step_nv12(y, uv, w, h, prev_blur=None):
    """One MotionFilter.should_process step on an NV12 frame: returns (changed-pixel count or -1, new blur)."""
    y = np.ascontiguousarray(y); uv = np.ascontiguousarray(uv)
    out = np.empty((h, w), np.uint8)
    L = lib()
    L.orc_motion_step_nv12.restype = C.c_long
    pp = C.c_void_p(prev_blur.ctypes.data) if prev_blur is not None else None
    n = L.orc_motion_step_nv12(C.c_void_p(y.ctypes.data), C.c_void_p(uv.ctypes.data), C.c_int(y.shape[1]), w, h, pp,
                               C.c_void_p(out.ctypes.data))
    return int(n), out

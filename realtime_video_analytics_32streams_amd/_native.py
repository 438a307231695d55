"""ctypes binding of librva.so (the C ABI declared in include/rva.h).

The library is built in-tree by :func:`build` (``hipcc --offload-arch=gfx950``) and travels to
the GPU box with the repository snapshot.  There is deliberately no fallback: if the shared
object is missing or no HIP device is usable, every product entry point raises ``RuntimeError``
(same contract as the reference's backends when their runtime is missing, detector.py:496-502).
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path
from typing import Optional

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB_PATH = Path(os.environ["RVA_LIB_PATH"]) if os.environ.get("RVA_LIB_PATH") else PKG / "librva.so"   # override: diagnostic builds (tools/)
SOURCES = ["rva_ctx.hip", "rva_preprocess.hip", "rva_postprocess.hip", "rva_tracker.hip", "rva_conv.hip", "rva_plan.hip", "rva_gates.hip",
           "rva_decode.hip", "rva_preview.hip", "rva_jpeg.hip"]
# -ffp-contract=off: parity kernels must not fuse a*b+c (SURVEY.md hard part 4)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function"]

RVA_OK, RVA_ERR_ARG, RVA_ERR_HIP, RVA_ERR_CAPACITY, RVA_ERR_UNAVAILABLE = range(5)
RVA_F16, RVA_F32, RVA_F64 = 0, 1, 2
NORM_IMAGENET_F32, NORM_VIDEO_F32, NORM_IMAGENET_F64 = 0, 1, 2      # enum rva_frame_norm
LAYOUT_NCHW, LAYOUT_CNHW = 0, 1                                    # enum rva_frame_layout
RVA_MAX_BATCH = 64
RVA_CODEC_H264, RVA_CODEC_HEVC = 0, 1


class Letterbox(C.Structure):
    _fields_ = [("src_w", C.c_int32), ("src_h", C.c_int32), ("dst_w", C.c_int32), ("dst_h", C.c_int32),
                ("new_w", C.c_int32), ("new_h", C.c_int32), ("pad_left", C.c_int32), ("pad_top", C.c_int32),
                ("scale", C.c_double)]

    def as_meta(self) -> dict:
        """The ``meta`` dict of detector.py:259-263."""
        return {"orig_shape": (self.src_h, self.src_w), "scale": self.scale, "pad": (self.pad_left, self.pad_top)}


class YoloV8Desc(C.Structure):
    """``rva_yolov8_desc`` (include/rva.h)."""
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("widths", C.c_int32 * 5),
                ("depth_backbone", C.c_int32 * 4), ("depth_head", C.c_int32), ("nc", C.c_int32), ("reg_max", C.c_int32),
                ("n_convs", C.c_int32), ("flags", C.c_int32)]


class ConvWeights(C.Structure):
    """``rva_conv_weights`` (include/rva.h): one convolution in the checkpoint's own fp32 layout."""
    _fields_ = [("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)), ("cout", C.c_int32), ("cin", C.c_int32),
                ("k", C.c_int32), ("stride", C.c_int32)]


RVA_PLAN_NO_STEM2 = 1
RVA_PLAN_NO_CIN_PAD = 2
RVA_PLAN_NO_PAIR32 = 4


def _stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "rva_internal.h", ROOT / "include" / "rva.h"]
    return any(d.exists() and d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile librva.so for gfx950 (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        if LIB_PATH.exists():
            return LIB_PATH
        raise RuntimeError("hipcc not found and librva.so is not built")
    cmd = [hipcc, *HIPCC_FLAGS, f"-I{ROOT / 'include'}", "-o", str(LIB_PATH), *[str(CSRC / s) for s in SOURCES], "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=str(CSRC))
    return LIB_PATH


_LIB: Optional[C.CDLL] = None
_P = C.c_void_p


def lib() -> C.CDLL:
    """Load librva.so (never builds implicitly on a box without sources changes; fails loudly)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        if os.environ.get("RVA_NO_AUTOBUILD"):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        build()
    try:
        L = C.CDLL(str(LIB_PATH))
    except OSError as exc:  # noqa: BLE001
        raise RuntimeError(f"cannot load {LIB_PATH}: {exc}. The HIP hot path has no CPU fallback.") from exc
    i32p, i64p, f32p, f64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double))
    pp = C.POINTER(_P)
    sig = {
        "rva_abi_version": (C.c_int, []),
        "rva_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
        "rva_destroy": (None, [_P]),
        "rva_last_error": (C.c_char_p, [_P]),
        "rva_reserve": (C.c_int, [_P, C.c_int, C.c_int]),
        "rva_letterbox_meta": (C.c_int, [C.c_int] * 4 + [C.POINTER(Letterbox)]),
        "rva_preprocess_nv12_batch": (C.c_int, [_P, pp, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                C.c_int, C.POINTER(Letterbox), _P]),
        "rva_preprocess_nv12_content_batch": (C.c_int, [_P, pp, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                        C.c_int, C.POINTER(Letterbox), _P]),
        "rva_preprocess_bgr_batch": (C.c_int, [_P, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(Letterbox), _P]),
        "rva_preprocess_clip_nv12_batch": (C.c_int, [_P, pp, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                     C.c_int, _P]),
        "rva_preprocess_clip_bgr_batch": (C.c_int, [_P, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                    C.c_int, _P]),
        "rva_profile_next_preprocess": (C.c_int, [_P, _P, _P]),
        "rva_preprocess_frames_nv12_batch": (C.c_int, [_P, pp, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, _P]),
        "rva_preprocess_frames_bgr_batch": (C.c_int, [_P, pp, i32p, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                      C.c_int, C.c_int, C.c_int, _P]),
        "rva_postprocess_batch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i32p,
                                            C.c_int, C.POINTER(Letterbox), C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P,
                                            _P]),
        "rva_post_status": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
        "rva_post_filter_stats": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
        "rva_tracker_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(_P)]),
        "rva_tracker_destroy": (None, [_P]),
        "rva_tracker_update_f32": (C.c_int, [_P, i32p, _P, _P, _P, _P, C.c_int, C.c_double, _P]),
        "rva_tracker_update_f64": (C.c_int, [_P, i32p, i32p, _P, _P, _P, _P]),
        "rva_tracker_update_gated_f32": (C.c_int, [_P, i32p, _P, _P, _P, _P, C.c_int, C.c_double, _P, i32p, _P]),
        "rva_tracker_set_gates": (C.c_int, [_P, i32p, i32p, i32p, i32p, C.c_int]),
        "rva_tracker_snapshot_status": (C.c_int, [_P, C.c_int, _P, _P, C.POINTER(C.c_int32)]),
        "rva_tracker_new_counts": (_P, [_P]),
        "rva_tracker_assign_ids": (C.c_int, [_P, _P, C.c_int, i32p, _P]),
        "rva_tracker_read": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, i32p, _P]),
        "rva_tracker_read_all": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
        "rva_tracker_snapshot_async": (C.c_int, [_P, C.c_int, _P]),
        "rva_tracker_snapshot_fetch": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
        "rva_tracker_state": (C.c_int, [_P, i64p, C.POINTER(C.c_int), _P]),
        "rva_tracker_set_next_id": (C.c_int, [_P, C.c_int64, _P]),
        "rva_preview_nv12": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int,
                                       C.c_int, _P]),
        "rva_decode_available": (C.c_int, [C.c_char_p, C.c_int]),
        "rva_decoder_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
        "rva_decoder_destroy": (None, [_P]),
        "rva_decoder_feed": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int64, C.c_int]),
        "rva_decoder_next_frame": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), i32p, i32p, i32p, i64p, i32p]),
        "rva_decoder_release": (C.c_int, [_P, C.c_int]),
        "rva_motion_nv12_batch": (C.c_int, [_P, pp, pp, i32p, pp, pp, C.c_int, C.c_int, C.c_int, _P, _P]),
        "rva_motion_nv12_masked_batch": (C.c_int, [_P, pp, pp, i32p, pp, pp, pp, C.c_int, C.c_int, C.c_int, _P, _P]),
        "rva_motion_bgr_batch": (C.c_int, [_P, pp, i32p, pp, pp, C.c_int, C.c_int, C.c_int, _P, _P]),
        "rva_preprocess_nv12_masked_batch": (C.c_int, [_P, pp, pp, i32p, pp, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                                       C.c_int, C.POINTER(Letterbox), _P]),
        "rva_resize_nv12_to_bgr_batch": (C.c_int, [_P, pp, pp, i32p, pp, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
        "rva_tracker_set_box_scale": (C.c_int, [_P, f64p]),
        "rva_conv_cout_pad": (C.c_int, [C.c_int]),
        "rva_conv_num_variants": (C.c_int, []),
        "rva_conv2d_nhwc_f16": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int] + [C.c_int] * 8 + [_P]),
        "rva_conv2d_nhwc_f16_v": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int] + [C.c_int] * 9 + [_P]),
        "rva_stem_conv_f16": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_stem2_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_c2f_pair32_f16": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_conv1x1_head_f16": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                           C.c_int, C.c_int, C.c_float, C.c_int, _P]),
        "rva_conv1x1_upcat_f16": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_yolo_head3_f16": (C.c_int, [_P, pp, i32p, pp, i32p, _P, C.c_int, i32p, i32p, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
        "rva_sppf_pool3_nhwc_f16": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_maxpool5_nhwc_f16": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_upsample2x_nhwc_f16": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
        "rva_yolo_head_f16": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, _P]),
        "rva_yolov8_plan_create": (C.c_int, [_P, C.POINTER(YoloV8Desc), C.POINTER(ConvWeights), C.POINTER(_P)]),
        "rva_yolov8_plan_destroy": (None, [_P]),
        "rva_yolov8_plan_info": (C.c_int, [_P, i32p, i32p, i32p, i32p, i32p]),
        "rva_yolov8_plan_run": (C.c_int, [_P, _P, _P, _P]),
        "rva_yolov8_plan_run_lanes": (C.c_int, [_P, _P, _P, _P, _P, _P]),
        "rva_yolov8_plan_run_range": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
        "rva_yolov8_plan_tunable_desc": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int]),
        "rva_yolov8_plan_launch_tunable": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
        "rva_yolov8_plan_set_variant": (C.c_int, [_P, C.c_int, C.c_int]),
        "rva_yolov8_plan_get_variant": (C.c_int, [_P, C.c_int]),
        "rva_jpeg_max_bytes": (C.c_int, [C.c_int, C.c_int]),
        "rva_jpeg_encode_bgr": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
        "rva_jpeg_status": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if L.rva_abi_version() != 1:
        raise RuntimeError("librva.so ABI version mismatch; rebuild it")
    _LIB = L
    return L


EXPORTS = [
    "rva_abi_version", "rva_create", "rva_destroy", "rva_last_error", "rva_reserve", "rva_letterbox_meta",
    "rva_preprocess_nv12_batch", "rva_preprocess_nv12_content_batch", "rva_preprocess_bgr_batch", "rva_preprocess_clip_nv12_batch",
    "rva_preprocess_clip_bgr_batch", "rva_profile_next_preprocess", "rva_preprocess_frames_nv12_batch", "rva_preprocess_frames_bgr_batch", "rva_postprocess_batch", "rva_post_status", "rva_post_filter_stats", "rva_tracker_create",
    "rva_tracker_destroy", "rva_tracker_update_f32", "rva_tracker_update_f64", "rva_tracker_update_gated_f32",
    "rva_tracker_set_gates", "rva_tracker_snapshot_status", "rva_tracker_new_counts",
    "rva_tracker_assign_ids", "rva_tracker_read", "rva_tracker_read_all", "rva_tracker_snapshot_async",
    "rva_tracker_snapshot_fetch", "rva_tracker_state",
    "rva_tracker_set_next_id", "rva_preview_nv12", "rva_decode_available", "rva_decoder_create", "rva_decoder_destroy", "rva_decoder_feed",
    "rva_decoder_next_frame", "rva_decoder_release", "rva_motion_nv12_batch", "rva_motion_nv12_masked_batch", "rva_motion_bgr_batch",
    "rva_preprocess_nv12_masked_batch", "rva_resize_nv12_to_bgr_batch", "rva_tracker_set_box_scale", "rva_conv_cout_pad", "rva_conv_num_variants", "rva_conv2d_nhwc_f16", "rva_conv2d_nhwc_f16_v", "rva_stem_conv_f16", "rva_stem2_f16", "rva_c2f_pair32_f16",
    "rva_conv1x1_head_f16", "rva_conv1x1_upcat_f16", "rva_sppf_pool3_nhwc_f16", "rva_maxpool5_nhwc_f16", "rva_upsample2x_nhwc_f16", "rva_yolo_head_f16", "rva_yolo_head3_f16",
    "rva_yolov8_plan_create", "rva_yolov8_plan_destroy", "rva_yolov8_plan_info", "rva_yolov8_plan_run", "rva_yolov8_plan_run_lanes",
    "rva_yolov8_plan_run_range", "rva_yolov8_plan_tunable_desc", "rva_yolov8_plan_launch_tunable", "rva_yolov8_plan_set_variant",
    "rva_yolov8_plan_get_variant", "rva_jpeg_max_bytes", "rva_jpeg_encode_bgr", "rva_jpeg_status",
]


class Context:
    """Owns one ``rva_ctx`` (one per GPU / process)."""

    def __init__(self, device: int = 0):
        L = lib()
        h = _P()
        rc = L.rva_create(device, C.byref(h))
        if rc != RVA_OK or not h:
            raise RuntimeError(
                f"rva_create(device={device}) failed with status {rc}: no usable HIP device. "
                "The MI355X hot path has no CPU fallback.")
        self.handle = h
        self.device = device

    def check(self, rc: int, what: str = "") -> None:
        if rc != RVA_OK:
            msg = lib().rva_last_error(self.handle)
            raise RuntimeError(f"{what or 'librva'} failed (status {rc}): {msg.decode() if msg else ''}")

    def close(self) -> None:
        if getattr(self, "handle", None):
            lib().rva_destroy(self.handle)
            self.handle = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def letterbox(src_w: int, src_h: int, dst_w: int, dst_h: int) -> Letterbox:
    m = Letterbox()
    rc = lib().rva_letterbox_meta(src_w, src_h, dst_w, dst_h, C.byref(m))
    if rc != RVA_OK:
        raise ValueError(f"bad letterbox geometry {(src_w, src_h)} -> {(dst_w, dst_h)}")
    return m


def ptr_array(ptrs):
    arr = (_P * len(ptrs))(*[_P(int(p)) for p in ptrs])
    return C.cast(arr, C.POINTER(_P)), arr


def i32_array(vals):
    arr = (C.c_int32 * len(vals))(*[int(v) for v in vals])
    return C.cast(arr, C.POINTER(C.c_int32)), arr

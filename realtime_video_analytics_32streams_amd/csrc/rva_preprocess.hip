// K1: fused pre-process for a whole tick of frames in one launch.
//
// Replaces _TensorRTBaseDetector._preprocess (detector.py:198-264) and the per-frame body of
// CNNLSTMDetector._preprocess_sequence (temporal_detector.py:340-359):
//   [NV12 -> BGR, BT.601 limited, nearest chroma: what FFmpeg/OpenCV would have handed over]
//   -> cv2.resize INTER_LINEAR (OpenCV's 11-bit fixed-point taps, bit-exact integer arithmetic)
//   -> letterbox pad 114 -> BGR2RGB -> astype(dtype) * (1/255) (binary16 multiply when half)
//   -> planar CHW written straight into the batch tensor.
// The colour conversion is applied to each TAP before interpolation (convert-then-resize), which is
// the reference's order; only the tapped pixels are converted.
//
// Two kernels:
//   k1_ratio<R>   source = R x letterboxed content (R integer: 1080p->640 is R=3, 4K is R=6, 720p
//                 R=2): the taps are src[R*y + (R-1)/2] (odd R, fraction 0) or the 2x2 block at
//                 R*y + R/2 - 1 (even R, fraction 1/2 => (a+b+c+d+2)>>2 in OpenCV's fixed point).
//                 Each lane owns 8 output pixels = 8R contiguous source bytes per plane row, loaded as
//                 R aligned 8-byte words (lanes contiguous => whole rows stream through once), and
//                 stores one 16-byte (fp16) vector per colour plane.
//   k1_generic    any geometry / BGR frames / clip (stretch + mean/std) epilogue: per-pixel taps from
//                 the tap tables (host-computed exactly like OpenCV, cached on the device).
#include <hip/hip_fp16.h>

#include <cstdlib>

#include <hip/hip_ext.h>

#include "rva_internal.h"

namespace {

struct K1Args {
    const uint8_t *p0[RVA_MAX_BATCH];  // Y plane (NV12) or BGR frame
    const uint8_t *p1[RVA_MAX_BATCH];  // interleaved UV plane (NV12)
    int32_t pitch[RVA_MAX_BATCH];      // bytes per row
    const uint8_t *mask[RVA_MAX_BATCH]; // optional ROI mask uint8 [src_h][src_w] (0 = outside: pixel becomes BGR 0,0,0)
    int src_w, src_h, dst_w, dst_h, new_w, new_h, left, top;
    const int32_t *xofs; const int16_t *xw0, *xw1;
    const int32_t *yofs; const int16_t *yw0, *yw1;
    void *out;
    int norm;                          // MODE 1: rva_frame_norm
    long fstride, cstride;             // MODE 1: output element strides between frames / channels
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// BT.601 limited-range integer matrix (D1 stand-in, see oracle/rva_oracle.c orc_yuv_to_bgr)
__device__ __forceinline__ void yuv2bgr(int Y, int U, int V, int &b, int &g, int &r)
{
    const int c = 298 * (Y - 16), d = U - 128, e = V - 128;
    b = clip8((c + 516 * d + 128) >> 8);
    g = clip8((c - 100 * d - 208 * e + 128) >> 8);
    r = clip8((c + 409 * e + 128) >> 8);
}

// ---- epilogues -------------------------------------------------------------------------------
constexpr unsigned short kInv255Half = 0x1C04;  // float16(1/255): NEP-50 casts the python float directly

template <typename OutT> struct Vec8;
template <> struct Vec8<__half> { uint4 v; };
template <> struct Vec8<float> { float4 lo, hi; };

// detector.py:248-251: uint8.astype(dtype) * (1.0/255.0)
__device__ __forceinline__ __half norm_yolo(int v, __half) { return __hmul(__int2half_rn(v), __ushort_as_half(kInv255Half)); }
__device__ __forceinline__ float norm_yolo(int v, float) { return (float)v * (float)(1.0 / 255.0); }

// temporal_detector.py:350-354 / :570-573 / :741-743, detector.py:988-993: float32 /255.0, then (x-mean)/std in
// float32 (norm 0, 1) or -- ConvGRU's float64 constant arrays -- in float64 (norm 2); the dtype cast comes last
__device__ __forceinline__ __half d2h_rn(double d)
{   // one rounding double -> half: truncate to float keeping a sticky bit (round to odd), then round to nearest
    float f = __double2float_rz(d);
    if ((double)f != d) f = __uint_as_float(__float_as_uint(f) | 1u);
    return __float2half_rn(f);
}
__device__ __forceinline__ void cast_out(double d, __half &o) { o = d2h_rn(d); }
__device__ __forceinline__ void cast_out(double d, float &o) { o = (float)d; }
__device__ __forceinline__ void cast_out(double d, double &o) { o = d; }
__device__ __forceinline__ void cast_out(float f, __half &o) { o = __float2half_rn(f); }
__device__ __forceinline__ void cast_out(float f, float &o) { o = f; }
__device__ __forceinline__ void cast_out(float f, double &o) { o = (double)f; }
template <typename OutT>
__device__ __forceinline__ OutT norm_frame(int v, int c, int norm)
{
    const float x = __fdiv_rn((float)v, 255.0f);
    OutT o;
    if (norm == 2) {
        const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
        cast_out(((double)x - mean[c]) / stdv[c], o);
    } else {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        const float m = norm == 1 ? 0.45f : mean[c], sd = norm == 1 ? 0.225f : stdv[c];
        cast_out(__fdiv_rn(x - m, sd), o);
    }
    return o;
}

template <typename OutT>
__device__ __forceinline__ void store8(OutT *dst, const OutT *v, bool vec_ok, int nvalid)
{
    if (vec_ok && nvalid == 8) {
        if constexpr (sizeof(OutT) == 2) {
            *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(v);
        } else if constexpr (sizeof(OutT) == 4) {
            reinterpret_cast<float4 *>(dst)[0] = reinterpret_cast<const float4 *>(v)[0];
            reinterpret_cast<float4 *>(dst)[1] = reinterpret_cast<const float4 *>(v)[1];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) reinterpret_cast<double2 *>(dst)[i] = reinterpret_cast<const double2 *>(v)[i];
        }
    } else {
        for (int i = 0; i < nvalid; ++i) dst[i] = v[i];
    }
}

// ---- integer-ratio fast path -----------------------------------------------------------------
template <int R, typename OutT, int PX, bool MASK>
__global__ void __launch_bounds__(256) k1_ratio(K1Args a)
{
    const int img = blockIdx.y;
    const int groups = a.dst_w / PX;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= groups * a.dst_h) return;
    const int oy = item / groups, xg = item - oy * groups;
    const int ox = xg * PX;
    const size_t plane = (size_t)a.dst_w * a.dst_h;
    OutT *out = (OutT *)a.out + (size_t)img * 3 * plane + (size_t)oy * a.dst_w + ox;
    alignas(16) OutT vr[PX], vg[PX], vb[PX];
    const int cy = oy - a.top, cx = ox - a.left;
    if (cy < 0 || cy >= a.new_h || cx < 0 || cx >= a.new_w) {  // letterbox border (detector.py:233-241)
        const OutT p = norm_yolo(114, OutT());
#pragma unroll
        for (int i = 0; i < PX; ++i) vr[i] = vg[i] = vb[i] = p;
    } else {
        const uint8_t *yp = a.p0[img];
        const uint8_t *uvp = a.p1[img];
        const int pitch = a.pitch[img];
        constexpr int NW = R * PX / 8; // 8-byte words per PX output pixels
        constexpr bool ODD = (R & 1) != 0;
        constexpr int ROWS = ODD ? 1 : 2;
        const int sy0 = ODD ? R * cy + (R - 1) / 2 : R * cy + R / 2 - 1;
        const size_t xoff = (size_t)R * cx;  // multiple of PX
        uint2 yw[ROWS][NW], uw[ROWS][NW];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint2 *ys = reinterpret_cast<const uint2 *>(yp + (size_t)(sy0 + r) * pitch + xoff);
            const uint2 *us = reinterpret_cast<const uint2 *>(uvp + (size_t)((sy0 + r) >> 1) * pitch + xoff);
            if constexpr (PX == 16) {   // 16-byte loads (two words each)
#pragma unroll
                for (int k = 0; k < NW; k += 2) {
                    const uint4 y4 = *reinterpret_cast<const uint4 *>(ys + k), u4v = *reinterpret_cast<const uint4 *>(us + k);
                    yw[r][k] = make_uint2(y4.x, y4.y); yw[r][k + 1] = make_uint2(y4.z, y4.w);
                    uw[r][k] = make_uint2(u4v.x, u4v.y); uw[r][k + 1] = make_uint2(u4v.z, u4v.w);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NW; ++k) { yw[r][k] = ys[k]; uw[r][k] = us[k]; }
            }
        }
        auto byte_at = [](const uint2 *w, int o) -> int {
            const uint32_t d = (o & 4) ? w[o >> 3].y : w[o >> 3].x;
            return (int)((d >> ((o & 3) * 8)) & 0xffu);
        };
        uint2 mw[ROWS][NW];
        const uint8_t *mp = MASK ? a.mask[img] : nullptr;       // apply_roi: frame & mask (utils/frame_filter.py:43-50)
        if (MASK && mp) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const uint2 *ms = reinterpret_cast<const uint2 *>(mp + (size_t)(sy0 + r) * a.src_w + xoff);
#pragma unroll
                for (int k = 0; k < NW; ++k) mw[r][k] = ms[k];
            }
        }
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int b, g, r;
            if constexpr (ODD) {
                const int o = R * i + (R - 1) / 2;
                const int uo = (o >> 1) << 1;
                yuv2bgr(byte_at(yw[0], o), byte_at(uw[0], uo), byte_at(uw[0], uo + 1), b, g, r);
                if (MASK && mp && byte_at(mw[0], o) == 0) b = g = r = 0;
            } else {
                int sb = 2, sg = 2, sr = 2;  // (a + b + c + d + 2) >> 2
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const int o = R * i + R / 2 - 1 + cc;
                        const int uo = (o >> 1) << 1;
                        int tb, tg, tr;
                        yuv2bgr(byte_at(yw[rr], o), byte_at(uw[rr], uo), byte_at(uw[rr], uo + 1), tb, tg, tr);
                        if (MASK && mp && byte_at(mw[rr], o) == 0) tb = tg = tr = 0;
                        sb += tb; sg += tg; sr += tr;
                    }
                b = sb >> 2; g = sg >> 2; r = sr >> 2;
            }
            vr[i] = norm_yolo(r, OutT()); vg[i] = norm_yolo(g, OutT()); vb[i] = norm_yolo(b, OutT());
        }
    }
#pragma unroll
    for (int h = 0; h < PX; h += 8) {
        store8<OutT>(out + h, vr + h, true, 8);              // R plane first (BGR2RGB, detector.py:245)
        store8<OutT>(out + plane + h, vg + h, true, 8);
        store8<OutT>(out + 2 * plane + h, vb + h, true, 8);
    }
}

// ---- integer-ratio fast path, steady state ---------------------------------------------------------
// The letterbox border of a detector input tensor is a constant (114/255): once a full launch has written it, the
// following ticks of the same geometry only owe the CONTENT rows (1080p -> 640: 360 of 640 rows; the border is 44 % of
// the tensor).  This kernel writes the content region only -- no border branch, so the loads of RPT content rows per
// thread (half a frame apart) are issued back to back before any arithmetic: twice the bytes in flight per thread and
// half the workgroups of k1_ratio.  Same taps, same arithmetic, same values as k1_ratio.
// NT (experiment, RVA_K1_NT=1|2|3 read once per context; profiles/r04_experiments_not_kept.txt #6): non-temporal loads (bit 0) and /
// or stores (bit 1), so that K1's once-read surfaces and once-written tensor do not evict the convolutions' lines from L2.
typedef unsigned int k1_u4 __attribute__((ext_vector_type(4)));
template <int R, typename OutT, int RPT, int NT = 0>
__global__ void __launch_bounds__(256) k1_ratio_content(K1Args a)
{
    constexpr int PX = 8;
    constexpr int NW = R * PX / 8;
    constexpr bool ODD = (R & 1) != 0;
    constexpr int ROWS = ODD ? 1 : 2;
    const int img = blockIdx.y;
    const int groups = a.new_w / PX, rows_per = a.new_h / RPT;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= groups * rows_per) return;
    const int r0 = item / groups, xg = item - r0 * groups;
    const int cx = xg * PX;
    const uint8_t *yp = a.p0[img];
    const uint8_t *uvp = a.p1[img];
    const int pitch = a.pitch[img];
    const size_t xoff = (size_t)R * cx;
    uint2 yw[RPT][ROWS][NW], uw[RPT][ROWS][NW];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int cy = r0 + q * rows_per;
        const int sy0 = ODD ? R * cy + (R - 1) / 2 : R * cy + R / 2 - 1;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint2 *ys = reinterpret_cast<const uint2 *>(yp + (size_t)(sy0 + r) * pitch + xoff);
            const uint2 *us = reinterpret_cast<const uint2 *>(uvp + (size_t)((sy0 + r) >> 1) * pitch + xoff);
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                if constexpr (NT & 1) {
                    const unsigned long long ty = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(ys) + k);
                    const unsigned long long tu = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(us) + k);
                    yw[q][r][k] = make_uint2((uint32_t)ty, (uint32_t)(ty >> 32));
                    uw[q][r][k] = make_uint2((uint32_t)tu, (uint32_t)(tu >> 32));
                } else {
                    yw[q][r][k] = ys[k]; uw[q][r][k] = us[k];
                }
            }
        }
    }
    auto byte_at = [](const uint2 *w, int o) -> int {
        const uint32_t d = (o & 4) ? w[o >> 3].y : w[o >> 3].x;
        return (int)((d >> ((o & 3) * 8)) & 0xffu);
    };
    const size_t plane = (size_t)a.dst_w * a.dst_h;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        alignas(16) OutT vr[PX], vg[PX], vb[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int b, g, r;
            if constexpr (ODD) {
                const int o = R * i + (R - 1) / 2;
                const int uo = (o >> 1) << 1;
                yuv2bgr(byte_at(yw[q][0], o), byte_at(uw[q][0], uo), byte_at(uw[q][0], uo + 1), b, g, r);
            } else {
                int sb = 2, sg = 2, sr = 2;  // (a + b + c + d + 2) >> 2
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const int o = R * i + R / 2 - 1 + cc;
                        const int uo = (o >> 1) << 1;
                        int tb, tg, tr;
                        yuv2bgr(byte_at(yw[q][rr], o), byte_at(uw[q][rr], uo), byte_at(uw[q][rr], uo + 1), tb, tg, tr);
                        sb += tb; sg += tg; sr += tr;
                    }
                b = sb >> 2; g = sg >> 2; r = sr >> 2;
            }
            vr[i] = norm_yolo(r, OutT()); vg[i] = norm_yolo(g, OutT()); vb[i] = norm_yolo(b, OutT());
        }
        OutT *out = (OutT *)a.out + (size_t)img * 3 * plane + (size_t)(a.top + r0 + q * rows_per) * a.dst_w + a.left + cx;
        if constexpr ((NT & 2) && sizeof(OutT) == 2) {
            __builtin_nontemporal_store(*reinterpret_cast<const k1_u4 *>(vr), reinterpret_cast<k1_u4 *>(out));
            __builtin_nontemporal_store(*reinterpret_cast<const k1_u4 *>(vg), reinterpret_cast<k1_u4 *>(out + plane));
            __builtin_nontemporal_store(*reinterpret_cast<const k1_u4 *>(vb), reinterpret_cast<k1_u4 *>(out + 2 * plane));
        } else {
            store8<OutT>(out, vr, true, 8);
            store8<OutT>(out + plane, vg, true, 8);
            store8<OutT>(out + 2 * plane, vb, true, 8);
        }
    }
}

// ---- general path ----------------------------------------------------------------------------
// MODE: 0 = detector tensor (letterbox, x 1/255), 1 = clip tensor (stretch, mean/std), 2 = uint8 BGR HWC image
// (stretch; the `downsample` stage of utils/frame_filter.py:53-57)
// PX: output pixels per thread.  8 (one 16-byte store per plane) when neighbouring output pixels share source cache lines;
// 1 for the heavy decimations of the clip heads (4K -> 224: 17 source pixels between two taps, every tap its own sector),
// where eight times the threads means eight times the loads in flight -- the kernel is latency-bound there, not byte-bound.
template <bool NV12, int MODE, typename OutT, int PX = 8>
__global__ void __launch_bounds__(256) k1_generic(K1Args a)
{
    const int img = blockIdx.y;
    const int groups = (a.dst_w + PX - 1) / PX;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= groups * a.dst_h) return;
    const int oy = item / groups, xg = item - oy * groups;
    const int ox = xg * PX;
    const int nvalid = a.dst_w - ox < PX ? a.dst_w - ox : PX;
    const size_t plane = (size_t)a.dst_w * a.dst_h;
    const size_t fstride = MODE == 1 ? (size_t)a.fstride : 3 * plane, cstride = MODE == 1 ? (size_t)a.cstride : plane;
    OutT *out = (OutT *)a.out + (size_t)img * fstride + (size_t)oy * a.dst_w + ox;
    alignas(16) OutT vr[PX], vg[PX], vb[PX];
    const uint8_t *p0 = a.p0[img];
    const uint8_t *p1 = a.p1[img];
    const int pitch = a.pitch[img];
    const int cy = oy - a.top;
    const bool row_in = cy >= 0 && cy < a.new_h;
    int sy0 = 0, sy1 = 0, wb0 = 0, wb1 = 0;
    if (row_in) {
        const int s = a.yofs[cy];
        sy0 = min(max(s, 0), a.src_h - 1);       // resizeGeneric_Invoker row clip
        sy1 = min(max(s + 1, 0), a.src_h - 1);
        wb0 = a.yw0[cy]; wb1 = a.yw1[cy];
    }
    const uint8_t *mp = a.mask[img];
    auto fetch = [&](int sy, int sx, int &b, int &g, int &r) {
        if (mp && mp[(size_t)sy * a.src_w + sx] == 0) { b = g = r = 0; return; }   // apply_roi
        if constexpr (NV12) {
            const int Y = p0[(size_t)sy * pitch + sx];
            const uint8_t *u = p1 + (size_t)(sy >> 1) * pitch + ((sx >> 1) << 1);
            yuv2bgr(Y, u[0], u[1], b, g, r);
        } else {
            const uint8_t *p = p0 + (size_t)sy * pitch + (size_t)sx * 3;
            b = p[0]; g = p[1]; r = p[2];
        }
    };
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        int b = 114, g = 114, r = 114;
        const int cx = ox + i - a.left;
        if (row_in && cx >= 0 && cx < a.new_w && i < nvalid) {
            const int sx0 = a.xofs[cx];
            const int sx1 = min(sx0 + 1, a.src_w - 1);
            const int wa0 = a.xw0[cx], wa1 = a.xw1[cx];
            int b00, g00, r00, b01 = 0, g01 = 0, r01 = 0, b10 = 0, g10 = 0, r10 = 0, b11 = 0, g11 = 0, r11 = 0;
            fetch(sy0, sx0, b00, g00, r00);
            if (wa1) fetch(sy0, sx1, b01, g01, r01);
            if (wb1) {
                fetch(sy1, sx0, b10, g10, r10);
                if (wa1) fetch(sy1, sx1, b11, g11, r11);
            }
            // HResizeLinear: S0*a0 + S1*a1 ; VResizeLinear: ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2 >> 2
            auto mix = [&](int s00, int s01, int s10, int s11) {
                const int h0 = s00 * wa0 + s01 * wa1, h1 = s10 * wa0 + s11 * wa1;
                return clip8((((wb0 * (h0 >> 4)) >> 16) + ((wb1 * (h1 >> 4)) >> 16) + 2) >> 2);
            };
            b = mix(b00, b01, b10, b11); g = mix(g00, g01, g10, g11); r = mix(r00, r01, r10, r11);
        }
        if constexpr (MODE == 2) {
            if (i < nvalid) {
                uint8_t *o8 = (uint8_t *)a.out + ((size_t)img * a.dst_h * a.dst_w + (size_t)oy * a.dst_w + ox + i) * 3;
                o8[0] = (uint8_t)b; o8[1] = (uint8_t)g; o8[2] = (uint8_t)r;
            }
        } else if constexpr (MODE == 1) {
            vr[i] = norm_frame<OutT>(r, 0, a.norm); vg[i] = norm_frame<OutT>(g, 1, a.norm); vb[i] = norm_frame<OutT>(b, 2, a.norm);
        } else if constexpr (sizeof(OutT) <= 4) {
            vr[i] = norm_yolo(r, OutT()); vg[i] = norm_yolo(g, OutT()); vb[i] = norm_yolo(b, OutT());
        }
    }
    if constexpr (MODE == 2) return;
    if constexpr (PX == 8) {
        const bool vec_ok = (a.dst_w & 7) == 0;
        store8<OutT>(out, vr, vec_ok, nvalid);
        store8<OutT>(out + cstride, vg, vec_ok, nvalid);
        store8<OutT>(out + 2 * cstride, vb, vec_ok, nvalid);
    } else {
#pragma unroll
        for (int i = 0; i < PX; ++i)
            if (i < nvalid) { out[i] = vr[i]; out[cstride + i] = vg[i]; out[2 * cstride + i] = vb[i]; }
    }
}

// kernel launch with optional start / stop events written by the dispatch itself (hipExtLaunchKernelGGL): the pair
// brackets exactly this kernel, without the inter-launch gap two hipEventRecord calls around it would include
template <int R, typename OutT, int PX, bool MASK>
void launch_ratio_r(dim3 grid, hipStream_t s, const K1Args &a, hipEvent_t e0, hipEvent_t e1)
{
    if (e0 && e1) hipExtLaunchKernelGGL((k1_ratio<R, OutT, PX, MASK>), grid, dim3(256), 0, s, e0, e1, 0, a);
    else k1_ratio<R, OutT, PX, MASK><<<grid, 256, 0, s>>>(a);
}

template <typename OutT, int PX, bool MASK = false>
bool launch_ratio(int R, dim3 grid, hipStream_t s, const K1Args &a, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    switch (R) {
        case 1: launch_ratio_r<1, OutT, PX, MASK>(grid, s, a, e0, e1); return true;
        case 2: launch_ratio_r<2, OutT, PX, MASK>(grid, s, a, e0, e1); return true;
        case 3: launch_ratio_r<3, OutT, PX, MASK>(grid, s, a, e0, e1); return true;
        case 4: launch_ratio_r<4, OutT, PX, MASK>(grid, s, a, e0, e1); return true;
        case 6: launch_ratio_r<6, OutT, PX, MASK>(grid, s, a, e0, e1); return true;
        default: return false;
    }
}

template <int R, typename OutT>
void launch_content_r(int n, hipStream_t s, const K1Args &a, hipEvent_t e0, hipEvent_t e1, int nt = 0)
{
    const int groups = a.new_w / 8;
    if (a.new_h % 2 == 0) {
        dim3 grid(rva_ceil_div(groups * (a.new_h / 2), 256), n);
        if (nt && R == 3 && sizeof(OutT) == 2) {     // the experiment's forms exist for the headline geometry only (1080p -> 640, fp16)
            auto go = [&](auto kern) {
                if (e0 && e1) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, s, e0, e1, 0, a);
                else hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
            };
            if (nt == 1) go(k1_ratio_content<R, OutT, 2, 1>);
            else if (nt == 2) go(k1_ratio_content<R, OutT, 2, 2>);
            else go(k1_ratio_content<R, OutT, 2, 3>);
            return;
        }
        if (e0 && e1) hipExtLaunchKernelGGL((k1_ratio_content<R, OutT, 2>), grid, dim3(256), 0, s, e0, e1, 0, a);
        else k1_ratio_content<R, OutT, 2><<<grid, 256, 0, s>>>(a);
    } else {
        dim3 grid(rva_ceil_div(groups * a.new_h, 256), n);
        if (e0 && e1) hipExtLaunchKernelGGL((k1_ratio_content<R, OutT, 1>), grid, dim3(256), 0, s, e0, e1, 0, a);
        else k1_ratio_content<R, OutT, 1><<<grid, 256, 0, s>>>(a);
    }
}

template <typename OutT>
bool launch_content(int R, int n, hipStream_t s, const K1Args &a, hipEvent_t e0, hipEvent_t e1, int nt = 0)
{
    switch (R) {
        case 1: launch_content_r<1, OutT>(n, s, a, e0, e1); return true;
        case 2: launch_content_r<2, OutT>(n, s, a, e0, e1); return true;
        case 3: launch_content_r<3, OutT>(n, s, a, e0, e1, nt); return true;
        case 4: launch_content_r<4, OutT>(n, s, a, e0, e1); return true;
        case 6: launch_content_r<6, OutT>(n, s, a, e0, e1); return true;
        default: return false;
    }
}

int preprocess_common(rva_ctx *ctx, bool nv12, int mode, const void *const *p0, const void *const *p1,
                      const int32_t *pitches, int n, int src_w, int src_h, void *out, int out_dtype, int dst_w,
                      int dst_h, rva_letterbox *meta_out, hipStream_t stream, const void *const *masks = nullptr,
                      int norm = 0, int layout = 0, bool content_only = false)
{
    const bool clip = mode != 0;   // modes 1 and 2 stretch to the full target, no letterbox border
    if (!ctx) return RVA_ERR_ARG;
    if (!p0 || (nv12 && !p1) || !pitches || n <= 0 || n > RVA_MAX_BATCH || !out || src_w <= 0 || src_h <= 0 ||
        dst_w <= 0 || dst_h <= 0)
        return rva_fail(ctx, RVA_ERR_ARG, "preprocess: bad argument (1 <= n <= %d)", RVA_MAX_BATCH);
    if (out_dtype == RVA_F64 ? !(mode == 1 && norm == RVA_NORM_IMAGENET_F64) : (out_dtype != RVA_F16 && out_dtype != RVA_F32))
        return rva_fail(ctx, RVA_ERR_ARG, "out_dtype must be RVA_F16|RVA_F32 (RVA_F64 only with RVA_NORM_IMAGENET_F64)");
    if (norm < 0 || norm > 2 || layout < 0 || layout > 1) return rva_fail(ctx, RVA_ERR_ARG, "bad norm / layout");
    if (nv12 && ((src_w | src_h) & 1)) return rva_fail(ctx, RVA_ERR_ARG, "NV12 surfaces need even dimensions");
    rva_letterbox m;
    if (clip) {  // stretch to the full target (temporal_detector.py:344), no border
        m.src_w = src_w; m.src_h = src_h; m.dst_w = dst_w; m.dst_h = dst_h; m.new_w = dst_w; m.new_h = dst_h;
        m.pad_left = m.pad_top = 0; m.scale = 0.0;
    } else if (rva_letterbox_meta(src_w, src_h, dst_w, dst_h, &m) != RVA_OK) {
        return rva_fail(ctx, RVA_ERR_ARG, "bad geometry");
    }
    if (m.new_w <= 0 || m.new_h <= 0) return rva_fail(ctx, RVA_ERR_ARG, "frame too thin for this input size");
    if (meta_out) *meta_out = m;

    K1Args a{};
    bool aligned8 = true;
    for (int i = 0; i < n; ++i) {
        a.p0[i] = (const uint8_t *)p0[i];
        a.p1[i] = nv12 ? (const uint8_t *)p1[i] : nullptr;
        a.pitch[i] = pitches[i];
        a.mask[i] = masks ? (const uint8_t *)masks[i] : nullptr;
        if (!p0[i] || (nv12 && !p1[i]) || pitches[i] < (nv12 ? src_w : 3 * src_w))
            return rva_fail(ctx, RVA_ERR_ARG, "preprocess: surface %d has a null plane or a short pitch", i);
        aligned8 = aligned8 && ((uintptr_t)p0[i] % 8 == 0) && (!nv12 || (uintptr_t)p1[i] % 8 == 0) && pitches[i] % 8 == 0;
    }
    a.src_w = src_w; a.src_h = src_h; a.dst_w = dst_w; a.dst_h = dst_h;
    a.new_w = m.new_w; a.new_h = m.new_h; a.left = m.pad_left; a.top = m.pad_top;
    a.out = out;
    bool any_mask = false, mask_aligned = true;
    for (int i = 0; i < n; ++i) {
        any_mask = any_mask || a.mask[i];
        mask_aligned = mask_aligned && ((uintptr_t)a.mask[i] % 8 == 0);
    }
    mask_aligned = mask_aligned && src_w % 8 == 0;
    const size_t osz = mode == 2 ? 1 : (out_dtype == RVA_F16 ? 2 : out_dtype == RVA_F32 ? 4 : 8);
    a.norm = norm;
    a.fstride = layout == RVA_LAYOUT_CNHW ? (long)dst_w * dst_h : 3L * dst_w * dst_h;
    a.cstride = layout == RVA_LAYOUT_CNHW ? (long)n * dst_w * dst_h : (long)dst_w * dst_h;
    const bool out_aligned = ((uintptr_t)out % 16 == 0) && ((dst_w * osz) % 16 == 0);

    // integer-ratio fast path
    const int R = src_w / m.new_w;
    const bool ratio_ok = nv12 && !clip && aligned8 && out_aligned && (!any_mask || mask_aligned) && R * m.new_w == src_w && R * m.new_h == src_h &&
                          (dst_w % 8 == 0) && (m.new_w % 8 == 0) && (m.pad_left % 8 == 0);
    if (ratio_ok) {
        // 16 pixels per lane (16-byte loads) when the geometry is 16-aligned and every surface 16-byte aligned
        if (ctx->k1_px < 0) { const char *e = getenv("RVA_K1_PX"); ctx->k1_px = e ? atoi(e) : 0; }
        const int px_env = ctx->k1_px;
        bool aligned16 = true;
        for (int i = 0; i < n; ++i)
            aligned16 = aligned16 && ((uintptr_t)p0[i] % 16 == 0) && ((uintptr_t)p1[i] % 16 == 0) && pitches[i] % 16 == 0;
        const bool px16 = !any_mask && px_env == 16 && aligned16   /* measured slower than 8 px/lane (profiles/r01_k1_variants.txt): opt-in only */ && dst_w % 16 == 0 && m.new_w % 16 == 0 && m.pad_left % 16 == 0 && R <= 3 &&
                          out_dtype == RVA_F16;
        const int PXv = px16 ? 16 : 8;
        dim3 grid(rva_ceil_div((dst_w / PXv) * dst_h, 256), n);
        hipEvent_t e0 = ctx->k1_start, e1 = ctx->k1_stop;
        ctx->k1_start = ctx->k1_stop = nullptr;                // one-shot
        if (content_only && !any_mask) {     // the border already holds the pad value: write the content rows only
            if (ctx->k1_nt < 0) { const char *e = getenv("RVA_K1_NT"); ctx->k1_nt = e ? atoi(e) & 3 : 0; }
            const bool okc = out_dtype == RVA_F16 ? launch_content<__half>(R, n, stream, a, e0, e1, ctx->k1_nt) : launch_content<float>(R, n, stream, a, e0, e1);
            if (okc) {
                RVA_HIP(ctx, hipGetLastError());
                return RVA_OK;
            }
        }
        const bool ok = any_mask ? (out_dtype == RVA_F16 ? launch_ratio<__half, 8, true>(R, grid, stream, a, e0, e1)
                                                          : launch_ratio<float, 8, true>(R, grid, stream, a, e0, e1))
                      : px16 ? launch_ratio<__half, 16>(R, grid, stream, a, e0, e1)
                             : (out_dtype == RVA_F16 ? launch_ratio<__half, 8>(R, grid, stream, a, e0, e1)
                                                     : launch_ratio<float, 8>(R, grid, stream, a, e0, e1));
        if (ok) {
            RVA_HIP(ctx, hipGetLastError());
            return RVA_OK;
        }
    }
    rva_resize_table tx, ty;
    int rc = rva_get_taps(ctx, src_w, m.new_w, true, &tx);
    if (rc != RVA_OK) return rc;
    rc = rva_get_taps(ctx, src_h, m.new_h, false, &ty);
    if (rc != RVA_OK) return rc;
    a.xofs = tx.ofs; a.xw0 = tx.w0; a.xw1 = tx.w1;
    a.yofs = ty.ofs; a.yw0 = ty.w0; a.yw1 = ty.w1;
    if (mode != 2 && !out_aligned && (dst_w % 8 == 0))  // vector stores need 16-byte rows
        return rva_fail(ctx, RVA_ERR_ARG, "output tensor must be 16-byte aligned");
    // clip heads at heavy decimation (4K -> 224 x 224): one output pixel per thread
    const bool px1 = mode == 1 && out_dtype != RVA_F64 && (long)src_w >= 4L * dst_w;
    dim3 grid(rva_ceil_div(rva_ceil_div(dst_w, px1 ? 1 : 8) * dst_h, 256), n);
    // one-shot profiling events (rva_profile_next_preprocess) bracket the generic kernel too
    hipEvent_t ge0 = ctx->k1_start, ge1 = ctx->k1_stop;
    ctx->k1_start = ctx->k1_stop = nullptr;
#define RVA_LAUNCH_GENERIC_T(NV, MD, T)                                                                           \
    do {                                                                                                          \
        if (ge0 && ge1) hipExtLaunchKernelGGL((k1_generic<NV, MD, T>), grid, dim3(256), 0, stream, ge0, ge1, 0, a); \
        else k1_generic<NV, MD, T><<<grid, 256, 0, stream>>>(a);                                                  \
    } while (0)
#define RVA_LAUNCH_GENERIC(NV, MD)                                                         \
    do {                                                                                   \
        if (out_dtype == RVA_F16) RVA_LAUNCH_GENERIC_T(NV, MD, __half);                    \
        else RVA_LAUNCH_GENERIC_T(NV, MD, float);                                          \
    } while (0)
#define RVA_LAUNCH_CLIP1(NV)                                                                                       \
    do {                                                                                                           \
        if (out_dtype == RVA_F16) {                                                                                \
            if (ge0 && ge1) hipExtLaunchKernelGGL((k1_generic<NV, 1, __half, 1>), grid, dim3(256), 0, stream, ge0, ge1, 0, a); \
            else k1_generic<NV, 1, __half, 1><<<grid, 256, 0, stream>>>(a);                                        \
        } else {                                                                                                   \
            if (ge0 && ge1) hipExtLaunchKernelGGL((k1_generic<NV, 1, float, 1>), grid, dim3(256), 0, stream, ge0, ge1, 0, a);  \
            else k1_generic<NV, 1, float, 1><<<grid, 256, 0, stream>>>(a);                                         \
        }                                                                                                          \
    } while (0)
    if (mode == 2) { if (nv12) RVA_LAUNCH_GENERIC(true, 2); else RVA_LAUNCH_GENERIC(false, 2); }
    else if (clip && out_dtype == RVA_F64) {
        if (nv12) RVA_LAUNCH_GENERIC_T(true, 1, double);
        else RVA_LAUNCH_GENERIC_T(false, 1, double);
    }
    else if (px1) { if (nv12) RVA_LAUNCH_CLIP1(true); else RVA_LAUNCH_CLIP1(false); }
    else if (nv12 && clip) RVA_LAUNCH_GENERIC(true, 1);
    else if (nv12) RVA_LAUNCH_GENERIC(true, 0);
    else if (clip) RVA_LAUNCH_GENERIC(false, 1);
    else RVA_LAUNCH_GENERIC(false, 0);
#undef RVA_LAUNCH_GENERIC
#undef RVA_LAUNCH_GENERIC_T
#undef RVA_LAUNCH_CLIP1
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

}  // namespace

extern "C" {

int rva_preprocess_nv12_content_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                      const int32_t *pitches, int n, int src_w, int src_h, void *out, int out_dtype,
                                      int dst_w, int dst_h, rva_letterbox *meta_out, rva_stream_t stream)
{
    return preprocess_common(ctx, true, 0, y_ptrs, uv_ptrs, pitches, n, src_w, src_h, out, out_dtype, dst_w, dst_h, meta_out,
                             (hipStream_t)stream, nullptr, 0, 0, true);
}

int rva_preprocess_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                              const int32_t *pitches, int n, int src_w, int src_h, void *out, int out_dtype,
                              int dst_w, int dst_h, rva_letterbox *meta_out, rva_stream_t stream)
{
    return preprocess_common(ctx, true, 0, y_ptrs, uv_ptrs, pitches, n, src_w, src_h, out, out_dtype, dst_w, dst_h,
                             meta_out, (hipStream_t)stream);
}

int rva_preprocess_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes, int n, int src_w,
                             int src_h, void *out, int out_dtype, int dst_w, int dst_h, rva_letterbox *meta_out,
                             rva_stream_t stream)
{
    return preprocess_common(ctx, false, 0, frames, nullptr, row_bytes, n, src_w, src_h, out, out_dtype, dst_w, dst_h,
                             meta_out, (hipStream_t)stream);
}

int rva_preprocess_clip_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                   const int32_t *pitches, int n, int src_w, int src_h, void *out, int out_dtype,
                                   int dst_w, int dst_h, rva_stream_t stream)
{
    return preprocess_common(ctx, true, 1, y_ptrs, uv_ptrs, pitches, n, src_w, src_h, out, out_dtype, dst_w, dst_h,
                             nullptr, (hipStream_t)stream);
}

int rva_preprocess_clip_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes, int n, int src_w,
                                  int src_h, void *out, int out_dtype, int dst_w, int dst_h, rva_stream_t stream)
{
    return preprocess_common(ctx, false, 1, frames, nullptr, row_bytes, n, src_w, src_h, out, out_dtype, dst_w, dst_h,
                             nullptr, (hipStream_t)stream);
}

int rva_profile_next_preprocess(rva_ctx *ctx, void *start_event, void *stop_event)
{
    if (!ctx) return RVA_ERR_ARG;
    ctx->k1_start = (hipEvent_t)start_event;
    ctx->k1_stop = (hipEvent_t)stop_event;
    return RVA_OK;
}

int rva_preprocess_frames_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                     const int32_t *pitches, int n, int src_w, int src_h, void *out, int out_dtype,
                                     int dst_w, int dst_h, int norm, int layout, rva_stream_t stream)
{
    return preprocess_common(ctx, true, 1, y_ptrs, uv_ptrs, pitches, n, src_w, src_h, out, out_dtype, dst_w, dst_h,
                             nullptr, (hipStream_t)stream, nullptr, norm, layout);
}

int rva_preprocess_frames_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes, int n, int src_w,
                                    int src_h, void *out, int out_dtype, int dst_w, int dst_h, int norm, int layout,
                                    rva_stream_t stream)
{
    return preprocess_common(ctx, false, 1, frames, nullptr, row_bytes, n, src_w, src_h, out, out_dtype, dst_w, dst_h,
                             nullptr, (hipStream_t)stream, nullptr, norm, layout);
}

// ---- ROI mask + downsample in front of the detector ------------------------------------------
int rva_preprocess_nv12_masked_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                     const int32_t *pitches, const void *const *masks, int n, int src_w, int src_h,
                                     void *out, int out_dtype, int dst_w, int dst_h, rva_letterbox *meta_out,
                                     rva_stream_t stream)
{
    return preprocess_common(ctx, true, 0, y_ptrs, uv_ptrs, pitches, n, src_w, src_h, out, out_dtype, dst_w, dst_h, meta_out,
                             (hipStream_t)stream, masks);
}

int rva_resize_nv12_to_bgr_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                 const int32_t *pitches, const void *const *masks, int n, int src_w, int src_h,
                                 void *out_bgr, int dst_w, int dst_h, rva_stream_t stream)
{
    return preprocess_common(ctx, true, 2, y_ptrs, uv_ptrs, pitches, n, src_w, src_h, out_bgr, RVA_F16, dst_w, dst_h, nullptr,
                             (hipStream_t)stream, masks);
}

}  // extern "C"

// K5: motion gate for a whole tick of streams, replaces MotionFilter.should_process
// (utils/frame_filter.py:26-40): gray = BGR2GRAY(frame) -> GaussianBlur 5x5 -> absdiff with the previous
// blurred frame -> count(diff > 25).  One launch for all streams; the blurred frame is kept per stream in
// HBM (uint8 [h][w]) as the next tick's reference.  Integer arithmetic identical to the oracle restatement
// (OpenCV's 8-bit paths: gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14, separable [1,4,6,4,1] with
// BORDER_REFLECT_101 and (sum + 128) >> 8).  HBM-bound: reads Y + UV + previous blur, writes the new blur.
#include "rva_internal.h"

namespace {

struct K5Args {
    const uint8_t *y[RVA_MAX_BATCH];
    const uint8_t *uv[RVA_MAX_BATCH];
    const uint8_t *prev[RVA_MAX_BATCH];   // nullptr: first frame of the stream (count = -1)
    const uint8_t *mask[RVA_MAX_BATCH];   // optional ROI mask uint8 [h][w] (apply_roi runs before the motion gate)
    uint8_t *out[RVA_MAX_BATCH];
    int32_t pitch[RVA_MAX_BATCH];
    int w, h;
    int32_t *counts;
};

constexpr int TW = 64, TH = 16, HW_ = TW + 4, HH_ = TH + 4;

__device__ __forceinline__ int refl101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

template <bool NV12>
__global__ void __launch_bounds__(256) k5_motion(K5Args a)
{
    __shared__ uint8_t gray[HH_][HW_ + 4];
    __shared__ uint16_t hsum[HH_][TW];
    __shared__ int wsum[4];
    const int s = blockIdx.z, tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const uint8_t *yp = a.y[s], *uvp = a.uv[s], *mk = a.mask[s];
    const int pitch = a.pitch[s];
    for (int i = tid; i < HH_ * HW_; i += 256) {
        const int r = i / HW_, c = i - r * HW_;
        const int py = refl101(y0 + r - 2, a.h), px = refl101(x0 + c - 2, a.w);
        int B, G, R;
        if (NV12) {
            const int Y = yp[(size_t)py * pitch + px];
            const uint8_t *u = uvp + (size_t)(py >> 1) * pitch + ((px >> 1) << 1);
            const int cc = 298 * (Y - 16), d = u[0] - 128, e = u[1] - 128;
            B = clip8((cc + 516 * d + 128) >> 8); G = clip8((cc - 100 * d - 208 * e + 128) >> 8); R = clip8((cc + 409 * e + 128) >> 8);
        } else {
            const uint8_t *q = yp + (size_t)py * pitch + (size_t)px * 3;
            B = q[0]; G = q[1]; R = q[2];
        }
        if (mk && mk[(size_t)py * a.w + px] == 0) B = G = R = 0;
        gray[r][c] = (uint8_t)((B * 1868 + G * 9617 + R * 4899 + 8192) >> 14);
    }
    __syncthreads();
    for (int i = tid; i < HH_ * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        hsum[r][c] = (uint16_t)(gray[r][c] + 4 * gray[r][c + 1] + 6 * gray[r][c + 2] + 4 * gray[r][c + 3] + gray[r][c + 4]);
    }
    __syncthreads();
    int cnt = 0;
    const uint8_t *prev = a.prev[s];
    uint8_t *out = a.out[s];
    for (int i = tid; i < TH * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < a.h && gx < a.w) {
            const int v = (hsum[r][c] + 4 * hsum[r + 1][c] + 6 * hsum[r + 2][c] + 4 * hsum[r + 3][c] + hsum[r + 4][c] + 128) >> 8;
            const size_t o = (size_t)gy * a.w + gx;
            if (prev) { int d = v - (int)prev[o]; d = d < 0 ? -d : d; cnt += d > 25; }
            out[o] = (uint8_t)v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0 && prev) atomicAdd(&a.counts[s], wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

__global__ void k5_init(int32_t *counts, K5Args a, int n)
{
    const int i = threadIdx.x;
    if (i < n) counts[i] = a.prev[i] ? 0 : -1;
}

}  // namespace

static int motion_common(rva_ctx *ctx, bool nv12, const void *const *y_ptrs, const void *const *uv_ptrs,
                         const int32_t *pitches, const void *const *masks, const void *const *prev_blur, void *const *blur_out,
                         int n, int w, int h, int32_t *counts, rva_stream_t stream_)
{
    if (!ctx || !y_ptrs || (nv12 && !uv_ptrs) || !pitches || !prev_blur || !blur_out || !counts || n <= 0 || n > RVA_MAX_BATCH ||
        w < 3 || h < 3 || (nv12 && ((w | h) & 1)))
        return rva_fail(ctx, RVA_ERR_ARG, "rva_motion_*_batch: bad argument");
    K5Args a{};
    for (int i = 0; i < n; ++i) {
        if (!y_ptrs[i] || (nv12 && !uv_ptrs[i]) || !blur_out[i] || pitches[i] < (nv12 ? w : 3 * w))
            return rva_fail(ctx, RVA_ERR_ARG, "rva_motion_*_batch: bad surface %d", i);
        a.y[i] = (const uint8_t *)y_ptrs[i]; a.uv[i] = nv12 ? (const uint8_t *)uv_ptrs[i] : nullptr;
        a.prev[i] = (const uint8_t *)prev_blur[i]; a.out[i] = (uint8_t *)blur_out[i]; a.pitch[i] = pitches[i];
        a.mask[i] = masks ? (const uint8_t *)masks[i] : nullptr;
    }
    a.w = w; a.h = h; a.counts = counts;
    hipStream_t s = (hipStream_t)stream_;
    k5_init<<<1, 64, 0, s>>>(counts, a, n);
    dim3 grid(rva_ceil_div(w, TW), rva_ceil_div(h, TH), n);
    if (nv12) k5_motion<true><<<grid, 256, 0, s>>>(a);
    else k5_motion<false><<<grid, 256, 0, s>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

extern "C" int rva_motion_nv12_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                     const int32_t *pitches, const void *const *prev_blur, void *const *blur_out, int n,
                                     int w, int h, int32_t *counts, rva_stream_t stream)
{
    return motion_common(ctx, true, y_ptrs, uv_ptrs, pitches, nullptr, prev_blur, blur_out, n, w, h, counts, stream);
}

// masks[i] (may be NULL): ROI mask applied first, as apply_roi precedes the gate (pipeline.py:149-158)
extern "C" int rva_motion_nv12_masked_batch(rva_ctx *ctx, const void *const *y_ptrs, const void *const *uv_ptrs,
                                            const int32_t *pitches, const void *const *masks, const void *const *prev_blur,
                                            void *const *blur_out, int n, int w, int h, int32_t *counts, rva_stream_t stream)
{
    return motion_common(ctx, true, y_ptrs, uv_ptrs, pitches, masks, prev_blur, blur_out, n, w, h, counts, stream);
}

// frames[i]: uint8 BGR [h][w][3] device images (e.g. the downsampled frames of rva_resize_nv12_to_bgr_batch)
extern "C" int rva_motion_bgr_batch(rva_ctx *ctx, const void *const *frames, const int32_t *row_bytes,
                                    const void *const *prev_blur, void *const *blur_out, int n, int w, int h,
                                    int32_t *counts, rva_stream_t stream)
{
    return motion_common(ctx, false, frames, nullptr, row_bytes, nullptr, prev_blur, blur_out, n, w, h, counts, stream);
}

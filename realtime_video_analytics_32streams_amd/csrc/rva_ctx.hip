// Context, geometry and scratch management of librva (host side only; no kernels here).
#include <cmath>
#include <cstring>
#include <mutex>
#include <utility>

#include "rva_internal.h"

hipError_t rva_func_smem(const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    size_t &have = done[std::make_pair(dev, fn)];
    if (bytes > have) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        have = bytes;
    }
    return hipSuccess;
}

extern "C" {

int rva_abi_version(void) { return RVA_ABI_VERSION; }

int rva_create(int device, rva_ctx **out)
{
    if (!out) return RVA_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return RVA_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return RVA_ERR_HIP;
    rva_ctx *ctx = new rva_ctx();
    ctx->device = device;
    if (hipMalloc(&ctx->post_flags, sizeof(int32_t) * (2 + RVA_MAX_BATCH)) != hipSuccess ||
        hipMemset(ctx->post_flags, 0, sizeof(int32_t) * (2 + RVA_MAX_BATCH)) != hipSuccess) {
        delete ctx;
        return RVA_ERR_HIP;
    }
    *out = ctx;
    return RVA_OK;
}

static void free_taps(std::map<uint64_t, rva_resize_table> &m)
{
    for (auto &kv : m) {
        (void)hipFree(kv.second.ofs);
        (void)hipFree(kv.second.w0);
        (void)hipFree(kv.second.w1);
    }
    m.clear();
}

static void free_post(rva_ctx *ctx)
{
    (void)hipFree(ctx->sp_box);
    (void)hipFree(ctx->sp_score);
    (void)hipFree(ctx->sp_cls);
    (void)hipFree(ctx->cand_bits);
    ctx->sp_box = ctx->sp_score = nullptr;
    ctx->sp_cls = nullptr;
    ctx->cand_bits = nullptr;
    ctx->cap_batch = ctx->cap_anchors = 0;
}

void rva_destroy(rva_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_post(ctx);
    rva_jpeg_free(ctx);
    free_taps(ctx->taps_x);
    free_taps(ctx->taps_y);
    (void)hipFree(ctx->post_flags);
    delete ctx;
}

const char *rva_last_error(const rva_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rva_reserve(rva_ctx *ctx, int batch, int anchors)
{
    if (!ctx || batch <= 0 || anchors <= 0) return rva_fail(ctx, RVA_ERR_ARG, "rva_reserve: bad sizes");
    if (batch <= ctx->cap_batch && anchors <= ctx->cap_anchors) return RVA_OK;
    RVA_HIP(ctx, hipSetDevice(ctx->device));
    RVA_HIP(ctx, hipDeviceSynchronize());
    int b = batch > ctx->cap_batch ? batch : ctx->cap_batch;
    int a = anchors > ctx->cap_anchors ? anchors : ctx->cap_anchors;
    free_post(ctx);
    size_t ba = (size_t)b * a;
    RVA_HIP(ctx, hipMalloc(&ctx->sp_box, ba * 4 * sizeof(float)));
    RVA_HIP(ctx, hipMalloc(&ctx->sp_score, ba * sizeof(float)));
    RVA_HIP(ctx, hipMalloc(&ctx->sp_cls, ba * sizeof(int32_t)));
    RVA_HIP(ctx, hipMalloc(&ctx->cand_bits, (size_t)b * rva_ceil_div(a, 32) * sizeof(uint32_t) + 8));
    ctx->cap_batch = b;
    ctx->cap_anchors = a;
    return RVA_OK;
}

// detector.py:209-230
int rva_letterbox_meta(int src_w, int src_h, int dst_w, int dst_h, rva_letterbox *out)
{
    if (!out || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0) return RVA_ERR_ARG;
    double sx = (double)dst_w / (double)src_w, sy = (double)dst_h / (double)src_h;
    double s = sx < sy ? sx : sy;
    out->src_w = src_w; out->src_h = src_h; out->dst_w = dst_w; out->dst_h = dst_h;
    out->scale = s;
    out->new_w = (int)((double)src_w * s);
    out->new_h = (int)((double)src_h * s);
    out->pad_left = (dst_w - out->new_w) / 2;
    out->pad_top = (dst_h - out->new_h) / 2;
    return RVA_OK;
}

}  // extern "C"

// OpenCV 4.x resize() tap computation for INTER_LINEAR on 8-bit data (detector.py:218-222 calls it;
// imgproc resize.cpp: fx=(dx+0.5)*scale-0.5 in double, cast to float, floor, 11-bit weights by
// round-half-even).  Columns clamp the offset and zero the fraction at the borders; rows keep the
// raw offset and are clamped where they are used.
int rva_get_taps(rva_ctx *ctx, int src, int dst, bool is_x, rva_resize_table *out)
{
    auto &cache = is_x ? ctx->taps_x : ctx->taps_y;
    uint64_t key = ((uint64_t)(uint32_t)src << 32) | (uint32_t)dst;
    auto it = cache.find(key);
    if (it != cache.end()) { *out = it->second; return RVA_OK; }
    std::vector<int32_t> ofs(dst);
    std::vector<int16_t> w0(dst), w1(dst);
    double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (is_x) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= src - 1) { f = 0.f; s = src - 1; }
        }
        auto q = [](float v) { long r = lrintf(v); return (int16_t)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r)); };
        ofs[d] = s;
        w0[d] = q((1.f - f) * 2048.f);
        w1[d] = q(f * 2048.f);
    }
    rva_resize_table t;
    t.n = dst;
    RVA_HIP(ctx, hipMalloc(&t.ofs, dst * sizeof(int32_t)));
    RVA_HIP(ctx, hipMalloc(&t.w0, dst * sizeof(int16_t)));
    RVA_HIP(ctx, hipMalloc(&t.w1, dst * sizeof(int16_t)));
    RVA_HIP(ctx, hipMemcpy(t.ofs, ofs.data(), dst * sizeof(int32_t), hipMemcpyHostToDevice));
    RVA_HIP(ctx, hipMemcpy(t.w0, w0.data(), dst * sizeof(int16_t), hipMemcpyHostToDevice));
    RVA_HIP(ctx, hipMemcpy(t.w1, w1.data(), dst * sizeof(int16_t), hipMemcpyHostToDevice));
    cache[key] = t;
    *out = t;
    return RVA_OK;
}

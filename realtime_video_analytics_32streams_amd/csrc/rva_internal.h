// Internal definitions shared by the librva translation units (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "rva.h"

struct rva_resize_table {  // device copy of the per-axis resize taps for one geometry
    int32_t *ofs = nullptr;   // [n]   first source index
    int16_t *w0 = nullptr;    // [n]   11-bit fixed-point weight of tap 0
    int16_t *w1 = nullptr;    // [n]   ... of tap 1
    int n = 0;
};

struct rva_geom_cache {
    rva_resize_table x, y;
};

struct rva_jpeg_state;           // scratch of the device JPEG encoder (rva_jpeg.hip)

struct rva_ctx {
    int device = 0;
    rva_jpeg_state *jpeg = nullptr;
    std::string err;
    // post-process scratch, sized by rva_reserve / grown on demand
    int cap_batch = 0, cap_anchors = 0;
    float *sp_box = nullptr;      // [B][A][4] xyxy of thresholded anchors (sparse, anchor-indexed)
    float *sp_score = nullptr;    // [B][A]
    int32_t *sp_cls = nullptr;    // [B][A]
    uint32_t *cand_bits = nullptr;// [B][ceil(A/32)] pass bitmap in anchor order
    int32_t *post_flags = nullptr;// [2 + RVA_MAX_BATCH]: overflow flags, K2's improper-box flag per image of the launch, images NMS'd with the centre-bin filter
    // resize tap tables keyed by (src, dst) per axis
    std::map<uint64_t, rva_resize_table> taps_x, taps_y;
    // one-shot profiling events for the next K1 (integer-ratio) launch: rva_profile_next_preprocess
    hipEvent_t k1_start = nullptr, k1_stop = nullptr;
    int k1_px = -1;               // RVA_K1_PX tuning switch, read once per context
    int k1_nt = -1;               // RVA_K1_NT experiment switch (non-temporal loads / stores in the steady-state K1), read once per context
    int num_cus = 0;              // multiProcessorCount of ctx->device (persistent-grid sizing)
};

inline int rva_fail(rva_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define RVA_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return rva_fail((ctx), RVA_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                    \
    } while (0)

// Host restatement of the OpenCV resize tap computation; fills a device table (cached in ctx).
int rva_get_taps(rva_ctx *ctx, int src, int dst, bool is_x, rva_resize_table *out);

static inline int rva_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) to at least `bytes` on the CURRENT
// device.  The attribute belongs to the (device, function) pair, so the bookkeeping is keyed by both: a second context
// on another device of the same process gets its own call (a process-global `static bool` would skip it).  Must not be
// reached for the first time inside a stream capture: callers run one eager launch of every kernel before capturing.
hipError_t rva_func_smem(const void *fn, size_t bytes);

// frees ctx->jpeg (rva_jpeg.hip); called by rva_destroy
void rva_jpeg_free(rva_ctx *ctx);

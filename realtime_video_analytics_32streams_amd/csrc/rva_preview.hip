// K6: annotated preview image of a frame, replacing the pixel work of KafkaSink._render_frame
// (sinks/kafka_sink.py:200-294: frame.copy() -> [cv2.resize INTER_AREA when larger than 1920x1080] -> per track
// cv2.rectangle outline + filled label bar + cv2.putText) and of StreamWorker._maybe_save_snapshot (pipeline.py:264-290).
// The reference copies every frame to the host and draws there; here the NV12 surface stays in HBM, one launch produces
// the uint8 BGR preview (converted, box-averaged down by an integer ratio, annotated) and only that small image is copied
// out for the encoder.  What to draw -- rectangles, colours, label origins, target size -- is decided on the host by
// preview.py and is pinned against a call-level recording of the reference; HOW a primitive covers pixels (OpenCV's line
// rasteriser, Hershey glyphs, the AREA resampler at non-integer ratios) is OpenCV-internal and unpinned here: filled
// inclusive rectangles, a 5x7 bitmap font, box mean (sum + n/2) / n -- the latter is OpenCV's rule at integer ratios.
#include "rva_internal.h"

namespace {

struct K6Args {
    const uint8_t *y, *uv;
    int pitch, src_w, src_h, ratio;      // ratio 0: `out` already holds the base image (draw only)
    uint8_t *out;
    int dst_w, dst_h;
    const int32_t *rects;                // [n][4] x0, y0, x1, y1 inclusive, painter's order
    const uint8_t *colors;               // [n][4] b, g, r, -
    int n_rects;
    const int32_t *glyphs;               // [m][3] x, y (top-left of a 5x7 cell), character code
    int n_glyphs, glyph_scale;
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// 5x7 font: digits, 'I', 'D', space, then 'c', 'l', 's' (the snapshot label "ID<id> cls<class>", pipeline.py:279); row-major,
// bit 4 = leftmost column
__constant__ uint8_t kFont[16][7] = {
    {0x0E, 0x11, 0x13, 0x15, 0x19, 0x11, 0x0E}, {0x04, 0x0C, 0x04, 0x04, 0x04, 0x04, 0x0E}, {0x0E, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1F},
    {0x1F, 0x02, 0x04, 0x02, 0x01, 0x11, 0x0E}, {0x02, 0x06, 0x0A, 0x12, 0x1F, 0x02, 0x02}, {0x1F, 0x10, 0x1E, 0x01, 0x01, 0x11, 0x0E},
    {0x06, 0x08, 0x10, 0x1E, 0x11, 0x11, 0x0E}, {0x1F, 0x01, 0x02, 0x04, 0x08, 0x08, 0x08}, {0x0E, 0x11, 0x11, 0x0E, 0x11, 0x11, 0x0E},
    {0x0E, 0x11, 0x11, 0x0F, 0x01, 0x02, 0x0C}, {0x0E, 0x04, 0x04, 0x04, 0x04, 0x04, 0x0E}, {0x1E, 0x11, 0x11, 0x11, 0x11, 0x11, 0x1E},
    {0, 0, 0, 0, 0, 0, 0},
    {0x00, 0x00, 0x0E, 0x10, 0x10, 0x11, 0x0E}, {0x0C, 0x04, 0x04, 0x04, 0x04, 0x04, 0x0E}, {0x00, 0x00, 0x0F, 0x10, 0x0E, 0x01, 0x1E}};
__device__ __forceinline__ int font_row(int ch)
{
    if (ch >= '0' && ch <= '9') return ch - '0';
    switch (ch) { case 'I': return 10; case 'D': return 11; case 'c': return 13; case 'l': return 14; case 's': return 15; default: return 12; }
}

__global__ void __launch_bounds__(256) k6_preview(K6Args a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.dst_w || yy >= a.dst_h) return;
    uint8_t *o = a.out + ((size_t)yy * a.dst_w + x) * 3;
    int b, g, r;
    if (a.ratio == 0) {
        b = o[0]; g = o[1]; r = o[2];
    } else {
        const int R = a.ratio;
        int sb = 0, sg = 0, sr = 0;
        for (int dy = 0; dy < R; ++dy)
            for (int dx = 0; dx < R; ++dx) {
                const int px = x * R + dx, py = yy * R + dy;
                const int Y = a.y[(size_t)py * a.pitch + px];
                const uint8_t *u = a.uv + (size_t)(py >> 1) * a.pitch + ((px >> 1) << 1);
                const int c = 298 * (Y - 16), d = u[0] - 128, e = u[1] - 128;     // the BT.601 matrix of K1 / K5
                sb += clip8((c + 516 * d + 128) >> 8);
                sg += clip8((c - 100 * d - 208 * e + 128) >> 8);
                sr += clip8((c + 409 * e + 128) >> 8);
            }
        const int n = R * R;
        b = (sb + n / 2) / n; g = (sg + n / 2) / n; r = (sr + n / 2) / n;
    }
    for (int i = 0; i < a.n_rects; ++i) {                      // painter's order: later primitives cover earlier ones
        const int32_t *q = a.rects + 4 * i;
        if (x >= q[0] && x <= q[2] && yy >= q[1] && yy <= q[3]) { b = a.colors[4 * i]; g = a.colors[4 * i + 1]; r = a.colors[4 * i + 2]; }
    }
    const int gs = a.glyph_scale;
    for (int i = 0; i < a.n_glyphs; ++i) {
        const int32_t *q = a.glyphs + 3 * i;
        const int cx = x - q[0], cy = yy - q[1];
        if (cx >= 0 && cy >= 0 && cx < 5 * gs && cy < 7 * gs && ((kFont[font_row(q[2])][cy / gs] >> (4 - cx / gs)) & 1)) b = g = r = 255;
    }
    o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
}

}  // namespace

extern "C" int rva_preview_nv12(rva_ctx *ctx, const void *y, const void *uv, int pitch, int src_w, int src_h, int ratio, void *out_bgr,
                                int dst_w, int dst_h, const int32_t *rects, const uint8_t *colors, int n_rects,
                                const int32_t *glyphs, int n_glyphs, int glyph_scale, rva_stream_t stream)
{
    if (!ctx || !out_bgr || dst_w <= 0 || dst_h <= 0 || n_rects < 0 || n_glyphs < 0 || (n_rects && (!rects || !colors)) ||
        (n_glyphs && !glyphs) || glyph_scale < 1 || ratio < 0)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_preview_nv12: bad argument");
    if (ratio > 0 && (!y || !uv || pitch < src_w || ((src_w | src_h) & 1) || dst_w * ratio > src_w || dst_h * ratio > src_h || ratio > 8))
        return rva_fail(ctx, RVA_ERR_ARG, "rva_preview_nv12: the surface does not cover dst x ratio (even NV12 size, ratio <= 8)");
    K6Args a{(const uint8_t *)y, (const uint8_t *)uv, pitch, src_w, src_h, ratio, (uint8_t *)out_bgr, dst_w, dst_h,
             rects, colors, n_rects, glyphs, n_glyphs, glyph_scale};
    dim3 grid(rva_ceil_div(dst_w, 64), rva_ceil_div(dst_h, 4));
    k6_preview<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

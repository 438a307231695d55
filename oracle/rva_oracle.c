/*
 * rva_oracle.c -- CPU restatement of the reference's detect/track hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (librva.so) never links or calls it.
 *
 * Pinning status (DESIGN.md "Oracle"):
 *   post-process + NMS, tracker, clip buffering : PINNED -- bit-exact against tests/golden/ JSON files,
 *       which oracle/gen_golden.py produced by executing the reference's own functions.
 *   pre-process (resize / colour / letterbox)   : PARITY UNPINNED -- OpenCV and FFmpeg are absent
 *       here and the reference holds no fixture; this restates the published OpenCV 4.x
 *       INTER_LINEAR 8-bit algorithm and a BT.601 limited-range integer matrix.
 *
 * Every function cites the reference lines (relative to /root/reference/src/realtime_analytics/)
 * it follows.  Compile with -ffp-contract=off: the reference's numpy/Python arithmetic never fuses
 * a multiply with an add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ letterbox geometry */
/* detector.py:209-230: scale=min(tw/w, th/h); new=int(w*scale); pad=target-new; top=pad_h//2 */
ORC_API void orc_letterbox(int w, int h, int tw, int th, double *scale, int *new_w, int *new_h,
                           int *left, int *top)
{
    double sx = (double)tw / (double)w, sy = (double)th / (double)h;
    double s = sx < sy ? sx : sy;
    int nw = (int)((double)w * s), nh = (int)((double)h * s);
    *scale = s;
    *new_w = nw;
    *new_h = nh;
    *left = (tw - nw) / 2; /* pads are >= 0 here, so C division == Python // */
    *top = (th - nh) / 2;
}

/* ------------------------------------------------------------------ post-process + NMS */
/* detector.py:282-283: a 2-D head [d1,d2] is transposed iff d1 != 0 and d1 < d2.
 * Returns 1 when the anchor axis is the LAST one (layout [C, A]). */
ORC_API int orc_head_is_channel_major(int d1, int d2) { return d1 != 0 && d1 < d2; }

typedef struct {
    float score;
    int anchor;
    int cand;
} orc_key;

static int key_cmp(const void *pa, const void *pb)
{
    const orc_key *a = (const orc_key *)pa, *b = (const orc_key *)pb;
    if (a->score > b->score) return -1; /* detector.py:365 argsort()[::-1]: descending score */
    if (a->score < b->score) return 1;
    return a->anchor < b->anchor ? -1 : (a->anchor > b->anchor); /* ties: project rule, anchor asc */
}

/* detector.py:469-481 (_iou), float32 throughout, a = the kept box */
static float iou_f32(const float *a, const float *b)
{
    float x1 = a[0] > b[0] ? a[0] : b[0];
    float y1 = a[1] > b[1] ? a[1] : b[1];
    float x2 = a[2] < b[2] ? a[2] : b[2];
    float y2 = a[3] < b[3] ? a[3] : b[3];
    float w = x2 - x1, h = y2 - y1;
    if (!(w > 0.0f)) w = 0.0f; /* np.maximum(0, .) */
    if (!(h > 0.0f)) h = 0.0f;
    float inter = w * h;
    float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    float uni = area_a + area_b - inter;
    if (uni < 1e-6f) uni = 1e-6f; /* np.clip(union, 1e-6, None) with a weak python scalar */
    return inter / uni;
}

/*
 * detector.py:266-338.  pred is read through (stride_a, stride_c) so both [A,C] and [C,A]
 * layouts are served without a copy.  Outputs are in NMS order (descending score):
 *   out_anchor[i]  anchor row of the i-th kept detection
 *   out_keep[i]    its index among the thresholded candidates (the reference's `keep` value)
 *   out_cls/out_conf/out_box (xyxy, frame pixels, clipped)
 * Returns the number kept; *n_cand = number of candidates that passed the threshold.
 * If cand_* are non-NULL they receive the candidate list in anchor order (size up to A).
 */
ORC_API int orc_postprocess(const float *pred, int A, int C, long stride_a, long stride_c,
                            double conf_thr, double iou_thr, const int *classes, int n_classes,
                            int orig_w, int orig_h, double scale, int left, int top,
                            int *out_anchor, int *out_keep, int *out_cls, float *out_conf,
                            float *out_box, int *n_cand)
{
    *n_cand = 0;
    if (C < 5 || A <= 0) return 0; /* :285-287 */
    const float thr = (float)conf_thr; /* NEP-50: python float vs float32 array -> float32 */
    const float iou_t = (float)iou_thr;
    const float fscale = (float)scale, fleft = (float)left, ftop = (float)top;
    const float xmax = (float)(orig_w - 1), ymax = (float)(orig_h - 1);

    float *box = (float *)malloc((size_t)A * 4 * sizeof(float));
    float *conf = (float *)malloc((size_t)A * sizeof(float));
    int *cls = (int *)malloc((size_t)A * sizeof(int));
    int *anc = (int *)malloc((size_t)A * sizeof(int));
    int K = 0;
    for (int a = 0; a < A; ++a) {
        const float *p = pred + (long)a * stride_a;
        float best;
        int bi = 0;
        if (C > 5) { /* :294-305 both branches: scores = pred[:,5:] * pred[:,4:5] */
            float obj = p[4 * stride_c];
            best = p[5 * stride_c] * obj;
            for (int c = 6; c < C; ++c) {
                float s = p[(long)c * stride_c] * obj;
                if (s > best) { best = s; bi = c - 5; } /* np.argmax: first maximum wins */
            }
        } else { /* :306-307 scores = pred[:,4:] */
            best = p[4 * stride_c];
        }
        if (!(best >= thr)) continue; /* :312 */
        if (n_classes > 0) {          /* :313-314 */
            int ok = 0;
            for (int i = 0; i < n_classes; ++i) ok |= (classes[i] == bi);
            if (!ok) continue;
        }
        float cx = p[0], cy = p[1 * stride_c], w = p[2 * stride_c], h = p[3 * stride_c];
        float b[4];
        b[0] = cx - w / 2.0f; /* :352-359 */
        b[1] = cy - h / 2.0f;
        b[2] = cx + w / 2.0f;
        b[3] = cy + h / 2.0f;
        b[0] -= fleft; b[2] -= fleft; /* :345-346 */
        b[1] -= ftop;  b[3] -= ftop;
        for (int i = 0; i < 4; ++i) b[i] = b[i] / fscale; /* :347 true division */
        for (int i = 0; i < 4; ++i) {                      /* :348-349 np.clip = min(max(x,lo),hi) */
            float hi = (i & 1) ? ymax : xmax;
            float v = b[i];
            v = v < 0.0f ? 0.0f : v;
            v = v > hi ? hi : v;
            b[i] = v;
        }
        memcpy(box + 4 * (size_t)K, b, sizeof b);
        conf[K] = best; cls[K] = bi; anc[K] = a;
        ++K;
    }
    *n_cand = K;
    int kept = 0;
    if (K > 0) {
        orc_key *ord = (orc_key *)malloc((size_t)K * sizeof(orc_key));
        for (int i = 0; i < K; ++i) { ord[i].score = conf[i]; ord[i].anchor = anc[i]; ord[i].cand = i; }
        qsort(ord, (size_t)K, sizeof(orc_key), key_cmp);
        /* :361-375 greedy: keep head, survivors are iou <= thr (NaN does not survive) */
        unsigned char *dead = (unsigned char *)calloc((size_t)K, 1);
        for (int i = 0; i < K; ++i) {
            if (dead[i]) continue;
            int ci = ord[i].cand;
            out_anchor[kept] = anc[ci];
            out_keep[kept] = ci;
            out_cls[kept] = cls[ci];
            out_conf[kept] = conf[ci];
            memcpy(out_box + 4 * (size_t)kept, box + 4 * (size_t)ci, 4 * sizeof(float));
            ++kept;
            for (int j = i + 1; j < K; ++j) {
                if (dead[j]) continue;
                float v = iou_f32(box + 4 * (size_t)ci, box + 4 * (size_t)ord[j].cand);
                if (!(v <= iou_t)) dead[j] = 1;
            }
        }
        free(dead);
        free(ord);
    }
    free(box); free(conf); free(cls); free(anc);
    return kept;
}

/* ------------------------------------------------------------------ IoU tracker */
typedef struct {
    int64_t id;
    int32_t cls, age, hits;
    double conf;
    double box[4];
    int32_t matched;
} orc_track;

typedef struct {
    orc_track *t;
    int n, cap;
} orc_stream;

typedef struct {
    int n_streams, max_age, min_hits;
    double min_iou;
    int64_t next_id; /* tracker.py:47 itertools.count(1): ONE counter for all streams */
    orc_stream *s;
} orc_tracker;

ORC_API orc_tracker *orc_tracker_new(int n_streams, int max_age, double min_iou, int min_hits)
{
    orc_tracker *k = (orc_tracker *)calloc(1, sizeof *k);
    k->n_streams = n_streams; k->max_age = max_age; k->min_hits = min_hits; k->min_iou = min_iou;
    k->next_id = 1;
    k->s = (orc_stream *)calloc((size_t)n_streams, sizeof(orc_stream));
    return k;
}

ORC_API void orc_tracker_free(orc_tracker *k)
{
    if (!k) return;
    for (int i = 0; i < k->n_streams; ++i) free(k->s[i].t);
    free(k->s);
    free(k);
}

ORC_API int64_t orc_tracker_next_id(const orc_tracker *k) { return k->next_id; }
ORC_API void orc_tracker_set_next_id(orc_tracker *k, int64_t v) { k->next_id = v; }

/* tracker.py:129-147, float64, a = track, b = detection */
static double iou_f64(const double *a, const double *b)
{
    double ix1 = a[0] > b[0] ? a[0] : b[0];
    double iy1 = a[1] > b[1] ? a[1] : b[1];
    double ix2 = a[2] < b[2] ? a[2] : b[2];
    double iy2 = a[3] < b[3] ? a[3] : b[3];
    double iw = ix2 - ix1, ih = iy2 - iy1;
    iw = iw > 0.0 ? iw : 0.0;
    ih = ih > 0.0 ? ih : 0.0;
    double inter = iw * ih;
    double aw = a[2] - a[0], ah = a[3] - a[1], bw = b[2] - b[0], bh = b[3] - b[1];
    aw = aw > 0.0 ? aw : 0.0; ah = ah > 0.0 ? ah : 0.0;
    bw = bw > 0.0 ? bw : 0.0; bh = bh > 0.0 ? bh : 0.0;
    double uni = aw * ah + bw * bh - inter;
    if (uni <= 0.0) return 0.0;
    return inter / uni;
}

/*
 * tracker.py:50-126.  One update of one stream: D detections in order.  Returns the number of
 * surviving tracks (insertion order) and copies up to out_cap of them to the out_* arrays.
 * *n_new = tracks created by this call.
 */
ORC_API int orc_tracker_update(orc_tracker *k, int stream, int D, const double *boxes,
                               const double *conf, const int64_t *cls, int out_cap,
                               int64_t *out_id, int32_t *out_cls, int32_t *out_age,
                               int32_t *out_hits, double *out_conf, double *out_box, int *n_new)
{
    orc_stream *st = &k->s[stream];
    int created = 0;
    for (int i = 0; i < st->n; ++i) st->t[i].matched = 0;
    for (int d = 0; d < D; ++d) {
        const double *b = boxes + 4 * (size_t)d;
        double best = 0.0; /* :100 */
        int bi = -1;
        for (int i = 0; i < st->n; ++i) { /* :102-108 dict order == insertion order */
            orc_track *t = &st->t[i];
            if (t->cls != (int32_t)cls[d]) continue;
            double v = iou_f64(t->box, b);
            if (v >= k->min_iou && v > best) { best = v; bi = i; }
        }
        if (bi < 0) { /* :69-80 new track is inserted immediately (matchable by later dets) */
            if (st->n == st->cap) {
                st->cap = st->cap ? st->cap * 2 : 64;
                st->t = (orc_track *)realloc(st->t, (size_t)st->cap * sizeof(orc_track));
            }
            orc_track *t = &st->t[st->n++];
            t->id = k->next_id++;
            t->cls = (int32_t)cls[d]; t->age = 0; t->hits = 1; t->conf = conf[d];
            memcpy(t->box, b, sizeof t->box);
            t->matched = 1;
            ++created;
        } else { /* :81-92 class_id is NOT updated */
            orc_track *t = &st->t[bi];
            memcpy(t->box, b, sizeof t->box);
            t->conf = conf[d]; t->hits += 1; t->age = 0; t->matched = 1;
        }
    }
    /* :111-126 prune */
    int w = 0;
    for (int i = 0; i < st->n; ++i) {
        orc_track *t = &st->t[i];
        if (!t->matched) {
            t->age += 1;
            if (t->age > k->max_age || t->hits < k->min_hits) continue;
        }
        if (w != i) st->t[w] = *t;
        ++w;
    }
    st->n = w;
    if (n_new) *n_new = created;
    int m = w < out_cap ? w : out_cap;
    for (int i = 0; i < m; ++i) {
        const orc_track *t = &st->t[i];
        out_id[i] = t->id; out_cls[i] = t->cls; out_age[i] = t->age; out_hits[i] = t->hits;
        out_conf[i] = t->conf; memcpy(out_box + 4 * (size_t)i, t->box, sizeof t->box);
    }
    return w;
}

/* ------------------------------------------------------------------ pre-process (UNPINNED, see header) */
static inline uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* D1 stand-in: what cv2.VideoCapture hands over is BGR; our surfaces are NV12.  BT.601
 * limited-range integer matrix, nearest chroma (video_stream.py:76,173 -> swscale default). */
ORC_API void orc_yuv_to_bgr(int Y, int U, int V, uint8_t *bgr)
{
    int c = 298 * (Y - 16), d = U - 128, e = V - 128;
    bgr[0] = clip8((c + 516 * d + 128) >> 8);
    bgr[1] = clip8((c - 100 * d - 208 * e + 128) >> 8);
    bgr[2] = clip8((c + 409 * e + 128) >> 8);
}

ORC_API void orc_nv12_to_bgr(const uint8_t *y, const uint8_t *uv, int pitch, int w, int h, uint8_t *bgr)
{
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            const uint8_t *p = uv + (size_t)(r >> 1) * pitch + (c >> 1) * 2;
            orc_yuv_to_bgr(y[(size_t)r * pitch + c], p[0], p[1], bgr + ((size_t)r * w + c) * 3);
        }
}

static inline short sat_short_round(float v)
{
    long r = lrintf(v); /* default rounding mode: nearest-even == cvRound */
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

/* OpenCV 4.x resize(), INTER_LINEAR, 8-bit: per-axis offset + 11-bit fixed-point weights
 * (imgproc/src/resize.cpp: resize_ coefficient loop; INTER_RESIZE_COEF_BITS = 11). */
ORC_API void orc_resize_coeffs(int src, int dst, int *ofs, short *w0, short *w1, int is_x)
{
    double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (is_x) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= src - 1) { f = 0.f; s = src - 1; }
        }
        ofs[d] = s; /* rows are clamped at use (resizeGeneric_Invoker clip) */
        w0[d] = sat_short_round((1.f - f) * 2048.f);
        w1[d] = sat_short_round(f * 2048.f);
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv2.resize(src, (dw, dh), interpolation=INTER_LINEAR) for uint8 HxWx3 (detector.py:218-222,
 * temporal_detector.py:344).  The 2:1 INTER_AREA shortcut and the 1:1 copy produce the same
 * bytes as this fixed-point path (weights 1024/1024 resp. 2048/0), so one path serves all. */
ORC_API void orc_resize_linear_u8c3(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    int *xo = (int *)malloc(sizeof(int) * (size_t)dw), *yo = (int *)malloc(sizeof(int) * (size_t)dh);
    short *a0 = (short *)malloc(2 * (size_t)dw), *a1 = (short *)malloc(2 * (size_t)dw);
    short *b0 = (short *)malloc(2 * (size_t)dh), *b1 = (short *)malloc(2 * (size_t)dh);
    orc_resize_coeffs(sw, dw, xo, a0, a1, 1);
    orc_resize_coeffs(sh, dh, yo, b0, b1, 0);
    for (int y = 0; y < dh; ++y) {
        const uint8_t *r0 = src + (size_t)clampi(yo[y], 0, sh - 1) * sw * 3;
        const uint8_t *r1 = src + (size_t)clampi(yo[y] + 1, 0, sh - 1) * sw * 3;
        for (int x = 0; x < dw; ++x) {
            int s0 = xo[x], s1 = s0 + 1 < sw ? s0 + 1 : sw - 1;
            for (int c = 0; c < 3; ++c) {
                int h0 = r0[s0 * 3 + c] * a0[x] + r0[s1 * 3 + c] * a1[x]; /* HResizeLinear */
                int h1 = r1[s0 * 3 + c] * a0[x] + r1[s1 * 3 + c] * a1[x];
                int v = (((b0[y] * (h0 >> 4)) >> 16) + ((b1[y] * (h1 >> 4)) >> 16) + 2) >> 2; /* VResizeLinear */
                dst[((size_t)y * dw + x) * 3 + c] = clip8(v);
            }
        }
    }
    free(xo); free(yo); free(a0); free(a1); free(b0); free(b1);
}

/* IEEE binary16 helpers (gcc 11 has no _Float16 on x86) */
static uint16_t f64_to_f16(double v)
{
    if (v != v) return 0x7e00;
    uint16_t sign = 0;
    if (signbit(v)) { sign = 0x8000; v = -v; }
    if (v == 0.0) return sign;
    if (v >= 65520.0) return (uint16_t)(sign | 0x7c00);
    int e;
    double m = frexp(v, &e); /* v = m * 2^e, m in [0.5,1) */
    int exp = e - 1;         /* v = (2m) * 2^exp */
    double scaled;
    int biased;
    if (exp < -14) { scaled = ldexp(v, 24); biased = 0; }         /* subnormal: units of 2^-24 */
    else { scaled = ldexp(m * 2.0 - 1.0, 10); biased = exp + 15; } /* 10 fraction bits */
    double r = nearbyint(scaled); /* nearest-even */
    int frac = (int)r;
    if (biased == 0) return (uint16_t)(sign | frac); /* frac may reach 0x400 == smallest normal */
    if (frac == 1024) { frac = 0; ++biased; }
    if (biased >= 31) return (uint16_t)(sign | 0x7c00);
    return (uint16_t)(sign | (biased << 10) | frac);
}

static double f16_to_f64(uint16_t h)
{
    int e = (h >> 10) & 31, f = h & 1023;
    double v = e == 0 ? ldexp((double)f, -24) : (e == 31 ? (f ? NAN : INFINITY) : ldexp(1.0 + f / 1024.0, e - 15));
    return (h & 0x8000) ? -v : v;
}

ORC_API uint16_t orc_f32_to_f16(float v) { return f64_to_f16((double)v); }
ORC_API float orc_f16_to_f32(uint16_t h) { return (float)f16_to_f64(h); }

/*
 * detector.py:198-264 (_preprocess): letterbox-resize, pad 114, BGR->RGB, astype(dtype)*(1/255),
 * HWC->CHW.  half != 0: uint8.astype(float16) * float16(1/255) is a binary16 multiply (NEP-50),
 * output uint16 bit patterns; else float32 multiply by float32(1/255).
 */
ORC_API void orc_preprocess_bgr(const uint8_t *bgr, int w, int h, int tw, int th, int half, void *out,
                                double *scale, int *left, int *top)
{
    int nw, nh;
    orc_letterbox(w, h, tw, th, scale, &nw, &nh, left, top);
    uint8_t *rs = (uint8_t *)malloc((size_t)nw * nh * 3);
    orc_resize_linear_u8c3(bgr, w, h, rs, nw, nh);
    const float k32 = (float)(1.0 / 255.0);
    const double k16 = f16_to_f64(f64_to_f16(1.0 / 255.0));
    uint16_t lut16[256];
    float lut32[256];
    for (int v = 0; v < 256; ++v) {
        lut32[v] = (float)v * k32;
        lut16[v] = f64_to_f16((double)v * k16); /* product of two binary16 values is exact in double */
    }
    size_t plane = (size_t)tw * th;
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
            int sy = y - *top, sx = x - *left;
            for (int c = 0; c < 3; ++c) { /* c: output channel R,G,B = input channel 2,1,0 */
                int v = 114;
                if (sy >= 0 && sy < nh && sx >= 0 && sx < nw) v = rs[((size_t)sy * nw + sx) * 3 + (2 - c)];
                size_t o = (size_t)c * plane + (size_t)y * tw + x;
                if (half) ((uint16_t *)out)[o] = lut16[v];
                else ((float *)out)[o] = lut32[v];
            }
        }
    free(rs);
}

/* NV12 surface -> the same tensor: D1 stand-in (NV12->BGR at full resolution) then P1. */
ORC_API void orc_preprocess_nv12(const uint8_t *y, const uint8_t *uv, int pitch, int w, int h, int tw,
                                 int th, int half, void *out, double *scale, int *left, int *top)
{
    uint8_t *bgr = (uint8_t *)malloc((size_t)w * h * 3);
    orc_nv12_to_bgr(y, uv, pitch, w, h, bgr);
    orc_preprocess_bgr(bgr, w, h, tw, th, half, out, scale, left, top);
    free(bgr);
}

/*
 * temporal_detector.py:330-373 (CNNLSTMDetector._preprocess_sequence), one frame:
 * cv2.resize(frame,(W,H)) stretch, BGR->RGB, float32 /255.0, (x-mean)/std, CHW; half: cast at the end.
 */
ORC_API void orc_preprocess_clip_frame_bgr(const uint8_t *bgr, int w, int h, int tw, int th, int half, void *out)
{
    static const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    uint8_t *rs = (uint8_t *)malloc((size_t)tw * th * 3);
    orc_resize_linear_u8c3(bgr, w, h, rs, tw, th);
    size_t plane = (size_t)tw * th;
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < plane; ++i) {
            float v = (float)rs[i * 3 + (2 - c)] / 255.0f;
            v = (v - mean[c]) / stdv[c];
            if (half) ((uint16_t *)out)[c * plane + i] = orc_f32_to_f16(v);
            else ((float *)out)[c * plane + i] = v;
        }
    free(rs);
}

/*
 * The same frame body with the constants / precision of the other heads:
 *   norm 0: float32 ImageNet constants  -- temporal_detector.py:350-354 (CNN-LSTM), detector.py:988-993 (ResNet)
 *   norm 1: float32 0.45 / 0.225        -- temporal_detector.py:570-573 (3D-CNN)
 *   norm 2: float64 ImageNet constants  -- temporal_detector.py:741-743 (ConvGRU: np.array([...]) without dtype, so
 *           the float32 image is promoted and (image - mean) / std runs in float64)
 * out_dtype 0 = float16 (cast of the float32 / float64 result, one rounding), 1 = float32, 2 = float64.
 * Channel c is written at out + c * cstride (elements): cstride = th*tw for [C,H,W], T*th*tw inside a [C,T,H,W] clip.
 */
ORC_API void orc_preprocess_norm_frame_bgr(const uint8_t *bgr, int w, int h, int tw, int th, int norm, int out_dtype,
                                           void *out, long cstride)
{
    static const float meanf[3] = {0.485f, 0.456f, 0.406f}, stdf[3] = {0.229f, 0.224f, 0.225f};
    static const double meand[3] = {0.485, 0.456, 0.406}, stdd[3] = {0.229, 0.224, 0.225};
    uint8_t *rs = (uint8_t *)malloc((size_t)tw * th * 3);
    orc_resize_linear_u8c3(bgr, w, h, rs, tw, th);
    size_t plane = (size_t)tw * th;
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < plane; ++i) {
            float x = (float)rs[i * 3 + (2 - c)] / 255.0f;
            double d;
            if (norm == 2) {
                d = ((double)x - meand[c]) / stdd[c];
            } else {
                float m = norm == 1 ? 0.45f : meanf[c], sd = norm == 1 ? 0.225f : stdf[c];
                float r = (x - m) / sd;
                d = (double)r;
            }
            size_t o = (size_t)c * (size_t)cstride + i;
            if (out_dtype == 0) ((uint16_t *)out)[o] = f64_to_f16(d);
            else if (out_dtype == 1) ((float *)out)[o] = (float)d;
            else ((double *)out)[o] = d;
        }
    free(rs);
}

ORC_API void orc_preprocess_norm_frame_nv12(const uint8_t *y, const uint8_t *uv, int pitch, int w, int h, int tw, int th,
                                            int norm, int out_dtype, void *out, long cstride)
{
    uint8_t *bgr = (uint8_t *)malloc((size_t)w * h * 3);
    orc_nv12_to_bgr(y, uv, pitch, w, h, bgr);
    orc_preprocess_norm_frame_bgr(bgr, w, h, tw, th, norm, out_dtype, out, cstride);
    free(bgr);
}

ORC_API void orc_preprocess_clip_frame_nv12(const uint8_t *y, const uint8_t *uv, int pitch, int w, int h,
                                            int tw, int th, int half, void *out)
{
    uint8_t *bgr = (uint8_t *)malloc((size_t)w * h * 3);
    orc_nv12_to_bgr(y, uv, pitch, w, h, bgr);
    orc_preprocess_clip_frame_bgr(bgr, w, h, tw, th, half, out);
    free(bgr);
}

/* ------------------------------------------------------------------ clip buffering */
/* temporal_detector.py:58-120: feeds frame ids 0..n_frames-1 of one stream; writes the frame
 * index at which each clip fires and the L frame ids forming it.  Returns the clip count. */
ORC_API int orc_clip_schedule(int L, int stride, double overlap, int n_frames, int max_clips,
                              int *fired_at, int *clip_ids)
{
    int step = (int)((double)L * (1.0 - overlap)); /* :66-68 */
    if (step < 1) step = 1;
    int need = L * stride, keep = need - step;
    if (keep < 0) keep = 0;
    int *buf = (int *)malloc(sizeof(int) * (size_t)need);
    int n = 0, clips = 0;
    for (int f = 0; f < n_frames; ++f) {
        if (n == need) { memmove(buf, buf + 1, sizeof(int) * (size_t)(need - 1)); --n; } /* deque(maxlen) */
        buf[n++] = f;
        if (n < need) continue;
        if (clips < max_clips) {
            fired_at[clips] = f;
            for (int i = 0; i < L; ++i) clip_ids[clips * L + i] = buf[i * stride];
        }
        ++clips;
        if (keep > 0) { memmove(buf, buf + (need - keep), sizeof(int) * (size_t)keep); n = keep; }
        else n = 0;
    }
    free(buf);
    return clips;
}

/* ------------------------------------------------------------------ frame gates (UNPINNED: OpenCV absent) */
/* utils/frame_filter.py:26-40 MotionFilter.should_process, one step:
 *   gray = cvtColor(BGR2GRAY): (B*1868 + G*9617 + R*4899 + 8192) >> 14   (OpenCV 8-bit, yuv_shift = 14)
 *   blur = GaussianBlur(gray, (5,5), 0): separable [1,4,6,4,1]/16, BORDER_REFLECT_101, OpenCV's 8-bit
 *          fixed-point path is exact for these dyadic weights: (sum + 128) >> 8
 *   diff = absdiff(blur, prev); count of diff > 25 (THRESH_BINARY)
 * Writes the new blurred frame to `blur_out`; returns the changed-pixel count (-1 when prev == NULL). */
static inline int refl101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }

ORC_API void orc_bgr_to_gray(const uint8_t *bgr, int w, int h, uint8_t *gray)
{
    for (size_t i = 0; i < (size_t)w * h; ++i)
        gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}

ORC_API void orc_gaussian5_u8(const uint8_t *src, int w, int h, uint8_t *dst)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int t = -2; t <= 2; ++t) s += k[t + 2] * src[(size_t)y * w + refl101(x + t, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int t = -2; t <= 2; ++t) s += k[t + 2] * tmp[(size_t)refl101(y + t, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)((s + 128) >> 8);
        }
    free(tmp);
}

ORC_API long orc_motion_step_bgr(const uint8_t *bgr, int w, int h, const uint8_t *prev_blur, uint8_t *blur_out)
{
    uint8_t *gray = (uint8_t *)malloc((size_t)w * h);
    orc_bgr_to_gray(bgr, w, h, gray);
    orc_gaussian5_u8(gray, w, h, blur_out);
    free(gray);
    if (!prev_blur) return -1;
    long cnt = 0;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        int d = (int)blur_out[i] - (int)prev_blur[i];
        if (d < 0) d = -d;
        cnt += d > 25;
    }
    return cnt;
}

ORC_API long orc_motion_step_nv12(const uint8_t *y, const uint8_t *uv, int pitch, int w, int h, const uint8_t *prev_blur,
                                  uint8_t *blur_out)
{
    uint8_t *bgr = (uint8_t *)malloc((size_t)w * h * 3);
    orc_nv12_to_bgr(y, uv, pitch, w, h, bgr);
    long c = orc_motion_step_bgr(bgr, w, h, prev_blur, blur_out);
    free(bgr);
    return c;
}

// rva_yolov8_plan_*: the fused YOLOv8 detector as ONE object behind the C ABI.
//
// Replaces the reference's `session.run` (detector.py:597-609: ONNX Runtime on the exported ultralytics graph) for the fp16 path:
// create() takes the network's convolutions in module order (BatchNorm folded, the checkpoint's own fp32 [Cout][Cin][k][k]
// layout), packs them for the kernels of rva_conv.hip, allocates every NHWC fp16 activation buffer in HBM and lays down the static
// list of launches (one per Conv-BN-SiLU; concat / chunk / upsample never materialise: producers write channel slices of the
// concat buffers, consumers read slices through a row stride); run() replays that list -- one ABI call per forward pass.  The
// per-layer kernel variant is a field of each step: a tuner (engine.py) times the applicable variants through
// rva_yolov8_plan_launch_tunable and fixes its choice with rva_yolov8_plan_set_variant; without a tuner every step uses the
// library's heuristic (variant 0).  No host synchronisation, no allocation after create(): a run can be captured into a hipGraph.
//
// Graph (ultralytics YOLOv8 n / s / m / l / x: widths c1..c5, C2f depths, nc classes, reg_max 16):
//   b0 stem 3x3 s2 | b1 3x3 s2 | b2 C2f | b3 3x3 s2 | b4 C2f | b5 3x3 s2 | b6 C2f | b7 3x3 s2 | b8 C2f | b9 SPPF |
//   h12 C2f(cat[up(p5), p4]) | h15 C2f(cat[up(n4), p3]) | h16 3x3 s2 | h18 C2f(cat[h16, n4]) | h19 3x3 s2 | h21 C2f(cat[h19, p5]) |
//   detect: per level box = 3x3, 3x3, 1x1 (4 reg_max) and cls = 3x3, 3x3, 1x1 (nc); DFL + dist2bbox + sigmoid -> [B, 4+nc, A].
// Module order of `convs` (what create() consumes, checked against the descriptor): b0, b1, C2f(b2), b3, C2f(b4), b5, C2f(b6), b7,
// C2f(b8), SPPF(b9) = cv1, cv2, C2f(h12), C2f(h15), h16, C2f(h18), h19, C2f(h21), then per level l = 0..2: box[l][0..2], then per
// level: cls[l][0..2];  C2f(x) = cv1, cv2, then per bottleneck: cv1, cv2.
#include <cstring>
#include <memory>

#include "rva_internal.h"

namespace {

enum StepKind { K_CONV, K_UPCAT, K_HEAD, K_STEM2, K_STEM, K_SPPF3, K_POOL5, K_UP2, K_HEAD3, K_PAIR32 };

struct View {           // a channel slice of an NHWC buffer
    char *base = nullptr; int ld = 0, off = 0, ch = 0;
    void *ptr() const { return base + 2 * (size_t)off; }
    View sub(int o, int c) const { View v = *this; v.off += o; v.ch = c; return v; }
};

struct Step {
    StepKind kind;
    int lane = 0;                         // 0 = main stream, 1 / 2 = side stream of a detect branch
    int tunable = -1;                     // index into plan->tunables, -1 = fixed kernel
    const void *in = nullptr, *in2 = nullptr; int ldi = 0, ldi2 = 0, c_in = 0, c_in2 = 0;
    const void *w = nullptr, *w2 = nullptr; const float *b = nullptr, *b2 = nullptr;
    void *out = nullptr, *out2 = nullptr, *out3 = nullptr; int ldo = 0;
    const void *res = nullptr; int ldr = 0;
    int H = 0, W = 0, Cin = 0, Cout = 0, k = 0, stride = 0, act = 0, variant = 0;
    int mode = 0, a0 = 0; float stride_px = 0.f;
    // head3
    const void *hb[3] = {nullptr, nullptr, nullptr}, *hk[3] = {nullptr, nullptr, nullptr};
    int32_t hldb[3] = {0, 0, 0}, hldc[3] = {0, 0, 0}, hh[3] = {0, 0, 0}, hw[3] = {0, 0, 0}; float hs[3] = {0, 0, 0};
};

struct Tunable { int step; std::string desc; };

}  // namespace

struct rva_yolov8_plan {
    rva_ctx *ctx = nullptr;
    rva_yolov8_desc d{};
    int B = 0, H = 0, W = 0, nc = 0, A = 0;
    std::vector<void *> allocs;
    std::vector<Step> steps;
    std::vector<Tunable> tunables;
    int fork_step[3] = {-1, -1, -1};      // lane l forks off the main stream in front of this step
    int quiet_step = 0;
    hipEvent_t fork_ev[3] = {nullptr, nullptr, nullptr}, join_ev[3] = {nullptr, nullptr, nullptr};
    bool fused_stem = false, fused_head = false;
};

namespace {

struct Builder {
    rva_yolov8_plan *p;
    const rva_conv_weights *convs;
    int next = 0;                         // next convolution of the module-order list
    int lane = 0;
    std::string err;

    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }

    void *dev_alloc(size_t bytes, bool zero)
    {
        void *ptr = nullptr;
        if (hipMalloc(&ptr, bytes ? bytes : 16) != hipSuccess) { fail("hipMalloc failed"); return nullptr; }
        p->allocs.push_back(ptr);
        if (zero && hipMemset(ptr, 0, bytes) != hipSuccess) { fail("hipMemset failed"); return nullptr; }
        return ptr;
    }
    View buf(long m, int ch)
    {
        View v;
        // + 64 B: a convolution whose Cin is not a multiple of 32 runs with Cin rounded up (zero weights for the extra channels, see
        // conv()) and reads up to 31 channels past its slice -- the next slice of the row, the next row, or, behind the last row, this slack
        v.base = (char *)dev_alloc((size_t)m * ch * 2 + 64, true);
        v.ld = ch; v.off = 0; v.ch = ch;
        return v;
    }
    // Pack sibling convolutions (equal Cin / kernel / stride, outputs concatenated) as [CoutPad][k*k][CinPad] fp16 + [CoutPad] fp32
    bool pack(const rva_conv_weights *const *cv, int n, const void **w_dev, const float **b_dev, int *cin, int *cout, int *k, int *stride)
    {
        const rva_conv_weights &c0 = *cv[0];
        int co = 0;
        for (int i = 0; i < n; ++i) {
            if (!cv[i]->weight || cv[i]->cin != c0.cin || cv[i]->k != c0.k || cv[i]->stride != c0.stride) return fail("sibling convolutions differ");
            co += cv[i]->cout;
        }
        const int kk = c0.k * c0.k, cpad = rva_conv_cout_pad(co), cinp = (c0.cin + 31) / 32 * 32;
        std::vector<_Float16> hw((size_t)cpad * kk * cinp, (_Float16)0.f);
        std::vector<float> hb(cpad, 0.f);
        int o = 0;
        for (int i = 0; i < n; ++i) {
            const rva_conv_weights &c = *cv[i];
            for (int oc = 0; oc < c.cout; ++oc) {
                for (int ic = 0; ic < c.cin; ++ic)
                    for (int t = 0; t < kk; ++t)
                        hw[((size_t)(o + oc) * kk + t) * cinp + ic] = (_Float16)c.weight[((size_t)oc * c.cin + ic) * kk + t];
                if (c.bias) hb[o + oc] = c.bias[oc];
            }
            o += c.cout;
        }
        void *dw = dev_alloc(hw.size() * 2, false), *db = dev_alloc(hb.size() * 4, false);
        if (!dw || !db) return false;
        if (hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("weight upload failed");
        *w_dev = dw; *b_dev = (const float *)db; *cin = c0.cin; *cout = co; *k = c0.k; *stride = c0.stride;
        return true;
    }
    const rva_conv_weights *take(int cin, int cout, int k, int stride, const char *what)
    {
        if (next >= p->d.n_convs) { fail(std::string("convolution list too short at ") + what); return nullptr; }
        const rva_conv_weights *c = &convs[next++];
        if (c->cin != cin || c->cout != cout || c->k != k || c->stride != stride) {
            char m[200];
            snprintf(m, sizeof m, "convolution %d (%s): expected %d->%d k%d s%d, got %d->%d k%d s%d", next - 1, what, cin, cout, k, stride,
                     c->cin, c->cout, c->k, c->stride);
            fail(m);
            return nullptr;
        }
        return c;
    }
    void push(Step &s, const char *desc_fmt, ...)
    {
        s.lane = lane;
        if (desc_fmt) {
            char d[96];
            va_list ap; va_start(ap, desc_fmt); vsnprintf(d, sizeof d, desc_fmt, ap); va_end(ap);
            s.tunable = (int)p->tunables.size();
            p->tunables.push_back(Tunable{(int)p->steps.size(), d});
        }
        p->steps.push_back(s);
    }
    // one Conv-BN-SiLU (or a fused group of siblings); returns false on error
    bool conv(const rva_conv_weights *const *cv, int n, View src, View dst, int h, int w, int act, const View *res = nullptr)
    {
        Step s{}; s.kind = K_CONV;
        if (!pack(cv, n, &s.w, &s.b, &s.Cin, &s.Cout, &s.k, &s.stride)) return false;
        if (s.Cin != src.ch || s.Cout != dst.ch) return fail("convolution does not fit its buffers");
        s.in = src.ptr(); s.ldi = src.ld; s.out = dst.ptr(); s.ldo = dst.ld; s.H = h; s.W = w; s.act = act;
        if (res) { s.res = res->ptr(); s.ldr = res->ld; }
        const int cin_real = s.Cin;
        // Cin 48 / 16 (YOLOv8m / n): the LDS-DMA kernels want whole 32-channel chunks.  The packed weights already carry zero
        // columns up to the next multiple of 32, so the step simply declares the padded Cin: the extra channels it reads are whatever
        // follows the slice (finite activations of the neighbouring slice / pixel, or the buffer's zero slack) times zero.
        if (s.Cin % 32 && !(p->d.flags & RVA_PLAN_NO_CIN_PAD)) s.Cin = (s.Cin + 31) / 32 * 32;
        push(s, "%d->%d k%ds%d %dx%d", cin_real, s.Cout, s.k, s.stride, h, w);
        return true;
    }
    bool conv1(const rva_conv_weights *c, View src, View dst, int h, int w, int act, const View *res = nullptr)
    {
        return c && conv(&c, 1, src, dst, h, w, act, res);
    }
    bool upcat(const rva_conv_weights *c, View low, View skip, View dst, int h, int w)
    {
        if (!c) return false;
        Step s{}; s.kind = K_UPCAT;
        if (!pack(&c, 1, &s.w, &s.b, &s.Cin, &s.Cout, &s.k, &s.stride)) return false;
        if (s.k != 1 || s.stride != 1 || s.Cin != low.ch + skip.ch || s.Cout != dst.ch) return fail("upsample + concat convolution does not fit");
        s.in = low.ptr(); s.ldi = low.ld; s.c_in = low.ch; s.in2 = skip.ptr(); s.ldi2 = skip.ld; s.c_in2 = skip.ch;
        s.out = dst.ptr(); s.ldo = dst.ld; s.H = h; s.W = w; s.act = 1;
        push(s, "up%d+%d->%d k1s1 %dx%d", low.ch, skip.ch, s.Cout, h, w);
        return true;
    }
    bool head(const rva_conv_weights *c, View src, int mode, int h, int w, int a0, float stride_px)
    {
        if (!c) return false;
        Step s{}; s.kind = K_HEAD;
        if (!pack(&c, 1, &s.w, &s.b, &s.Cin, &s.Cout, &s.k, &s.stride)) return false;
        if (s.k != 1 || s.stride != 1 || s.Cin != src.ch) return fail("head convolution does not fit");
        s.in = src.ptr(); s.ldi = src.ld; s.H = h; s.W = w; s.mode = mode; s.a0 = a0; s.stride_px = stride_px;
        push(s, "head%d:%d->%d k1s1 %dx%d", mode, s.Cin, s.Cout, h, w);
        return true;
    }
    // C2f: src is one view, or (low, skip) standing for cat([upsample2x(low), skip])
    bool c2f(int c1, int c2, int n, bool shortcut, const View *src, const View *low, const View *skip, View dst, int h, int w, const char *what)
    {
        const int c = c2 / 2;
        const long m = (long)p->B * h * w;
        const rva_conv_weights *cv1 = take(c1, 2 * c, 1, 1, what), *cv2 = take((2 + n) * c, c2, 1, 1, what);
        if (!cv1 || !cv2) return false;
        View cat = buf(m, (2 + n) * c), tmp = buf(m, c);
        if (!cat.base || !tmp.base) return false;
        if (low) { if (!upcat(cv1, *low, *skip, cat.sub(0, 2 * c), h, w)) return false; }
        else if (!conv1(cv1, *src, cat.sub(0, 2 * c), h, w, 1)) return false;
        for (int i = 0; i < n; ++i) {
            const rva_conv_weights *b1 = take(c, c, 3, 1, what), *b2 = take(c, c, 3, 1, what);
            View x = cat.sub((1 + i) * c, c);
            if (c == 32 && shortcut && !(p->d.flags & RVA_PLAN_NO_PAIR32)) {
                // both 3x3 convolutions and the shortcut in one launch, the intermediate in LDS (rva_c2f_pair32_f16)
                Step s{}; s.kind = K_PAIR32;
                int ci, co, kk, st;
                if (!pack(&b1, 1, &s.w, &s.b, &ci, &co, &kk, &st) || !pack(&b2, 1, &s.w2, &s.b2, &ci, &co, &kk, &st)) return false;
                View y = cat.sub((2 + i) * c, c);
                s.in = x.ptr(); s.ldi = x.ld; s.out = y.ptr(); s.ldo = y.ld; s.H = h; s.W = w; s.Cin = s.Cout = c; s.k = 3; s.stride = 1; s.act = 1;
                push(s, nullptr);
                continue;
            }
            if (!conv1(b1, x, tmp, h, w, 1)) return false;
            if (!conv1(b2, tmp, cat.sub((2 + i) * c, c), h, w, 1, shortcut ? &x : nullptr)) return false;
        }
        return conv1(cv2, cat, dst, h, w, 1);
    }

    bool build()
    {
        const rva_yolov8_desc &d = p->d;
        const int B = d.batch, H = d.height, W = d.width;
        const int c1 = d.widths[0], c2 = d.widths[1], c3 = d.widths[2], c4 = d.widths[3], c5 = d.widths[4];
        const int h1 = H / 2, w1 = W / 2, h2 = H / 4, w2 = W / 4, h3 = H / 8, w3 = W / 8, h4 = H / 16, w4 = W / 16, h5 = H / 32, w5 = W / 32;
        const int nh = d.depth_head;
        // stem (planar input from K1): weights [64][32] fp16, column order k = 2 j + kx (kx in {0, 1}), k = 18 + j (kx = 2), j = c*3 + ky
        const rva_conv_weights *b0 = take(3, c1, 3, 2, "b0"), *b1 = take(c1, c2, 3, 2, "b1");
        if (!b0 || !b1) return false;
        if (c1 > 64) return fail("stem wider than 64 channels");
        std::vector<_Float16> sw(64 * 32, (_Float16)0.f);
        std::vector<float> sb(64, 0.f);
        for (int co = 0; co < c1; ++co) {
            for (int j = 0; j < 9; ++j)                       // j = c*3 + ky;  weight[co][c][ky][kx]
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = b0->weight[((size_t)co * 3 + j / 3) * 9 + (j % 3) * 3 + kx];
                    sw[co * 32 + (kx < 2 ? 2 * j + kx : 18 + j)] = (_Float16)v;
                }
            if (b0->bias) sb[co] = b0->bias[co];
        }
        void *dsw = dev_alloc(sw.size() * 2, false), *dsb = dev_alloc(sb.size() * 4, false);
        if (!dsw || !dsb) return false;
        if (hipMemcpy(dsw, sw.data(), sw.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dsb, sb.data(), sb.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("stem upload failed");
        View x1 = buf((long)B * h2 * w2, c2);
        if (!x1.base) return false;
        p->fused_stem = c1 == 32 && c2 == 64 && !(d.flags & RVA_PLAN_NO_STEM2);
        if (p->fused_stem) {
            Step s{}; s.kind = K_STEM2; s.w = dsw; s.b = (const float *)dsb; s.out = x1.ptr(); s.ldo = x1.ld; s.H = H; s.W = W;
            int ci, co, k, st;
            if (!pack(&b1, 1, &s.w2, &s.b2, &ci, &co, &k, &st)) return false;
            push(s, nullptr);
        } else {
            View x0 = buf((long)B * h1 * w1, c1);
            if (!x0.base) return false;
            Step s{}; s.kind = K_STEM; s.w = dsw; s.b = (const float *)dsb; s.out = x0.ptr(); s.ldo = x0.ld; s.H = H; s.W = W; s.Cout = c1;
            push(s, nullptr);
            if (!conv1(b1, x0, x1, h1, w1, 1)) return false;
        }
        View x2 = buf((long)B * h2 * w2, c2);
        if (!x2.base || !c2f(c2, c2, d.depth_backbone[0], true, &x1, nullptr, nullptr, x2, h2, w2, "b2")) return false;
        // concat buffers of the neck: producers write their slice directly
        View cat15 = buf((long)B * h3 * w3, c4 + c3), cat12 = buf((long)B * h4 * w4, c5 + c4), cat18 = buf((long)B * h4 * w4, c3 + c4),
             cat21 = buf((long)B * h5 * w5, c4 + c5);
        if (!cat15.base || !cat12.base || !cat18.base || !cat21.base) return false;
        View p3 = cat15.sub(c4, c3), p4 = cat12.sub(c5, c4), n4 = cat18.sub(c3, c4), p5 = cat21.sub(c4, c5);
        View t3 = buf((long)B * h3 * w3, c3);
        if (!t3.base || !conv1(take(c2, c3, 3, 2, "b3"), x2, t3, h2, w2, 1)) return false;
        if (!c2f(c3, c3, d.depth_backbone[1], true, &t3, nullptr, nullptr, p3, h3, w3, "b4")) return false;
        View t4 = buf((long)B * h4 * w4, c4);
        if (!t4.base || !conv1(take(c3, c4, 3, 2, "b5"), p3, t4, h3, w3, 1)) return false;
        if (!c2f(c4, c4, d.depth_backbone[2], true, &t4, nullptr, nullptr, p4, h4, w4, "b6")) return false;
        View t5 = buf((long)B * h5 * w5, c5);
        p->quiet_step = (int)p->steps.size();
        if (!t5.base || !conv1(take(c4, c5, 3, 2, "b7"), p4, t5, h4, w4, 1)) return false;
        View t5b = buf((long)B * h5 * w5, c5);
        if (!t5b.base || !c2f(c5, c5, d.depth_backbone[3], true, &t5, nullptr, nullptr, t5b, h5, w5, "b8")) return false;
        {   // SPPF
            const int c_ = c5 / 2;
            const rva_conv_weights *cv1 = take(c5, c_, 1, 1, "b9.cv1"), *cv2 = take(4 * c_, c5, 1, 1, "b9.cv2");
            View cat = buf((long)B * h5 * w5, 4 * c_);
            if (!cat.base || !conv1(cv1, t5b, cat.sub(0, c_), h5, w5, 1)) return false;
            if (h5 * w5 <= 2400) {
                Step s{}; s.kind = K_SPPF3; s.in = cat.sub(0, c_).ptr(); s.ldi = cat.ld; s.out = cat.sub(c_, c_).ptr(); s.out2 = cat.sub(2 * c_, c_).ptr();
                s.out3 = cat.sub(3 * c_, c_).ptr(); s.ldo = cat.ld; s.H = h5; s.W = w5; s.Cin = c_;
                push(s, nullptr);
            } else {
                for (int i = 0; i < 3; ++i) {
                    Step s{}; s.kind = K_POOL5; s.in = cat.sub(i * c_, c_).ptr(); s.ldi = cat.ld; s.out = cat.sub((i + 1) * c_, c_).ptr(); s.ldo = cat.ld;
                    s.H = h5; s.W = w5; s.Cin = c_;
                    push(s, nullptr);
                }
            }
            if (!conv1(cv2, cat, p5, h5, w5, 1)) return false;
        }
        const bool fuse_up = c5 % 64 == 0 && c4 % 64 == 0 && c3 % 64 == 0 && h4 == 2 * h5 && w4 == 2 * w5 && h3 == 2 * h4 && w3 == 2 * w4;
        View n3 = buf((long)B * h3 * w3, c3);
        if (!n3.base) return false;
        if (fuse_up) {    // FPN top-down path: upsample + concat folded into the consuming 1x1 convolutions
            if (!c2f(c5 + c4, c4, nh, false, nullptr, &p5, &p4, n4, h4, w4, "h12")) return false;
            if (!c2f(c4 + c3, c3, nh, false, nullptr, &n4, &p3, n3, h3, w3, "h15")) return false;
        } else {
            Step u1{}; u1.kind = K_UP2; u1.in = p5.ptr(); u1.ldi = p5.ld; u1.out = cat12.sub(0, c5).ptr(); u1.ldo = cat12.ld; u1.H = h5; u1.W = w5; u1.Cin = c5;
            push(u1, nullptr);
            if (!c2f(c5 + c4, c4, nh, false, &cat12, nullptr, nullptr, n4, h4, w4, "h12")) return false;
            Step u2{}; u2.kind = K_UP2; u2.in = n4.ptr(); u2.ldi = n4.ld; u2.out = cat15.sub(0, c4).ptr(); u2.ldo = cat15.ld; u2.H = h4; u2.W = w4; u2.Cin = c4;
            push(u2, nullptr);
            if (!c2f(c4 + c3, c3, nh, false, &cat15, nullptr, nullptr, n3, h3, w3, "h15")) return false;
        }
        const int fork_n3 = (int)p->steps.size();            // n3 is complete here: the stride-8 detect branch may start
        if (!conv1(take(c3, c3, 3, 2, "h16"), n3, cat18.sub(0, c3), h3, w3, 1)) return false;
        View m4 = buf((long)B * h4 * w4, c4);
        if (!m4.base || !c2f(c3 + c4, c4, nh, false, &cat18, nullptr, nullptr, m4, h4, w4, "h18")) return false;
        const int fork_m4 = (int)p->steps.size();            // m4 is complete: the stride-16 detect branch may start
        if (!conv1(take(c4, c4, 3, 2, "h19"), m4, cat21.sub(0, c4), h4, w4, 1)) return false;
        View m5 = buf((long)B * h5 * w5, c5);
        if (!m5.base || !c2f(c4 + c5, c5, nh, false, &cat21, nullptr, nullptr, m5, h5, w5, "h21")) return false;
        // detect head
        const int rm4 = 4 * d.reg_max;
        int cb = c3 / 4 > rm4 ? c3 / 4 : rm4;
        if (cb < 16) cb = 16;
        int cc = c3 > (d.nc < 100 ? d.nc : 100) ? c3 : (d.nc < 100 ? d.nc : 100);
        p->A = h3 * w3 + h4 * w4 + h5 * w5;
        // module order: all box branches first, then all class branches
        const rva_conv_weights *box[3][3], *cls[3][3];
        const int chs[3] = {c3, c4, c5};
        for (int l = 0; l < 3; ++l) {
            box[l][0] = take(chs[l], cb, 3, 1, "detect.box.0"); box[l][1] = take(cb, cb, 3, 1, "detect.box.1"); box[l][2] = take(cb, rm4, 1, 1, "detect.box.2");
            if (!box[l][0] || !box[l][1] || !box[l][2]) return false;
        }
        for (int l = 0; l < 3; ++l) {
            cls[l][0] = take(chs[l], cc, 3, 1, "detect.cls.0"); cls[l][1] = take(cc, cc, 3, 1, "detect.cls.1"); cls[l][2] = take(cc, d.nc, 1, 1, "detect.cls.2");
            if (!cls[l][0] || !cls[l][1] || !cls[l][2]) return false;
        }
        if (next != d.n_convs) return fail("convolution list longer than the architecture");
        p->fused_head = cb % 64 == 0 && cc % 64 == 0 && rm4 == 64 && d.nc % 8 == 0;
        const View feats[3] = {n3, m4, m5};
        const int hs[3] = {h3, h4, h5}, ws[3] = {w3, w4, w5};
        const float strides[3] = {8.f, 16.f, 32.f};
        Step h3s{}; h3s.kind = K_HEAD3;
        int a0 = 0;
        for (int l = 0; l < 3; ++l) {
            // the three detect branches only depend on their own feature map and write disjoint anchor ranges: the stride-8 and
            // stride-16 branches may run on side streams beside the rest of the neck (rva_yolov8_plan_run_lanes)
            lane = (p->fused_head && l < 2) ? l + 1 : 0;
            if (lane) p->fork_step[lane] = l == 0 ? fork_n3 : fork_m4;
            const long m = (long)B * hs[l] * ws[l];
            View first = buf(m, cb + cc), b2 = buf(m, cb), k2 = buf(m, cc);
            if (!first.base || !b2.base || !k2.base) return false;
            const rva_conv_weights *sib[2] = {box[l][0], cls[l][0]};
            if (!conv(sib, 2, feats[l], first, hs[l], ws[l], 1)) return false;      // one launch, Cout = cb + cc
            if (p->fused_head) {
                if (!conv1(box[l][1], first.sub(0, cb), b2, hs[l], ws[l], 1) || !head(box[l][2], b2, 1, hs[l], ws[l], a0, strides[l])) return false;
                if (!conv1(cls[l][1], first.sub(cb, cc), k2, hs[l], ws[l], 1) || !head(cls[l][2], k2, 2, hs[l], ws[l], a0, strides[l])) return false;
            } else {
                View bo = buf(m, rm4), ko = buf(m, d.nc);
                if (!bo.base || !ko.base) return false;
                if (!conv1(box[l][1], first.sub(0, cb), b2, hs[l], ws[l], 1) || !conv1(box[l][2], b2, bo, hs[l], ws[l], 0)) return false;
                if (!conv1(cls[l][1], first.sub(cb, cc), k2, hs[l], ws[l], 1) || !conv1(cls[l][2], k2, ko, hs[l], ws[l], 0)) return false;
                h3s.hb[l] = bo.ptr(); h3s.hldb[l] = bo.ld; h3s.hk[l] = ko.ptr(); h3s.hldc[l] = ko.ld; h3s.hh[l] = hs[l]; h3s.hw[l] = ws[l]; h3s.hs[l] = strides[l];
            }
            a0 += hs[l] * ws[l];
        }
        lane = 0;
        if (!p->fused_head) push(h3s, nullptr);
        return err.empty();
    }
};

int launch_step(rva_yolov8_plan *p, const Step &s, int variant, const void *input, void *output, rva_stream_t st)
{
    rva_ctx *c = p->ctx;
    switch (s.kind) {
    case K_CONV: return rva_conv2d_nhwc_f16_v(c, s.in, s.ldi, s.w, s.b, s.out, s.ldo, s.res, s.ldr, p->B, s.H, s.W, s.Cin, s.Cout, s.k, s.stride, s.act, variant, st);
    case K_UPCAT: return rva_conv1x1_upcat_f16(c, s.in, s.ldi, s.c_in, s.in2, s.ldi2, s.c_in2, s.w, s.b, s.out, s.ldo, p->B, s.H, s.W, s.Cout, s.act, variant, st);
    case K_HEAD: return rva_conv1x1_head_f16(c, s.in, s.ldi, s.w, s.b, p->B, s.H, s.W, s.Cin, s.Cout, s.mode, output, p->nc, p->A, s.a0, s.stride_px, variant, st);
    case K_PAIR32: return rva_c2f_pair32_f16(c, s.in, s.ldi, s.w, s.b, s.w2, s.b2, s.out, s.ldo, p->B, s.H, s.W, st);
    case K_STEM2: return rva_stem2_f16(c, input, s.w, s.b, s.w2, s.b2, s.out, s.ldo, p->B, s.H, s.W, st);
    case K_STEM: return rva_stem_conv_f16(c, input, s.w, s.b, s.out, s.ldo, p->B, s.H, s.W, s.Cout, st);
    case K_SPPF3: return rva_sppf_pool3_nhwc_f16(c, s.in, s.ldi, s.out, s.out2, s.out3, s.ldo, p->B, s.H, s.W, s.Cin, st);
    case K_POOL5: return rva_maxpool5_nhwc_f16(c, s.in, s.ldi, s.out, s.ldo, p->B, s.H, s.W, s.Cin, st);
    case K_UP2: return rva_upsample2x_nhwc_f16(c, s.in, s.ldi, s.out, s.ldo, p->B, s.H, s.W, s.Cin, st);
    case K_HEAD3: return rva_yolo_head3_f16(c, s.hb, s.hldb, s.hk, s.hldc, output, p->B, s.hh, s.hw, p->nc, p->A, s.hs, st);
    }
    return RVA_ERR_ARG;
}

bool variant_fits(const Step &s, int variant)
{
    if (s.kind == K_CONV) return variant >= 0;
    return variant == 0 || (variant >= 33 && variant <= 39);     // upcat / head: the LDS-DMA gather family
}

}  // namespace

extern "C" {

int rva_yolov8_plan_create(rva_ctx *ctx, const rva_yolov8_desc *desc, const rva_conv_weights *convs, rva_yolov8_plan **out)
{
    if (!ctx || !desc || !convs || !out) return RVA_ERR_ARG;
    *out = nullptr;
    if (desc->batch <= 0 || desc->height <= 0 || desc->width <= 0 || desc->height % 32 || desc->width % 32 || desc->nc <= 0 ||
        desc->nc % 8 || desc->reg_max != 16 || desc->n_convs <= 0)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_yolov8_plan_create: batch > 0, height and width multiples of 32, nc %% 8 == 0, reg_max 16");
    for (int i = 0; i < 5; ++i)
        if (desc->widths[i] <= 0 || desc->widths[i] % 8) return rva_fail(ctx, RVA_ERR_ARG, "rva_yolov8_plan_create: channel widths must be multiples of 8");
    RVA_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<rva_yolov8_plan> p(new rva_yolov8_plan());
    p->ctx = ctx; p->d = *desc; p->B = desc->batch; p->H = desc->height; p->W = desc->width; p->nc = desc->nc;
    Builder b{p.get(), convs};
    const bool ok = b.build();
    if (!ok) {
        for (void *a : p->allocs) (void)hipFree(a);
        return rva_fail(ctx, b.err.find("hip") == 0 || b.err.find("upload") != std::string::npos ? RVA_ERR_HIP : RVA_ERR_ARG,
                        "rva_yolov8_plan_create: %s", b.err.c_str());
    }
    for (int l = 1; l <= 2; ++l)
        if (p->fork_step[l] >= 0) {
            RVA_HIP(ctx, hipEventCreateWithFlags(&p->fork_ev[l], hipEventDisableTiming));
            RVA_HIP(ctx, hipEventCreateWithFlags(&p->join_ev[l], hipEventDisableTiming));
        }
    RVA_HIP(ctx, hipDeviceSynchronize());                  // weights are in place before the first run on any stream
    *out = p.release();
    return RVA_OK;
}

void rva_yolov8_plan_destroy(rva_yolov8_plan *p)
{
    if (!p) return;
    for (void *a : p->allocs) (void)hipFree(a);
    for (int l = 0; l < 3; ++l) {
        if (p->fork_ev[l]) (void)hipEventDestroy(p->fork_ev[l]);
        if (p->join_ev[l]) (void)hipEventDestroy(p->join_ev[l]);
    }
    delete p;
}

int rva_yolov8_plan_info(const rva_yolov8_plan *p, int32_t *anchors, int32_t *out_rows, int32_t *n_steps, int32_t *n_tunable, int32_t *quiet_step)
{
    if (!p) return RVA_ERR_ARG;
    if (anchors) *anchors = p->A;
    if (out_rows) *out_rows = 4 + p->nc;
    if (n_steps) *n_steps = (int32_t)p->steps.size();
    if (n_tunable) *n_tunable = (int32_t)p->tunables.size();
    if (quiet_step) *quiet_step = p->quiet_step;
    return RVA_OK;
}

int rva_yolov8_plan_tunable_desc(const rva_yolov8_plan *p, int index, char *buf, int len)
{
    if (!p || index < 0 || index >= (int)p->tunables.size() || !buf || len <= 0) return RVA_ERR_ARG;
    snprintf(buf, (size_t)len, "%s", p->tunables[index].desc.c_str());
    return RVA_OK;
}

int rva_yolov8_plan_set_variant(rva_yolov8_plan *p, int index, int variant)
{
    if (!p || index < 0 || index >= (int)p->tunables.size()) return RVA_ERR_ARG;
    Step &s = p->steps[p->tunables[index].step];
    if (!variant_fits(s, variant) || variant > rva_conv_num_variants()) return rva_fail(p->ctx, RVA_ERR_ARG, "rva_yolov8_plan_set_variant: variant %d does not exist for this layer", variant);
    s.variant = variant;
    return RVA_OK;
}

int rva_yolov8_plan_get_variant(const rva_yolov8_plan *p, int index)
{
    if (!p || index < 0 || index >= (int)p->tunables.size()) return -1;
    return p->steps[p->tunables[index].step].variant;
}

int rva_yolov8_plan_launch_tunable(rva_yolov8_plan *p, int index, int variant, void *output, rva_stream_t stream)
{
    if (!p || index < 0 || index >= (int)p->tunables.size()) return RVA_ERR_ARG;
    const Step &s = p->steps[p->tunables[index].step];
    if (!variant_fits(s, variant)) return RVA_ERR_ARG;
    if (s.kind == K_HEAD && !output) return RVA_ERR_ARG;
    return launch_step(p, s, variant, nullptr, output, stream);
}

int rva_yolov8_plan_run_range(rva_yolov8_plan *p, const void *input, void *output, int first, int last, rva_stream_t stream)
{
    if (!p || !input || !output || first < 0 || last > (int)p->steps.size() || first > last) return RVA_ERR_ARG;
    if (((uintptr_t)input | (uintptr_t)output) % 16) return rva_fail(p->ctx, RVA_ERR_ARG, "rva_yolov8_plan_run: input and output must be 16-byte aligned");
    for (int i = first; i < last; ++i) {
        const Step &s = p->steps[i];
        const int rc = launch_step(p, s, s.variant, input, output, stream);
        if (rc != RVA_OK) return rc;
    }
    return RVA_OK;
}

int rva_yolov8_plan_run(rva_yolov8_plan *p, const void *input, void *output, rva_stream_t stream)
{
    return p ? rva_yolov8_plan_run_range(p, input, output, 0, (int)p->steps.size(), stream) : RVA_ERR_ARG;
}

int rva_yolov8_plan_run_lanes(rva_yolov8_plan *p, const void *input, void *output, rva_stream_t stream, rva_stream_t side1, rva_stream_t side2)
{
    if (!p || !input || !output) return RVA_ERR_ARG;
    if (!side1 || !side2 || p->fork_step[1] < 0) return rva_yolov8_plan_run(p, input, output, stream);
    if (((uintptr_t)input | (uintptr_t)output) % 16) return rva_fail(p->ctx, RVA_ERR_ARG, "rva_yolov8_plan_run: input and output must be 16-byte aligned");
    hipStream_t lanes[3] = {(hipStream_t)stream, (hipStream_t)side1, (hipStream_t)side2};
    bool started[3] = {true, false, false};
    for (int i = 0; i < (int)p->steps.size(); ++i) {
        for (int l = 1; l <= 2; ++l)
            if (p->fork_step[l] == i) RVA_HIP(p->ctx, hipEventRecord(p->fork_ev[l], lanes[0]));      // everything the main lane has been given so far
        const Step &s = p->steps[i];
        if (s.lane && !started[s.lane]) {
            RVA_HIP(p->ctx, hipStreamWaitEvent(lanes[s.lane], p->fork_ev[s.lane], 0));
            started[s.lane] = true;
        }
        const int rc = launch_step(p, s, s.variant, input, output, (rva_stream_t)lanes[s.lane]);
        if (rc != RVA_OK) return rc;
    }
    for (int l = 1; l <= 2; ++l)
        if (started[l]) {
            RVA_HIP(p->ctx, hipEventRecord(p->join_ev[l], lanes[l]));
            RVA_HIP(p->ctx, hipStreamWaitEvent(lanes[0], p->join_ev[l], 0));
        }
    return RVA_OK;
}

}  // extern "C"

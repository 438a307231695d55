// K2 (decode / score / threshold / compact) and K3 (sort + greedy NMS) for a batch of YOLO heads.
//
// Replaces _TensorRTBaseDetector._postprocess and helpers (detector.py:266-375, 469-481).
// float32 arithmetic in the reference's operation order; the library is built with
// -ffp-contract=off so no multiply is fused into a following add (that would flip borderline
// `iou <= thr` decisions).  Thresholds are rounded to float32 first, as numpy's NEP-50 does.
//
// Data flow per image b (no host round trip, count stays on the device):
//   K2: one thread per anchor, coalesced along the anchor axis of [C, A] heads.  Anchors that pass
//       write box/score/class at their own anchor index (sparse arrays), set their bit in a pass
//       bitmap (anchor order) and append their anchor to an UNORDERED list (one atomic per wave).
//   K3: one workgroup per image.  64-bit keys (~score, anchor) are bitonic-sorted in LDS, which makes
//       the result independent of the append order; greedy NMS runs chunk-wise: wave 0 resolves 64
//       sorted boxes among themselves with ballots/shuffles, then every thread clears later boxes
//       against the chunk's survivors.  Same keep set as the reference's sequential loop because
//       greedy NMS is order-determined.  `keep` indices are recovered from the bitmap by popcount.
//   K3 under load (more than K3_SMALL = 128 and at most 4096 sorted candidates in an image): the chunk loop above is
//       serial per image (~10 us per 64 candidates).  Such an image leaves k3_nms after the sort; k3_mask computes its
//       whole suppression matrix on all CUs -- one wave per 64 x 64 tile of (higher-priority box i, box j > i), the same
//       float32 IoU and the same `!(iou <= thr)` test, one ballot per row -> bit j of word j/64 of row i -- and
//       k3_reduce replays the greedy loop over that matrix with one wave per image: a 64-bit scalar walk inside a
//       chunk (next alive box, clear what its diagonal word suppresses), then the kept rows are OR-ed into the
//       per-chunk "removed" words the lanes hold.  Order-determined like the loop it replaces: identical keep set.
#include <hip/hip_fp16.h>

#include "rva_internal.h"

namespace {

struct PostMeta {
    float left, top, scale, xmax, ymax;
};

struct K2Args {
    const void *raw;
    int A, C;
    long sb, sa, sc;  // element strides: image, anchor, channel
    float thr;
    int use_cls;
    uint32_t cls_mask[32];
    PostMeta meta[RVA_MAX_BATCH];
    float4 *sp_box;
    float *sp_score;
    int32_t *sp_cls;
    int32_t *list;
    int32_t *count;
    uint32_t *bits;
    int nwords;
};

__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const __half *p) { return __half2float(*p); }

template <typename T>
__global__ void __launch_bounds__(256) k2_decode(K2Args a)
{
    const int b = blockIdx.y;
    const int anchor = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool pass = false;
    if (anchor < a.A) {
        const T *p = (const T *)a.raw + (long)b * a.sb + (long)anchor * a.sa;
        float best;
        int bi = 0;
        if (a.C > 5) {  // detector.py:294-305 (both branches): scores = pred[:,5:] * pred[:,4:5]
            const float obj = ldf(p + 4 * a.sc);
            best = ldf(p + 5 * a.sc) * obj;
#pragma unroll 8
            for (int c = 6; c < a.C; ++c) {
                float s = ldf(p + (long)c * a.sc) * obj;
                if (s > best) { best = s; bi = c - 5; }  // argmax: first maximum wins
            }
        } else {  // :306-307
            best = ldf(p + 4 * a.sc);
        }
        pass = best >= a.thr;  // :312
        if (a.use_cls) pass = pass && bi < 1024 && ((a.cls_mask[bi >> 5] >> (bi & 31)) & 1u);  // :313-314
        if (pass) {
            const PostMeta m = a.meta[b];
            const float cx = ldf(p), cy = ldf(p + a.sc), w = ldf(p + 2 * a.sc), h = ldf(p + 3 * a.sc);
            float x1 = cx - w / 2.0f, y1 = cy - h / 2.0f, x2 = cx + w / 2.0f, y2 = cy + h / 2.0f;  // :352-359
            x1 -= m.left; x2 -= m.left; y1 -= m.top; y2 -= m.top;                                // :345-346
            x1 = __fdiv_rn(x1, m.scale); y1 = __fdiv_rn(y1, m.scale);                           // :347
            x2 = __fdiv_rn(x2, m.scale); y2 = __fdiv_rn(y2, m.scale);
            x1 = fminf(fmaxf(x1, 0.0f), m.xmax); x2 = fminf(fmaxf(x2, 0.0f), m.xmax);           // :348-349
            y1 = fminf(fmaxf(y1, 0.0f), m.ymax); y2 = fminf(fmaxf(y2, 0.0f), m.ymax);
            const long o = (long)b * a.A + anchor;
            a.sp_box[o] = make_float4(x1, y1, x2, y2);
            a.sp_score[o] = best;
            a.sp_cls[o] = bi;
        }
    }
    const unsigned long long mask = __ballot(pass);
    // pass bitmap: this wave owns anchors [a0, a0+64) = words a0/32 and a0/32+1 (written whole)
    const int a0 = blockIdx.x * 256 + (threadIdx.x & ~63);
    if (lane < 2) {
        const int w = (a0 >> 5) + lane;
        if (w < a.nwords) a.bits[(long)b * a.nwords + w] = (uint32_t)(mask >> (32 * lane));
    }
    if (mask) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&a.count[b], __popcll(mask));
        base = __shfl(base, 0);
        if (pass) a.list[(long)b * a.A + base + __popcll(mask & ((1ull << lane) - 1ull))] = anchor;
    }
}

// detector.py:469-481, float32; a = the kept (higher-priority) box
__device__ __forceinline__ float iou32(const float4 a, const float4 b)
{
    const float x1 = fmaxf(a.x, b.x), y1 = fmaxf(a.y, b.y);
    const float x2 = fminf(a.z, b.z), y2 = fminf(a.w, b.w);
    float w = x2 - x1, h = y2 - y1;
    w = w > 0.0f ? w : 0.0f;
    h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    float uni = area_a + area_b - inter;
    uni = fmaxf(uni, 1e-6f);
    return __fdiv_rn(inter, uni);
}

struct K3Args {
    const float4 *sp_box;
    const float *sp_score;
    const int32_t *sp_cls;
    const int32_t *list;
    const int32_t *count;
    const uint32_t *bits;
    int nwords, A, kcap;  // kcap: power of two, LDS key capacity
    float iou_thr;
    int max_det;
    float4 *out_boxes;
    float *out_scores;
    int32_t *out_cls, *out_anchor, *out_cand, *out_counts, *out_ncand;
    int32_t *flags;
    // hand-off to the suppression-bitmask path (null m_mask: every image finishes here)
    int km;
    int32_t *m_state, *m_anchor, *m_wprefix;
    float4 *m_box;
    unsigned long long *m_mask;
};

constexpr int K3_THREADS = 512;
constexpr int K3_SMALL = 128;      // up to two 64-box chunks an image finishes inside k3_nms (cheaper than two more launches' work)

extern __shared__ __attribute__((aligned(16))) unsigned char k3_smem[];

__global__ void __launch_bounds__(K3_THREADS) k3_nms(K3Args a)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long *keys = (unsigned long long *)k3_smem;           // [kcap]
    float4 *kept_box = (float4 *)(k3_smem + (size_t)a.kcap * 8);        // [64]
    int *s_ctl = (int *)(k3_smem + (size_t)a.kcap * 8 + 64 * 16);       // [0]=kept in chunk, [1]=out count
    unsigned char *removed = k3_smem + (size_t)a.kcap * 8 + 64 * 16 + 16;  // [kcap]
    int *wprefix = (int *)(k3_smem + (((size_t)a.kcap * 9 + 64 * 16 + 16 + 15) & ~(size_t)15));  // [nwords] passing anchors before word w
    int *wave_tot = wprefix + a.nwords;                                  // [K3_THREADS/64] (all LDS is dynamic: keeps the base 16-byte aligned)

    int K = a.count[b];
    if (a.out_ncand && tid == 0) a.out_ncand[b] = K;
    if (K > a.kcap) {
        if (tid == 0) atomicOr(a.flags, 2);
        K = a.kcap;
    }
    if (K == 0) {
        if (tid == 0) { a.out_counts[b] = 0; if (a.m_state) a.m_state[b] = 0; }
        return;
    }
    int Kpad = 64;
    while (Kpad < K) Kpad <<= 1;
    const int32_t *list = a.list + (long)b * a.A;
    const float *score = a.sp_score + (long)b * a.A;
    const float4 *box = a.sp_box + (long)b * a.A;
    for (int i = tid; i < Kpad; i += K3_THREADS) {
        unsigned long long k = ~0ull;
        if (i < K) {
            const int an = list[i];
            uint32_t u = __float_as_uint(score[an]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
            k = ((unsigned long long)(~u) << 32) | (uint32_t)an;  // descending score, ascending anchor
        }
        keys[i] = k;
        removed[i] = 0;
    }
    if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0; }
    __syncthreads();
    // prefix popcounts of the pass bitmap (for the `keep` index of each survivor), all threads cooperate
    if (a.out_cand) {
        const uint32_t *bw = a.bits + (long)b * a.nwords;
        for (int w0 = 0; w0 < a.nwords; w0 += K3_THREADS) {   // block-wide exclusive scan, 512 words per round
            const int w = w0 + tid;
            const int c = w < a.nwords ? __popc(bw[w]) : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            int base = w0 ? wprefix[w0 - 1] + s_ctl[2] : 0;   // carry of the previous round
            for (int q = 0; q < wave; ++q) base += wave_tot[q];
            if (w < a.nwords) wprefix[w] = base + incl - c;
            __syncthreads();
            if (tid == K3_THREADS - 1) s_ctl[2] = c;           // popcount of the round's last word (prefix is exclusive)
            __syncthreads();
        }
    }
    if (Kpad <= K3_THREADS) {
        // one key per thread: partner exchange by wave shuffle while the stride stays inside a wave (j < 64),
        // through LDS only for the few wider strides -> a handful of block barriers instead of one per stage
        unsigned long long key = tid < Kpad ? keys[tid] : ~0ull;
        for (int k = 2; k <= Kpad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                unsigned long long other;
                if (j >= 64) {
                    __syncthreads();
                    if (tid < Kpad) keys[tid] = key;
                    __syncthreads();
                    other = tid < Kpad ? keys[tid ^ j] : ~0ull;
                } else {
                    const unsigned lo = __shfl_xor((unsigned)key, j), hi = __shfl_xor((unsigned)(key >> 32), j);
                    other = ((unsigned long long)hi << 32) | lo;
                }
                const bool up = (tid & k) == 0, lower = (tid & j) == 0;
                const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
                key = (lower == up) ? mn : mx;
            }
        }
        __syncthreads();
        if (tid < Kpad) keys[tid] = key;
        __syncthreads();
    } else {
        for (int k = 2; k <= Kpad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < Kpad; i += K3_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long x = keys[i], y = keys[ixj];
                        const bool up = (i & k) == 0;
                        if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        }
    }
    // busy image: hand the sorted candidates to k3_mask / k3_reduce (all CUs build the suppression matrix)
    const bool hand = a.m_mask != nullptr && K > K3_SMALL && K <= a.km;
    if (a.m_state && tid == 0) a.m_state[b] = hand ? K : 0;
    if (hand) {
        for (int i = tid; i < K; i += K3_THREADS) {
            const int an = (int)(uint32_t)keys[i];
            a.m_anchor[(size_t)b * a.km + i] = an;
            a.m_box[(size_t)b * a.km + i] = box[an];
        }
        if (a.out_cand)
            for (int w = tid; w < a.nwords; w += K3_THREADS) a.m_wprefix[(size_t)b * a.nwords + w] = wprefix[w];
        return;
    }
    const int nchunks = (K + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int base = c << 6;
        if (wave == 0) {
            const int i = base + lane;
            const bool valid = i < K;
            const int an = valid ? (int)(uint32_t)keys[i] : 0;
            float4 bx = valid ? box[an] : make_float4(0.f, 0.f, 0.f, 0.f);
            bool alive = valid && !removed[i];
            unsigned long long am = __ballot(alive);
            for (int s = 0; s < 64; ++s) {
                if (!((am >> s) & 1ull)) continue;  // wave-uniform
                float4 kb;
                kb.x = __shfl(bx.x, s); kb.y = __shfl(bx.y, s); kb.z = __shfl(bx.z, s); kb.w = __shfl(bx.w, s);
                if (lane > s && alive && !(iou32(kb, bx) <= a.iou_thr)) alive = false;  // detector.py:373
                am = __ballot(alive);
            }
            const int rank = __popcll(am & ((1ull << lane) - 1ull));
            const int nk = __popcll(am);
            const int out0 = s_ctl[1];
            if (alive) {
                kept_box[rank] = bx;
                const int pos = out0 + rank;
                if (pos < a.max_det) {
                    const long o = (long)b * a.max_det + pos;
                    a.out_boxes[o] = bx;
                    a.out_scores[o] = score[an];
                    a.out_cls[o] = a.sp_cls[(long)b * a.A + an];
                    if (a.out_anchor) a.out_anchor[o] = an;
                    if (a.out_cand) {  // index among the thresholded candidates in anchor order
                        const int w = an >> 5;
                        a.out_cand[o] = wprefix[w] + __popc(a.bits[(long)b * a.nwords + w] & ((1u << (an & 31)) - 1u));
                    }
                } else {
                    atomicOr(a.flags, 1);
                }
            }
            if (lane == 0) { s_ctl[0] = nk; s_ctl[1] = out0 + nk; }
        }
        __syncthreads();
        const int nk = s_ctl[0];
        if (nk > 0) {
            for (int j = base + 64 + tid; j < K; j += K3_THREADS) {
                if (removed[j]) continue;
                const float4 bj = box[(int)(uint32_t)keys[j]];
                for (int q = 0; q < nk; ++q) {
                    if (!(iou32(kept_box[q], bj) <= a.iou_thr)) { removed[j] = 1; break; }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int n = s_ctl[1];
        a.out_counts[b] = n < a.max_det ? n : a.max_det;
    }
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Suppression matrix of the images k3_nms handed over.  grid = (K3M_BX, images), 4 waves per block, a wave per 64 x 64 tile
// (row tile ti = the suppressing boxes, column tile tj >= ti), tiles dealt round-robin over the image's waves.
constexpr int K3M_BX = 32;

__global__ void __launch_bounds__(256) k3_mask(K3Args a)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int K = a.m_state[b];
    if (K == 0) return;
    const int nt = (K + 63) >> 6, kmw = a.km >> 6;
    const float4 *mb = a.m_box + (size_t)b * a.km;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < nt * nt; t += K3M_BX * 4) {
        const int ti = t / nt, tj = t - ti * nt;
        if (tj < ti) continue;
        const int col = tj * 64 + lane, row0 = ti * 64;
        const float4 cb = col < K ? mb[col] : zero;
        const float4 rb = row0 + lane < K ? mb[row0 + lane] : zero;
        const int nrow = K - row0 < 64 ? K - row0 : 64;
        unsigned long long mine = 0ull;
        for (int i = 0; i < nrow; ++i) {                       // i is wave-uniform: the row box travels through scalar registers
            float4 kb;
            kb.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.x), i));
            kb.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.y), i));
            kb.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.z), i));
            kb.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rb.w), i));
            const bool sup = col < K && col > row0 + i && !(iou32(kb, cb) <= a.iou_thr);     // detector.py:373, a = the kept box
            const unsigned long long w = __ballot(sup);
            if (lane == i) mine = w;
        }
        if (row0 + lane < K) a.m_mask[((size_t)b * a.km + row0 + lane) * kmw + tj] = mine;
    }
}

// Greedy pass over the suppression matrix: one wave per image, lane w keeps the "removed" word of chunk w.
__global__ void __launch_bounds__(64) k3_reduce(K3Args a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int K = a.m_state[b];
    if (K == 0) return;
    const int nt = (K + 63) >> 6, kmw = a.km >> 6;
    const unsigned long long *mm = a.m_mask + (size_t)b * a.km * kmw;
    unsigned long long removed = 0ull;
    int out = 0;
    for (int c = 0; c < nt; ++c) {
        const int base = c << 6, i = base + lane;
        const unsigned long long diag = i < K ? mm[(size_t)i * kmw + c] : 0ull;
        const unsigned long long rem = readlane64(removed, c);
        const unsigned long long vm = K - base >= 64 ? ~0ull : ((1ull << (K - base)) - 1ull);
        unsigned long long alive = ~rem & vm, kept = 0ull;
        while (alive) {                                           // scalar walk: next alive box survives and clears its victims
            const int l = __builtin_ctzll(alive);
            kept |= 1ull << l;
            alive &= ~readlane64(diag, l);
            alive &= ~(1ull << l);
        }
        // the survivors' rows suppress boxes of later chunks: OR them into the removed words (four loads in flight)
        const bool later = lane > c && lane < nt;
        unsigned long long k2 = kept;
        while (k2) {
            int l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { l[q] = k2 ? __builtin_ctzll(k2) : -1; if (k2) k2 &= k2 - 1ull; }
            unsigned long long r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = (l[q] >= 0 && later) ? mm[(size_t)(base + l[q]) * kmw + lane] : 0ull;
            removed |= r[0] | r[1] | r[2] | r[3];
        }
        if ((kept >> lane) & 1ull) {
            const int pos = out + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < a.max_det) {
                const int an = a.m_anchor[(size_t)b * a.km + i];
                const long o = (long)b * a.max_det + pos;
                a.out_boxes[o] = a.m_box[(size_t)b * a.km + i];
                a.out_scores[o] = a.sp_score[(long)b * a.A + an];
                a.out_cls[o] = a.sp_cls[(long)b * a.A + an];
                if (a.out_anchor) a.out_anchor[o] = an;
                if (a.out_cand) {
                    const int w = an >> 5;
                    a.out_cand[o] = a.m_wprefix[(size_t)b * a.nwords + w] +
                                    __popc(a.bits[(long)b * a.nwords + w] & ((1u << (an & 31)) - 1u));
                }
            } else {
                atomicOr(a.flags, 1);
            }
        }
        out += __popcll(kept);
    }
    if (lane == 0) a.out_counts[b] = out < a.max_det ? out : a.max_det;
}

__global__ void k_zero_counts(int32_t *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

}  // namespace

extern "C" int rva_postprocess_batch(rva_ctx *ctx, const void *raw, int raw_dtype, int batch, int d1, int d2,
                                     double conf_thr, double iou_thr, const int32_t *classes, int n_classes,
                                     const rva_letterbox *metas, int n_metas, int max_det, float *out_boxes,
                                     float *out_scores, int32_t *out_cls, int32_t *out_anchor, int32_t *out_cand,
                                     int32_t *out_counts, int32_t *out_ncand, rva_stream_t stream_)
{
    if (!ctx) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (!raw || batch <= 0 || d1 <= 0 || d2 <= 0 || !metas || (n_metas != 1 && n_metas != batch) || max_det <= 0 ||
        !out_boxes || !out_scores || !out_cls || !out_counts)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_postprocess_batch: bad argument");
    if (raw_dtype != RVA_F16 && raw_dtype != RVA_F32) return rva_fail(ctx, RVA_ERR_ARG, "raw_dtype must be RVA_F16|RVA_F32");
    // detector.py:282-283: transpose iff rows < cols
    const bool channel_major = d1 < d2;
    const int C = channel_major ? d1 : d2, A = channel_major ? d2 : d1;
    if (C < 5) {  // :285-287 -> []
        k_zero_counts<<<rva_ceil_div(batch, 256), 256, 0, stream>>>(out_counts, batch);
        if (out_ncand) k_zero_counts<<<rva_ceil_div(batch, 256), 256, 0, stream>>>(out_ncand, batch);
        RVA_HIP(ctx, hipGetLastError());
        return RVA_OK;
    }
    if (C - 5 > 1024 && n_classes > 0) return rva_fail(ctx, RVA_ERR_ARG, "class filter supports at most 1024 classes");
    int rc = rva_reserve(ctx, batch, A);
    if (rc != RVA_OK) return rc;

    int kcap = 64;
    while (kcap < A && kcap < 16384) kcap <<= 1;
    const int nwords = rva_ceil_div(A, 32);
    const size_t smem = (((size_t)kcap * 9 + 64 * 16 + 16 + 15) & ~(size_t)15) + (size_t)nwords * 4 + 64;
    RVA_HIP(ctx, rva_func_smem((const void *)k3_nms, smem));

    for (int b0 = 0; b0 < batch; b0 += RVA_MAX_BATCH) {
        const int nb = batch - b0 < RVA_MAX_BATCH ? batch - b0 : RVA_MAX_BATCH;
        const size_t esz = raw_dtype == RVA_F16 ? 2 : 4;
        K2Args k2{};
        k2.raw = (const char *)raw + (size_t)b0 * d1 * d2 * esz;
        k2.A = A; k2.C = C;
        k2.sb = (long)d1 * d2;
        k2.sa = channel_major ? 1 : d2;
        k2.sc = channel_major ? d2 : 1;
        k2.thr = (float)conf_thr;
        k2.use_cls = n_classes > 0 && classes;
        for (int i = 0; k2.use_cls && i < n_classes; ++i)
            if (classes[i] >= 0 && classes[i] < 1024) k2.cls_mask[classes[i] >> 5] |= 1u << (classes[i] & 31);
        for (int i = 0; i < nb; ++i) {
            const rva_letterbox &m = metas[n_metas == 1 ? 0 : b0 + i];
            k2.meta[i] = PostMeta{(float)m.pad_left, (float)m.pad_top, (float)m.scale, (float)(m.src_w - 1),
                                  (float)(m.src_h - 1)};
        }
        k2.sp_box = (float4 *)ctx->sp_box; k2.sp_score = ctx->sp_score; k2.sp_cls = ctx->sp_cls;
        k2.list = ctx->cand_list; k2.count = ctx->cand_count; k2.bits = ctx->cand_bits; k2.nwords = nwords;
        k_zero_counts<<<1, 64, 0, stream>>>(ctx->cand_count, nb);
        dim3 g2(rva_ceil_div(A, 256), nb);
        if (raw_dtype == RVA_F16) k2_decode<__half><<<g2, 256, 0, stream>>>(k2);
        else k2_decode<float><<<g2, 256, 0, stream>>>(k2);

        K3Args k3{};
        k3.sp_box = (const float4 *)ctx->sp_box; k3.sp_score = ctx->sp_score; k3.sp_cls = ctx->sp_cls;
        k3.list = ctx->cand_list; k3.count = ctx->cand_count; k3.bits = ctx->cand_bits;
        k3.nwords = nwords; k3.A = A; k3.kcap = kcap;
        k3.iou_thr = (float)iou_thr;
        k3.max_det = max_det;
        k3.out_boxes = (float4 *)out_boxes + (size_t)b0 * max_det;
        k3.out_scores = out_scores + (size_t)b0 * max_det;
        k3.out_cls = out_cls + (size_t)b0 * max_det;
        k3.out_anchor = out_anchor ? out_anchor + (size_t)b0 * max_det : nullptr;
        k3.out_cand = out_cand ? out_cand + (size_t)b0 * max_det : nullptr;
        k3.out_counts = out_counts + b0;
        k3.out_ncand = out_ncand ? out_ncand + b0 : nullptr;
        k3.flags = ctx->post_flags;
        k3.km = ctx->k3_km; k3.m_state = ctx->k3_state; k3.m_anchor = ctx->k3_anchor; k3.m_box = (float4 *)ctx->k3_box;
        k3.m_mask = ctx->k3_mask; k3.m_wprefix = ctx->k3_wprefix;
        k3_nms<<<nb, K3_THREADS, smem, stream>>>(k3);
        // busy images (k3_state[b] > 0) continue here; for the others both launches return at once
        k3_mask<<<dim3(K3M_BX, nb), 256, 0, stream>>>(k3);
        k3_reduce<<<nb, 64, 0, stream>>>(k3);
    }
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

extern "C" int rva_post_status(rva_ctx *ctx, rva_stream_t stream_, int *flags)
{
    if (!ctx || !flags) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int32_t v = 0;
    RVA_HIP(ctx, hipMemcpyAsync(&v, ctx->post_flags, sizeof v, hipMemcpyDeviceToHost, stream));
    RVA_HIP(ctx, hipMemsetAsync(ctx->post_flags, 0, sizeof v, stream));
    RVA_HIP(ctx, hipStreamSynchronize(stream));
    *flags = v;
    return RVA_OK;
}

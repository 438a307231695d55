// K2 (decode / score / threshold) and K3 (sort + greedy NMS) for a batch of YOLO heads.
//
// Replaces _TensorRTBaseDetector._postprocess and helpers (detector.py:266-375, 469-481).
// float32 arithmetic in the reference's operation order; the library is built with
// -ffp-contract=off so no multiply is fused into a following add (that would flip borderline
// `iou <= thr` decisions).  Thresholds are rounded to float32 first, as numpy's NEP-50 does.
//
// Data flow per image b (no host round trip, counts stay on the device):
//   K2: one thread per anchor, coalesced along the anchor axis of [C, A] heads.  Anchors that pass write box / score /
//       class at their own anchor index (sparse arrays) and their bit in a pass bitmap (anchor order, one ballot per
//       wave: no atomics, no memset).
//   K3: one 1024-thread workgroup per image.
//       * the candidates are the set bits of the bitmap: a block scan of the word popcounts gives every word its first
//         position (and every survivor its `keep` index later), each thread expands words into 64-bit keys
//         (~score, anchor);
//       * the keys are sorted -- up to 4096 of them by a stable LSD radix sort on the score half (k3_radix), more by a bitonic
//         network in registers / wave shuffles / LDS (k3_sort).  A total order, so the result does not depend on anything but
//         the values: (score desc, anchor asc), the project's tie rule;
//       * greedy NMS over the sorted list in rounds of 64, 128, 256, 512, 512 ... boxes.  Phase 1: every box against the boxes
//         kept so far (an LDS list, from 96 boxes on sorted into centre-x bins so that a box scans only the bins a suppressor's
//         centre can lie in) -- in a busy scene nine boxes in ten die here.  Phase 2: the survivors (compacted in order) among
//         themselves: their suppression matrix into LDS by all waves, then one wave replays the greedy order on it, 64
//         survivors per step.  Greedy NMS is order-determined, so the keep set equals the reference's sequential loop
//         (detector.py:365-375); the work is K x (a slice of) kept + survivors^2 / 2 IoUs instead of the K^2 / 2 of a full
//         suppression matrix.  DESIGN.md section 4 (K3) has the details and the cycle counts.
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

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
    uint32_t *bits;
    int nwords;
    int32_t *irr;     // [batch] set when a candidate box of the image has x1 > x2 or y1 > y2 (negative raw width / height): K3's
                      // centre-bin filter assumes proper boxes and steps aside for such an image
};

__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const __half *p) { return __half2float(*p); }

template <typename T>
__global__ void __launch_bounds__(256) k2_decode(K2Args a)
{
    const int b = blockIdx.y;
    const int anchor = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool pass = false, odd = false;
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
            odd = !(x1 <= x2) || !(y1 <= y2);
        }
    }
    if (__any(odd) && lane == 0) atomicOr(a.irr + b, 1);          // practically never: real heads have w, h >= 0
    const unsigned long long mask = __ballot(pass);
    // pass bitmap: this wave owns anchors [a0, a0+64) = words a0/32 and a0/32+1 (written whole)
    const int a0 = blockIdx.x * 256 + (threadIdx.x & ~63);
    if (lane < 2) {
        const int w = (a0 >> 5) + lane;
        if (w < a.nwords) a.bits[(long)b * a.nwords + w] = (uint32_t)(mask >> (32 * lane));
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

// `!(iou32(a, b) <= thr)` without the division.  q = RN(inter / u) is monotonic in the real quotient, so q <= thr exactly when
// inter / u does not exceed the point where rounding leaves thr: the midpoint m of thr and the next float above it, itself
// included when a tie rounds to thr (thr's mantissa even), excluded otherwise.  m has 25 significant bits and u 24: the
// double product m * u is exact, and so is the comparison.  NaN operands make every comparison false -> suppressed, as a NaN
// quotient does in the reference's `iou <= thr` (detector.py:373).  Same float32 operations as iou32() up to u.
struct SupTest { double m; int tie_incl; float thr, thr_up; };

// v_max_f32 / v_min_f32 as fmaxf / fminf compute them for quiet NaNs (the other operand); written as instructions because
// the compiler puts a canonicalising v_max_f32 x, x in front of every fmaxf / fminf operand that comes from memory
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// numerator and denominator of iou32(a, b), the same float32 operations
__device__ __forceinline__ void iou_terms(const float4 a, const float area_a, const float4 b, const float area_b, float &inter, float &uni)
{
    const float x1 = vmax(a.x, b.x), y1 = vmax(a.y, b.y);
    const float x2 = vmin(a.z, b.z), y2 = vmin(a.w, b.w);
    float w = x2 - x1, h = y2 - y1;
    w = w > 0.0f ? w : 0.0f;
    h = h > 0.0f ? h : 0.0f;
    inter = w * h;
    uni = area_a + area_b - inter;
    uni = vmax(uni, 1e-6f);
}

__device__ __forceinline__ bool suppresses(const float4 a, const float area_a, const float4 b, const float area_b, const SupTest t)
{
    float inter, uni;
    iou_terms(a, area_a, b, area_b, inter, uni);
    const double di = (double)inter, rhs = t.m * (double)uni;
    return !(t.tie_incl ? di <= rhs : di < rhs);
}

// The same decision from two float32 FMAs wherever they settle it.  fma(-c, u, inter) is the correctly rounded inter - c u, so
// its sign is the sign of inter / u - c: not above thr -> the rounded quotient is <= thr, kept; at or above the next float
// after thr -> the rounded quotient is above thr, suppressed.  Strictly between the two (or NaN) -> `unsure`, and the caller
// asks suppresses().  Returns "suppressed for certain".
__device__ __forceinline__ bool suppresses_fast(const float4 a, const float area_a, const float4 b, const float area_b, const SupTest t, bool &unsure)
{
    float inter, uni;
    iou_terms(a, area_a, b, area_b, inter, uni);
    const float lo = __builtin_fmaf(-t.thr, uni, inter), hi = __builtin_fmaf(-t.thr_up, uni, inter);
    const bool yes = hi >= 0.0f;
    unsure |= !(lo <= 0.0f) & !yes;
    return yes;
}
__device__ __forceinline__ float box_area(const float4 b) { return (b.z - b.x) * (b.w - b.y); }

// a 16-byte load that is a global_load for certain (a flat load would also count as an LDS operation and hold up every LDS wait
// while it is in flight)
typedef float k3_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_global_f4(const float4 *p)
{
    const k3_f4 v = *reinterpret_cast<const __attribute__((address_space(1))) k3_f4 *>(reinterpret_cast<size_t>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

struct K3Args {
    const float4 *sp_box;
    const float *sp_score;
    const int32_t *sp_cls;
    const uint32_t *bits;
    int nwords, A, kcap;  // kcap: power of two, LDS key capacity
    float iou_thr;
    SupTest sup;          // the same decision as `!(iou <= iou_thr)`, division-free (see suppresses())
    int max_det;
    float4 *out_boxes;
    float *out_scores;
    int32_t *out_cls, *out_anchor, *out_cand, *out_counts, *out_ncand;
    int32_t *flags;
    int32_t *irr;         // [batch] K2's "improper box" flag per image (read, then cleared for the next launch)
    int bitonic_upto;     // experiment switch (RVA_K3_BITONIC_UPTO): candidate counts up to this take the bitonic network instead of the radix sort
    float rfac;           // centre-bin filter: a kept box can suppress a box of width w only if their centres are within rfac w
                          // (+ rounding slack) in x; 0 = filter off (threshold below K3_BIN_MIN_THR)
    float bin_scale[RVA_MAX_BATCH];      // K3_BINS / source width of the image
};

#ifdef RVA_K3_STAMPS
// diagnostic build only (tools/k3_stamps.py): s_memtime sums per phase, written by thread 0 of every block
__device__ unsigned long long g_k3_stamps[64][8];
#define K3_STAMP(slot) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[slot] += t_ - st_last; st_last = t_; } } while (0)
#else
#define K3_STAMP(slot) do { } while (0)
#endif

constexpr int K3_THREADS = 1024;   // 16 waves: four per SIMD hide the LDS / ALU latencies of the IoU loops (two did not)
constexpr int K3_WAVES = K3_THREADS / 64;
#ifndef RVA_K3_NEWEST
#define RVA_K3_NEWEST 64
#endif
constexpr int K3_NEWEST = RVA_K3_NEWEST;     // phase 1, stage A: the kept boxes every box of a round is tested against before the round is thinned out
constexpr int K3_KBL = 1024;       // kept boxes held in LDS for phase 1 at most (K3Args::kbl; further ones are read back from out_boxes)
constexpr int K3_BINS = 64;        // centre-x bins of the kept list (phase 1, see k3_nms)
constexpr int K3_BIN_MIN_KEPT = 96; // kept boxes from which a round sorts the list into bins
constexpr double K3_BIN_MIN_THR = 0.15;  // below it the centre distance bound is wider than three box widths: not worth a filter

extern __shared__ __attribute__((aligned(16))) unsigned char k3_smem[];

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Bitonic sort of Kpad = 512 E keys, ascending; thread t holds elements t E .. t E + E - 1 in registers.
template <int E>
__device__ __forceinline__ void k3_sort(unsigned long long *keys, int tid)
{
    constexpr int KP = K3_THREADS * E;
    unsigned long long v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = keys[tid * E + e];
    for (int k = 2; k <= KP; k <<= 1) {
        for (int j = k >> 1; j >= E; j >>= 1) {
            if (j >= 64 * E) {            // partner in another wave: through LDS
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) keys[tid * E + e] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int g = tid * E + e;
                    const unsigned long long o = keys[g ^ j];
                    const bool up = (g & k) == 0, lower = (g & j) == 0;
                    const unsigned long long mn = v[e] < o ? v[e] : o, mx = v[e] < o ? o : v[e];
                    v[e] = (lower == up) ? mn : mx;
                }
            } else {                      // partner thread in this wave: shuffle
                const int dl = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int g = tid * E + e;
                    const unsigned lo = __shfl_xor((unsigned)v[e], dl), hi = __shfl_xor((unsigned)(v[e] >> 32), dl);
                    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                    const bool up = (g & k) == 0, lower = (g & j) == 0;
                    const unsigned long long mn = v[e] < o ? v[e] : o, mx = v[e] < o ? o : v[e];
                    v[e] = (lower == up) ? mn : mx;
                }
            }
        }
        // partners in this thread's own registers: strides E/2 ... 1, every index a compile-time constant (a register
        // array indexed by a run-time stride would live in scratch)
#pragma unroll
        for (int jj = E / 2; jj >= 1; jj >>= 1) {
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const int g = tid * E + e;
                        const bool up = (g & k) == 0;
                        const unsigned long long x = v[e], y = v[e | jj];
                        const bool swap = (x > y) == up;
                        v[e] = swap ? y : x;
                        v[e | jj] = swap ? x : y;
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) keys[tid * E + e] = v[e];
    __syncthreads();
}

// Stable LSD radix sort of the K valid keys by their high 32 bits (the score part), 8 bits per pass over the bits in which the keys
// differ, src -> dst -> src ... (an odd number of passes is copied back into `a`).  The keys were written in anchor order and equal scores keep it, so the result is the bitonic sort's total
// order (score descending, anchor ascending) at a third of its vector instructions.  Wave w owns elements [w 64 E, (w + 1) 64 E),
// slot e of lane l is element w 64 E + 64 e + l: the stable order inside a wave is (slot, lane).  Per pass: a wave counts its
// digits slot by slot (lanes with the same digit find one another with eight ballots; the rank of an element among them is a
// popcount below the lane; the lowest lane of a group adds the group to the wave's counter), 256 threads turn the counters
// hist[wave][digit] into start positions in (digit, wave) order, and every element goes to start + its rank in the wave.
template <int E>
__device__ __forceinline__ void k3_radix(unsigned long long *a, unsigned long long *b2, int *hist, int *wtot, int K, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long *src = a, *dst = b2;
    int *wh = hist + wave * 256;
    // Only the bits in which the score keys DIFFER need sorting: the scores of an fp16 head carry 11 significant bits (the low 13
    // bits of every float32 key are zero) and sit within a factor of four of one another -- 12-13 varying bits, two passes
    // instead of four.  OR and AND of all keys: a wave reduction, then one LDS atomic per wave.
    unsigned vor = 0u, vand = ~0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = wave * 64 * E + e * 64 + lane;
        if (idx < K) { const unsigned h = (unsigned)(a[idx] >> 32); vor |= h; vand &= h; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { vor |= __shfl_xor(vor, o); vand &= __shfl_xor(vand, o); }
    if (tid == 0) { wtot[8] = 0; wtot[9] = -1; }
    __syncthreads();
    if (lane == 0) { atomicOr(reinterpret_cast<unsigned *>(wtot) + 8, vor); atomicAnd(reinterpret_cast<unsigned *>(wtot) + 9, vand); }
    __syncthreads();
    const unsigned varying = reinterpret_cast<unsigned *>(wtot)[8] ^ reinterpret_cast<unsigned *>(wtot)[9];
    if (varying == 0u) return;                            // all scores equal: anchor order is the answer already (uniform)
    const int lowbit = __builtin_ctz(varying), npass = (32 - __builtin_clz(varying) - lowbit + 7) >> 3;
    for (int pass = 0; pass < npass; ++pass) {
        const int sh = 32 + lowbit + 8 * pass;
        for (int i = tid; i < K3_WAVES * 256; i += K3_THREADS) hist[i] = 0;
        __syncthreads();
        unsigned long long v[E];
        int rk[E], dg[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = wave * 64 * E + e * 64 + lane;
            const bool in = idx < K;
            v[e] = in ? src[idx] : ~0ull;
            const int d = (int)(v[e] >> sh) & 255;
            dg[e] = d;
            unsigned long long m = __ballot(in);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (d >> bit) & 1;
                const unsigned long long bb = __ballot(in && one);
                m &= one ? bb : ~bb;
            }
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int old = in ? wh[d] : 0;
            rk[e] = old + below;
            if (in && below == 0) wh[d] = old + __popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the next slot reads what this one's group leaders wrote
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        int c[K3_WAVES], run = 0, incl = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < K3_WAVES; ++w) { const int t = hist[w * 256 + tid]; c[w] = run; run += t; }
            incl = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wtot[wave] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            int base = incl - run;
            for (int q = 0; q < wave; ++q) base += wtot[q];
#pragma unroll
            for (int w = 0; w < K3_WAVES; ++w) hist[w * 256 + tid] = base + c[w];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = wave * 64 * E + e * 64 + lane;
            if (idx < K) dst[wh[dg[e]] + rk[e]] = v[e];
        }
        __syncthreads();
        unsigned long long *t = src; src = dst; dst = t;
    }
    if (npass & 1) {                                      // an odd number of passes ends in the second buffer
        for (int i = tid; i < K; i += K3_THREADS) a[i] = b2[i];
        __syncthreads();
    }
}

// output row `pos` of image b: kept box, its score / class / anchor, its index among the thresholded candidates in anchor order
template <bool BOX = true>
__device__ __forceinline__ void k3_emit(const K3Args &a, int b, int pos, const float4 kbx, int kan, const float *score, const uint32_t *bw, const int *wprefix)
{
    const long o = (long)b * a.max_det + pos;
    if (BOX) a.out_boxes[o] = kbx;
    a.out_scores[o] = score[kan];
    a.out_cls[o] = a.sp_cls[(long)b * a.A + kan];
    if (a.out_anchor) a.out_anchor[o] = kan;
    if (a.out_cand) {
        const int w = kan >> 5;
        a.out_cand[o] = wprefix[w] + __popc(bw[w] & ((1u << (kan & 31)) - 1u));
    }
}

__global__ void __launch_bounds__(K3_THREADS) k3_nms(K3Args a)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS carve, fixed part (all offsets multiples of 16); the large arrays follow once the number of candidates is known
    size_t off = 0;
    int *wprefix = (int *)(k3_smem + off); off += (((size_t)a.nwords * 4 + 15) & ~(size_t)15);             // [nwords]
    int *wave_tot = (int *)(k3_smem + off); off += 64;                       // [16]
    unsigned long long *half_alive = (unsigned long long *)(k3_smem + off); off += 128;   // [16] phase-1 ballots of the two halves
    int *s_ctl = (int *)(k3_smem + off); off += 64;                          // [0] survivors, [1] kept so far, [2] scan carry, [3] K
    int *binstart = (int *)(k3_smem + off); off += (K3_BINS + 4) * 4;        // [K3_BINS + 1] first list position of a centre-x bin
    int *bincnt = (int *)(k3_smem + off); off += K3_BINS * 4;                // [K3_BINS]
    // Centre-bin filter of phase 1 (binned mode).  With proper boxes (x1 <= x2, y1 <= y2) a kept box B suppresses a box A only if
    // the float32 quotient exceeds thr, which needs  overlap_x >= thr' max(wA, wB)  (thr' = thr less a few ulps: inter <= overlap_x
    // min(hA, hB), union >= the larger area >= max(wA, wB) min(hA, hB)); with  |cA - cB| <= (wA + wB) / 2 - overlap_x  and
    // wB <= wA / thr'  that bounds the centre distance by  wA max(1 - thr', 1 / (2 thr') - 1 / 2).  a.rfac is that factor with 1 %
    // on top, the radius gets an absolute slack far above the rounding of the centres: no suppressor is ever outside
    // [cA - R, cA + R].  The kept list is therefore kept SORTED BY CENTRE-X BIN (a counting sort at the start of every round once it holds
    // K3_BIN_MIN_KEPT boxes; the
    // keep order lives in kj / out_boxes) and a box scans the slice of its bins only -- a real kept box that happens to sit in
    // the slice's alignment padding and hits is a true suppressor too, so nothing is masked.  Images with an improper box (K2
    // flags them) or a threshold below K3_BIN_MIN_THR take the unsorted list.
    const bool binned = a.rfac > 0.f && a.irr[b] == 0;
    const float bscale = a.bin_scale[b];
    auto xbin = [&](float x) { const int v = (int)floorf(x * bscale); return v < 0 ? 0 : (v > K3_BINS - 1 ? K3_BINS - 1 : v); };

#ifdef RVA_K3_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    // ---- candidates = set bits of the pass bitmap: exclusive prefix of the word popcounts (512 words per round)
    const uint32_t *bw = a.bits + (long)b * a.nwords;
    if (tid == 0) { s_ctl[1] = 0; s_ctl[2] = 0; }
    if (tid < K3_BINS) bincnt[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        a.irr[b] = 0;                                             // every thread has read it: clean for the next launch
        if (binned) atomicAdd(a.flags + 1 + RVA_MAX_BATCH, 1);    // rva_post_filter_stats()
    }
    for (int w0 = 0; w0 < a.nwords; w0 += K3_THREADS) {
        const int w = w0 + tid;
        const int c = w < a.nwords ? __popc(bw[w]) : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int base = s_ctl[2];
        for (int q = 0; q < wave; ++q) base += wave_tot[q];
        if (w < a.nwords) wprefix[w] = base + incl - c;
        __syncthreads();
        if (tid == K3_THREADS - 1) s_ctl[2] = base + incl;      // carry into the next round = total so far
        __syncthreads();
    }
    int K = s_ctl[2];
    if (a.out_ncand && tid == 0) a.out_ncand[b] = K;
    if (K > a.kcap) {
        if (tid == 0) atomicOr(a.flags, 2);
        K = a.kcap;
    }
    if (K == 0) {
        if (tid == 0) a.out_counts[b] = 0;
        return;
    }
    int Kpad = K3_THREADS;
    while (Kpad < K) Kpad <<= 1;
    // The large arrays.  Up to 8192 candidates (64 KB of keys) leave room for rounds of 512 boxes and 1024 kept boxes in LDS; only an
    // image with more (128 KB of keys) falls back to rounds of 256 / 512 kept boxes -- the launch reserves LDS for either.
    const int SC = Kpad <= 8192 ? 512 : 256, KBL = Kpad <= 8192 ? K3_KBL : K3_KBL / 2;
    unsigned long long *keys = (unsigned long long *)(k3_smem + off); off += (size_t)Kpad * 8;                // [Kpad]
    float4 *kb = (float4 *)(k3_smem + off); off += (size_t)(KBL + 8) * 16;   // kept boxes so far; the rest stays all-zero (such a box suppresses nothing a kept box would not)
    float4 *sv_box = (float4 *)(k3_smem + off); off += (size_t)SC * 16;     // survivors of phase 1, in order
    int *sv_j = (int *)(k3_smem + off); off += (size_t)SC * 4;              // their position in the sorted list
    int *kj = (int *)(k3_smem + off); off += (size_t)KBL * 4;               // kept boxes so far: position in the sorted list
    float *ka = (float *)(k3_smem + off); off += (size_t)(KBL + 8) * 4;     // their areas (zero past the list, like the boxes)
    unsigned long long *smask = (unsigned long long *)(k3_smem + off);      // [SC][SC/64]
    const float *score = a.sp_score + (long)b * a.A;
    const float4 *box = a.sp_box + (long)b * a.A;
    for (int i = K + tid; i < Kpad; i += K3_THREADS) keys[i] = ~0ull;
    for (int w = tid; w < a.nwords; w += K3_THREADS) {
        uint32_t bits = bw[w];
        int pos = wprefix[w];
        while (bits) {
            const int an = (w << 5) + __builtin_ctz(bits);
            bits &= bits - 1u;
            if (pos < a.kcap) {
                uint32_t u = __float_as_uint(score[an]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // ascending-orderable
                keys[pos] = ((unsigned long long)(~u) << 32) | (uint32_t)an;   // descending score, ascending anchor
            }
            ++pos;
        }
    }
    __syncthreads();
    K3_STAMP(0);
    // up to 4096 candidates: radix sort (second buffer, counters and wave totals borrow smask / kb / sv_box, all idle until the
    // first round); more: the bitonic network in place
    switch (Kpad <= a.bitonic_upto ? 64 + Kpad / K3_THREADS : Kpad / K3_THREADS) {
    case 65: k3_sort<1>(keys, tid); break;
    case 66: k3_sort<2>(keys, tid); break;
    case 68: k3_sort<4>(keys, tid); break;
    case 1: k3_radix<1>(keys, smask, (int *)kb, (int *)sv_box, K, tid); break;
    case 2: k3_radix<2>(keys, smask, (int *)kb, (int *)sv_box, K, tid); break;
    case 4: k3_radix<4>(keys, smask, (int *)kb, (int *)sv_box, K, tid); break;
    case 8: k3_sort<8>(keys, tid); break;
    default: k3_sort<16>(keys, tid); break;
    }
    for (int i = tid; i < KBL + 8; i += K3_THREADS) { kb[i] = make_float4(0.f, 0.f, 0.f, 0.f); ka[i] = 0.f; }
    __syncthreads();

    K3_STAMP(1);
    const int scw = SC >> 6;                                       // mask words per survivor row
    // Rounds grow 64, 128, 256, ... up to SC boxes: the first rounds have no kept list to thin them out, so every box of theirs
    // reaches the quadratic phase 2 -- 64^2/2 + 128^2/2 + 256^2/2 IoUs instead of 512^2/2 -- and they fill the kept list that
    // lets phase 1 kill most of the later, full-size rounds.
    int rs = 64;
    const int bt = tid & (K3_THREADS / 2 - 1), half = tid / (K3_THREADS / 2);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nbx = bt < rs && bt < K ? load_global_f4(box + (int)(uint32_t)keys[bt]) : zero4;       // the first round's box
    for (int base = 0; base < K; base += rs, rs = rs * 2 < SC ? rs * 2 : SC) {
        // ---- phase 1: a box against everything kept so far; two threads per box (t and t + 512) share the kept list
        // (even / odd groups of four)
        const int j = base + bt;
        const bool valid = bt < rs && j < K;
        const float4 bx = nbx;
        const float area_b = box_area(bx);
        asm volatile("" :: "v"(bx.x), "v"(bx.y), "v"(bx.z), "v"(bx.w), "v"(area_b));      // this round's box has arrived before ...
        {   // ... the next round's box sets out: it travels while this round works
            const int nrs = rs * 2 < SC ? rs * 2 : SC, jn = base + rs + bt;
            nbx = bt < nrs && jn < K ? load_global_f4(box + (int)(uint32_t)keys[jn]) : zero4;
        }
        bool alive = valid;
        const int nk = s_ctl[1] < a.max_det ? s_ctl[1] : a.max_det;
        const int nk_lds = nk < KBL ? nk : KBL;
        const bool rb = binned && nk_lds >= K3_BIN_MIN_KEPT;      // a short kept list is scanned whole: sorting it costs more than it saves
        if (rb) {
            // ---- the kept list into centre-x bin order: a counting sort in place -- every thread holds its entry in registers
            // across the barrier, so nothing is overwritten before it was read; every wave scans the 64 bin counts for itself
            const bool mine = tid < nk_lds, wave_in = (tid & ~63) < nk_lds;      // waves without an entry only keep the barriers
            float4 mb = zero4;
            float ma = 0.f;
            int mbin = 0, slot = 0;
            if (wave_in) {
                if (mine) { mb = kb[tid]; ma = ka[tid]; }
                mbin = xbin((mb.x + mb.z) * 0.5f);
                if (mine) slot = atomicAdd(&bincnt[mbin], 1);
            }
            __syncthreads();
            if (wave_in) {
                const int c = bincnt[lane];
                int incl = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                const int dst = __shfl(incl - c, mbin) + slot;
                if (wave == 0) {
                    binstart[lane] = incl - c;
                    if (lane == 63) binstart[64] = incl;
                }
                if (mine) { kb[dst] = mb; ka[dst] = ma; }
            }
            __syncthreads();
            if (tid < K3_BINS) bincnt[tid] = 0;                     // for the next round (many barriers away)
            K3_STAMP(6);
        }
        if (rb) {
            // ---- phase 1, binned, stage A: the kept boxes of the box's own centre-x bin and its two neighbours (a box's suppressor is
            // usually the head of its own cluster: nearly the same centre), four per step (per-lane addresses), the two threads of a
            // box take alternate groups of four.  Stage B below: the survivors against their whole slice.
            const float cx = (bx.x + bx.z) * 0.5f;
            const int bc = xbin(cx);
            const int s0 = nk_lds > 0 ? binstart[bc > 0 ? bc - 1 : 0] & ~3 : 0, s1 = nk_lds > 0 ? binstart[(bc < K3_BINS - 1 ? bc + 1 : bc) + 1] : 0;
            for (int i = s0 + 4 * half; ; i += 8) {
                if (!__any(alive && i < s1)) break;
                const int ii = i < KBL ? i : KBL;                  // lanes past their slice idle on the zero padding
                float4 k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) k4[u] = kb[ii + u];
                const float4 a4 = *reinterpret_cast<const float4 *>(ka + ii);
                const float ar[4] = {a4.x, a4.y, a4.z, a4.w};
                bool hit = false, unsure = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) hit |= suppresses_fast(k4[u], ar[u], bx, area_b, a.sup, unsure);     // detector.py:373
                if (__any(unsure)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) hit |= suppresses(k4[u], ar[u], bx, area_b, a.sup);
                }
                alive = alive && !hit;
            }
        }
        // Stage A: the K3_NEWEST most recently kept boxes (a box's suppressor scored only a little higher than the box itself,
        // so it is usually among them).  Four kept boxes per step, branch-free: the (broadcast) LDS reads and the four tests
        // of a step overlap, one early-exit test per step; reads past the list find all-zero boxes.
        const int lo = !rb && nk_lds > K3_NEWEST ? (nk_lds - K3_NEWEST) & ~7 : 0;        // stage B takes the kept boxes [0, lo)
        for (int i = lo + 4 * half; !rb && i < nk_lds; i += 8) {
            if (!__any(alive)) break;
            float4 k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k4[u] = kb[i + u];
            const float4 a4 = *reinterpret_cast<const float4 *>(ka + i);
            const float ar[4] = {a4.x, a4.y, a4.z, a4.w};
            bool hit = false, unsure = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) hit |= suppresses_fast(k4[u], ar[u], bx, area_b, a.sup, unsure);     // detector.py:373
            if (__any(unsure)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) hit |= suppresses(k4[u], ar[u], bx, area_b, a.sup);
            }
            alive = alive && !hit;
        }
        const float4 *kept_glb = a.out_boxes + (long)b * a.max_det;
        for (int i = KBL + half; i < nk; i += 2) {             // more boxes kept in one image than the LDS list holds: the rest from HBM
            if (!__any(alive)) break;
            const float4 kg = kept_glb[i];
            alive = alive && !suppresses(kg, box_area(kg), bx, area_b, a.sup);
        }
        K3_STAMP(2);
        // ---- boxes still alive (in BOTH halves of the kept list), compacted in sorted order by the lower 512 threads
        const unsigned long long am0 = __ballot(alive);
        if (lane == 0) half_alive[wave] = am0;
        __syncthreads();
        const unsigned long long am = half_alive[wave & (K3_WAVES / 2 - 1)] & half_alive[(wave & (K3_WAVES / 2 - 1)) + K3_WAVES / 2];
        alive = half == 0 && ((am >> lane) & 1ull);
        if (lane == 0) wave_tot[wave] = half == 0 ? __popcll(am) : 0;
        __syncthreads();
        int so, ns;
        {   // exclusive prefix / total of the eight wave counts: one LDS read, a scan in the wave
            const int c = lane < K3_WAVES / 2 ? wave_tot[lane] : 0;
            int incl = c;
#pragma unroll
            for (int o = 1; o < K3_WAVES / 2; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            so = __shfl(incl - c, wave & (K3_WAVES / 2 - 1));
            ns = __shfl(incl, K3_WAVES / 2 - 1);
            if (wave >= K3_WAVES / 2) so = 0;
        }
        if (alive) {
            const int sidx = so + __popcll(am & ((1ull << lane) - 1ull));
            sv_box[sidx] = bx;
            sv_j[sidx] = j;
        }
        __syncthreads();
        // Stage B: what is left (a fraction) against the older kept boxes [0, lo), all 1024 threads again: thread t takes box
        // t mod mp (mp = the count padded to whole waves) and every (1024 / mp)-th group of four kept boxes; a hit marks the box.
        if ((rb || lo > 0) && ns > 0) {           // uniform
            const int mp = (ns + 63) & ~63, parts = K3_THREADS / mp;
            const int part = tid / mp, bi = tid - part * mp;      // part is the same for a whole wave
            bool open = part < parts && bi < ns;
            const float4 ob = open ? sv_box[bi] : zero4;
            const float area_o = box_area(ob);
            if (rb) {                                             // the survivor's slice: every bin a suppressor's centre can lie in
                const float cx = (ob.x + ob.z) * 0.5f;
                const float rad = (ob.z - ob.x) * a.rfac + (ob.x + ob.z) * 1e-5f + 1e-2f;
                const int s0 = binstart[xbin(cx - rad)] & ~3, s1 = binstart[xbin(cx + rad) + 1];
                for (int i = s0 + 4 * part; ; i += 4 * parts) {
                    if (!__any(open && i < s1)) break;
                    const int ii = i < KBL ? i : KBL;
                    float4 k4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) k4[u] = kb[ii + u];
                    const float4 a4 = *reinterpret_cast<const float4 *>(ka + ii);
                    const float ar[4] = {a4.x, a4.y, a4.z, a4.w};
                    bool hit = false, unsure = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) hit |= suppresses_fast(k4[u], ar[u], ob, area_o, a.sup, unsure);
                    if (__any(unsure)) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) hit |= suppresses(k4[u], ar[u], ob, area_o, a.sup);
                    }
                    if (open && hit) sv_j[bi] = -1;
                    open = open && !hit;
                }
            }
            for (int i = 4 * part; !rb && i < lo; i += 4 * parts) {
                if (!__any(open)) break;
                float4 k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) k4[u] = kb[i + u];
                const float4 a4 = *reinterpret_cast<const float4 *>(ka + i);
                const float ar[4] = {a4.x, a4.y, a4.z, a4.w};
                bool hit = false, unsure = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) hit |= suppresses_fast(k4[u], ar[u], ob, area_o, a.sup, unsure);
                if (__any(unsure)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) hit |= suppresses(k4[u], ar[u], ob, area_o, a.sup);
                }
                if (open && hit) sv_j[bi] = -1;                   // several parts may say so: the same value
                open = open && !hit;
            }
            __syncthreads();
            // second compaction, in place: every survivor is read before any is written
            const bool mine = tid < ns;
            const float4 sb = mine ? sv_box[tid] : zero4;
            const int sj = mine ? sv_j[tid] : -1;
            const unsigned long long lm = __ballot(sj >= 0);
            if (lane == 0) wave_tot[wave] = __popcll(lm);
            __syncthreads();
            int so2, ns2;
            {
                const int c = lane < K3_WAVES / 2 ? wave_tot[lane] : 0;
                int incl = c;
#pragma unroll
                for (int o = 1; o < K3_WAVES / 2; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                so2 = __shfl(incl - c, wave & (K3_WAVES / 2 - 1));
                ns2 = __shfl(incl, K3_WAVES / 2 - 1);
            }
            if (sj >= 0) {
                const int sidx = so2 + __popcll(lm & ((1ull << lane) - 1ull));
                sv_box[sidx] = sb;
                sv_j[sidx] = sj;
            }
            ns = ns2;
            __syncthreads();
        }
        if (ns > 64) {                                            // words below the diagonal are ORed together in phase 2a
            for (int x = 64 + tid; x < ns; x += K3_THREADS)
                for (int t = 0; t < (x >> 6); ++t) smask[(size_t)x * scw + t] = 0ull;
            __syncthreads();
        }
        K3_STAMP(3);
        if (ns == 0) continue;                                    // uniform
        // ---- phase 2a: who suppresses whom among the survivors, smask[x][t] = the survivors of tile t (64 of them) that
        // x and they suppress one another -- the float32 terms of the test are symmetric, so one test per pair fills both
        // directions: for every pair of tiles rt <= ct a wave takes four rows of rt; lanes are the columns of ct; a row's
        // ballot is its word for tile ct, and a lane ORs its four answers into its own (zeroed) word for tile rt.
        const int nct = (ns + 63) >> 6;
        for (int ct = 0, rt = 0; ct < nct; rt == ct ? (++ct, rt = 0) : ++rt) {
            const int col = ct * 64 + lane;
            const float4 cb = col < ns ? sv_box[col] : zero4;
            const float area_c = box_area(cb);
            const int rbase = rt * 64 + wave * 4;
            if (rbase >= ns) continue;                            // wave-uniform
            float4 rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[u] = sv_box[rbase + u < ns ? rbase + u : rbase];      // one address for the wave: a broadcast read
            unsigned mine = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rbase + u;
                bool unsure = false;
                bool sup = suppresses_fast(rb[u], box_area(rb[u]), cb, area_c, a.sup, unsure);
                if (__any(unsure)) sup = suppresses(rb[u], box_area(rb[u]), cb, area_c, a.sup);
                sup = sup && col < ns && r < ns && col != r;
                const unsigned long long w = __ballot(sup);
                if (lane == 0 && r < ns) smask[(size_t)r * scw + ct] = w;
                mine |= (sup ? 1u : 0u) << u;
            }
            if (rt < ct && mine)
                __hip_atomic_fetch_or(reinterpret_cast<unsigned *>(&smask[(size_t)col * scw + rt]) + (wave >> 3), mine << ((wave & 7) * 4),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        K3_STAMP(4);
        // ---- phase 2b: the greedy order on the matrix, one wave, 64 survivors per step.  A survivor is out if a box kept in an
        // earlier step suppresses it; within the step it is kept exactly when no KEPT earlier survivor of the step suppresses it:
        // settled in passes -- out once such a one is kept, kept once none of them is still open -- the lowest open survivor
        // settles in every pass, clusters settle in two or three.
        if (wave == 0) {
            unsigned long long keptw = 0ull;                       // lane t: the survivors of tile t that were kept
            int out0 = s_ctl[1];
            for (int c = 0; c < nct; ++c) {
                const int r = c * 64 + lane;
                const bool in = r < ns;
                const unsigned long long own = in ? smask[(size_t)r * scw + c] : 0ull;
                unsigned long long hitw = 0ull;
                for (int t = 0; t < c; ++t) hitw |= (in ? smask[(size_t)r * scw + t] : 0ull) & readlane64(keptw, t);
                const unsigned long long before = own & ((1ull << lane) - 1ull);
                unsigned long long open = __ballot(in && hitw == 0ull), kept = 0ull;
                while (open) {
                    const bool mine = (open >> lane) & 1ull;
                    const bool out = (before & kept) != 0ull, wait = (before & open) != 0ull;
                    const unsigned long long nk_ = __ballot(mine && !out && !wait), no_ = __ballot(mine && out);
                    kept |= nk_;
                    open &= ~(nk_ | no_);
                }
                if (lane == c) keptw = kept;
                if ((kept >> lane) & 1ull) {
                    const int pos = out0 + __popcll(kept & ((1ull << lane) - 1ull));
                    const float4 kbx = sv_box[r];
                    if (pos < KBL) {                               // the rest of its output row is written after the last round
                        kb[pos] = kbx; ka[pos] = box_area(kbx); kj[pos] = sv_j[r];
                        if (pos < a.max_det) a.out_boxes[(long)b * a.max_det + pos] = kbx;       // (the LDS list does not stay in keep order)
                    }
                    else if (pos < a.max_det) k3_emit(a, b, pos, kbx, (int)(uint32_t)keys[sv_j[r]], score, bw, wprefix);
                    if (pos >= a.max_det) atomicOr(a.flags, 1);
                }
                out0 += __popcll(kept);
            }
            if (lane == 0) s_ctl[1] = out0;
        }
        __syncthreads();
        K3_STAMP(5);
    }
    // the output rows of the kept boxes (the walk only noted which candidates they are: the score / class / bitmap reads of a
    // row would otherwise sit in the one-wave loop, a memory round trip per 64 survivors)
    {
        int n = s_ctl[1] < a.max_det ? s_ctl[1] : a.max_det;
        n = n < KBL ? n : KBL;
        for (int pos = tid; pos < n; pos += K3_THREADS) k3_emit<false>(a, b, pos, zero4, (int)(uint32_t)keys[kj[pos]], score, bw, wprefix);
    }
    if (tid == 0) {
        const int n = s_ctl[1];
        a.out_counts[b] = n < a.max_det ? n : a.max_det;
#ifdef RVA_K3_STAMPS
        if (b < 64) for (int q = 0; q < 8; ++q) g_k3_stamps[b][q] = st_acc[q];
#endif
    }
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

    int kcap = K3_THREADS;
    while (kcap < A && kcap < 16384) kcap <<= 1;
    const int nwords = rva_ceil_div(A, 32);
    // LDS: the fixed arrays + the larger of the kernel's two layouts (see k3_nms: chosen per image from its candidate count)
    auto layout = [](size_t keys, size_t kbl, size_t sc) { return keys * 8 + (kbl + 8) * 20 + kbl * 4 + sc * 20 + sc * (sc / 64) * 8; };
    const size_t fixed = (((size_t)nwords * 4 + 15) & ~(size_t)15) + 64 + 128 + 64 + (K3_BINS + 4) * 4 + K3_BINS * 4;
    size_t smem = fixed + layout(kcap < 8192 ? kcap : 8192, K3_KBL, 512);
    if (kcap > 8192 && fixed + layout(kcap, K3_KBL / 2, 256) > smem) smem = fixed + layout(kcap, K3_KBL / 2, 256);
    if (smem > 160 * 1024) return rva_fail(ctx, RVA_ERR_CAPACITY, "post-process: %d anchors need %zu B of LDS", A, smem);
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
        k2.bits = ctx->cand_bits; k2.nwords = nwords;
        k2.irr = ctx->post_flags + 1;
        dim3 g2(rva_ceil_div(A, 256), nb);
        if (raw_dtype == RVA_F16) k2_decode<__half><<<g2, 256, 0, stream>>>(k2);
        else k2_decode<float><<<g2, 256, 0, stream>>>(k2);

        K3Args k3{};
        k3.sp_box = (const float4 *)ctx->sp_box; k3.sp_score = ctx->sp_score; k3.sp_cls = ctx->sp_cls;
        k3.bits = ctx->cand_bits;
        k3.nwords = nwords; k3.A = A; k3.kcap = kcap;
        k3.iou_thr = (float)iou_thr;
        {   // where rounding leaves thr: midpoint to the next float above, a tie goes to the even mantissa
            const float t = (float)iou_thr;
            uint32_t tb;
            std::memcpy(&tb, &t, 4);
            k3.sup.m = ((double)t + (double)std::nextafterf(t, INFINITY)) * 0.5;
            k3.sup.tie_incl = (tb & 1u) == 0;
            k3.sup.thr = t; k3.sup.thr_up = std::nextafterf(t, INFINITY);
        }
        k3.max_det = max_det;
        k3.out_boxes = (float4 *)out_boxes + (size_t)b0 * max_det;
        k3.out_scores = out_scores + (size_t)b0 * max_det;
        k3.out_cls = out_cls + (size_t)b0 * max_det;
        k3.out_anchor = out_anchor ? out_anchor + (size_t)b0 * max_det : nullptr;
        k3.out_cand = out_cand ? out_cand + (size_t)b0 * max_det : nullptr;
        k3.out_counts = out_counts + b0;
        k3.out_ncand = out_ncand ? out_ncand + b0 : nullptr;
        k3.flags = ctx->post_flags;
        k3.irr = ctx->post_flags + 1;
        k3.bitonic_upto = getenv("RVA_K3_BITONIC_UPTO") ? atoi(getenv("RVA_K3_BITONIC_UPTO")) : 0;
        {   // centre-bin filter (see k3_nms): the distance factor for this threshold, 1 % on top
            const double t = (double)(float)iou_thr * (1.0 - 1e-6);
            k3.rfac = std::isfinite(t) && t >= K3_BIN_MIN_THR && t <= 0.999 && !getenv("RVA_K3_NOBINS")
                          ? (float)((t >= 0.5 ? 1.0 - t : 0.5 / t - 0.5) * 1.01) : 0.f;
            for (int i = 0; i < nb; ++i) {
                const rva_letterbox &m = metas[n_metas == 1 ? 0 : b0 + i];
                k3.bin_scale[i] = m.src_w > 0 ? (float)K3_BINS / (float)m.src_w : 1.f;
            }
        }
        k3_nms<<<nb, K3_THREADS, smem, stream>>>(k3);
    }
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

extern "C" int rva_post_filter_stats(rva_ctx *ctx, rva_stream_t stream_, int *binned_images)
{
    if (!ctx || !binned_images) return RVA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int32_t v = 0;
    RVA_HIP(ctx, hipMemcpyAsync(&v, ctx->post_flags + 1 + RVA_MAX_BATCH, sizeof v, hipMemcpyDeviceToHost, stream));
    RVA_HIP(ctx, hipMemsetAsync(ctx->post_flags + 1 + RVA_MAX_BATCH, 0, sizeof v, stream));
    RVA_HIP(ctx, hipStreamSynchronize(stream));
    *binned_images = v;
    return RVA_OK;
}

#ifdef RVA_K3_STAMPS
extern "C" int rva_dbg_k3_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k3_stamps), sizeof(unsigned long long) * 64 * 8); }
#endif

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

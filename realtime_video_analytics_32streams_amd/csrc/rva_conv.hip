// Fused detector primitives (NHWC fp16) for the YOLOv8 network of the hot path.
//
// Why these exist: profiling the PyTorch/MIOpen network (profiles/r01_a_*) shows four kernels per
// convolution (zero-fill for split-K atomics, implicit GEMM, broadcast bias add, SiLU) plus chunk /
// concat copies; only ~40 % of the detector time is MFMA work and most layers are HBM-bound at
// YOLOv8 channel widths.  The kernels here do one pass per layer.
//
// GEMM view shared by every convolution kernel: D[co][px] = sum_k W[co][k] * X[px][k], k = tap*Cin + c, on
// v_mfma_f32_16x16x32_f16 with A = weights (rows = output channels) and B = activations (columns = pixels), so a
// lane's 4 accumulator registers are 4 consecutive output channels of one pixel.  Epilogue: + bias (fp32) -> SiLU
// -> (+ residual) -> fp16, transposed through LDS so that every pixel row leaves as full 16-byte vectors.  Input,
// output and residual take a row stride (ld*) and start at a channel offset, so producers write straight into
// concat buffers and consumers read channel slices: no cat / chunk / contiguous copies exist.
//
// Kernel families (rva_conv2d_nhwc_f16_v picks one by variant; the plan autotunes per layer):
//   k_conv_mfma    gather implicit GEMM, 4 waves, operands staged through registers into swizzled LDS rows
//   k_conv_res     persistent resident-chunk variant of the same
//   k_conv3_row    3x3 stride 1: one staged run of raster pixels serves the three horizontal taps
//   k_conv3_big    the same idea at 128-384 px x 64-128 ch per block, 8 waves, operands by LDS-DMA into a 2- or 3-slot
//                  ring (counted vmcnt, raw s_barrier)
//   k_conv_gbig    LDS-DMA gather for 1x1 and strided 3x3 (64- or 32-channel K-steps), optionally with the
//                  "upsample2x + concat" source folded in (rva_conv1x1_upcat_f16)
//   k_conv3_patch  resident weights + input patch staged once per output tile, for the small-channel layers
//   k_stem         3x3 stride-2 conv on the planar fp16 tensor K1 writes (Cin = 3), bias + SiLU, NHWC out
//   k_sppf_pool3 / k_maxpool5 / k_upsample2 / k_head3 / k_head   SPPF pooling, FPN upsample, DFL + dist2bbox + sigmoid
//
// Numerics: fp16 operands, fp32 accumulation and epilogue, one rounding to fp16 per layer output
// (PyTorch rounds after conv, after bias and after SiLU); the engine is therefore checked against the
// torch module within an fp16 tolerance, never bit-for-bit.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "rva_internal.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
// NOTE: staging registers use this native vector, not HIP's uint4 struct: assigning a dereferenced
// `const uint4*` into an array element lowers to a memcpy that SROA does not split, the array then lives in
// scratch and every prefetch load is followed by vmcnt(0) + scratch_store (serialised loads).
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const __half *in; int ldi;
    const __half *w;          // [CoutPad][taps][Cin]
    const float *bias;        // [CoutPad]
    __half *out; int ldo;
    const __half *res; int ldr;
    int CoutPad;
    // k_conv_gbig<..., HEAD>: the detect head's last 1x1 convolutions decode straight into out[B, 4+nc, A]
    __half *hout; int hmode, hnc, hA, ha0, hW, hHW; float hstride;   // hmode 1 = box branch (DFL + dist2bbox), 2 = class branch (sigmoid)
    const __half *in2; int ldi2, c_split;   // k_conv_gbig<..., UP>: `in` = low-res tensor (nearest 2x upsampled on the fly,
                                            // channels [0, c_split)), `in2` = full-res tensor (channels [c_split, Cin))
    int H, W, Cin, CinPad, Ho, Wo, Cout, stride, M, act, n_tiles, m_tiles;   // CinPad = Cin rounded up to 32 (weight rows are zero-padded)
};

// Timing-only ablation switches (skip the MFMA phase / the prefetch loads / the epilogue) exist in the private diagnostic
// build only (tools/row_stamps.py compiles with -DRVA_ROW_STAMPS); the shipped library has neither the fields nor the reads.
#ifdef RVA_ROW_STAMPS
#define RVA_DBG_FIELD int dbg;
#define RVA_DBG(a, bit) ((a).dbg & (bit))
#else
#define RVA_DBG_FIELD
#define RVA_DBG(a, bit) 0
#endif
#ifdef RVA_ROW_STAMPS
// tuning aid (tools/row_stamps.py builds a private library with this macro): per-phase s_memtime stamps of a few blocks
__device__ unsigned long long g_stamps[8][256];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP(k) do { if (st_on && st_n < 256) g_stamps[st_slot][st_n++] = stamp_now(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

constexpr int LDSROW = 40;  // halfs per staged row, padded layout (stem, resident kernel, K-step-64 rows use BK+8)

// Swizzled layout for 32-channel (64-byte) rows, no padding: the 16-byte chunk c of row r lives at physical chunk
// (c + 2*(r>>2)) & 3.  ds_read_b128 serves a wave in groups of 16 lanes that mix rows {0-3,12-15} at chunk c with
// rows {4-11} at chunk c+1 (MI355X_MICROARCH.md, LDS table); with 80-byte padded rows 3 of the 16 lanes of every
// group collide (PMC: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE on the 3x3 kernels).  For this mapping the four
// rows of a group that share (r & 3) get physical chunks {2q, 2q+2, 2q+3, 2q+1} mod 4 -- distinct for any row base --
// and the staging ds_write_b128 (4 lanes per row, consecutive rows) alternates 16-bank halves: both conflict-free.
__device__ __forceinline__ int swz32(int row, int chunk) { return row * 32 + (((chunk + 2 * (row >> 2)) & 3) << 3); }

// q / d for 0 <= q < 2^24 and d > 0, with inv = 1.0f / d: a float multiply and a one-step correction (~8 instructions) instead
// of the ~40-instruction 32-bit division.  The LDS-DMA kernels turn raster positions into image coordinates once per lane
// before their main loop, and that start-up arithmetic is not free: with the per-fragment tap masks compiled out (and with
// them their divisions) the 3x3 kernels ran 12-20 % faster (profiles/r04_experiments_not_kept.txt #2) -- the masks are now
// computed with div_s, AFTER the first LDS-DMA burst has been issued, under its latency.
__device__ __forceinline__ int div_s(int q, int d, float inv)
{
    int t = (int)((float)q * inv);
    const int r = q - t * d;
    if (r < 0) --t;
    if (r >= d) ++t;
    return t;
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp): __fdividef / operator/ expand to the ~12-instruction IEEE
// division sequence here, which dominated the epilogues (64-128 SiLUs per thread per tile)
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int BN, int WPX, int KS, int BK>
__global__ void __launch_bounds__(256) k_conv_mfma(ConvArgs a)
{
    constexpr int BM = 4 * WPX;
    constexpr int PARTS = BK / 8;                 // 16-byte chunks per staged row
    constexpr bool SWZ = BK == 32;                // 64-byte rows: swizzled, unpadded (see swz32)
    constexpr int ROWH = SWZ ? 32 : BK + 8;       // halfs per LDS row
    constexpr int NA = BM * PARTS / 256;          // activation chunks per thread per step
    constexpr int NW = BN * PARTS / 256;          // weight chunks per thread per step
    constexpr int TAPS = KS * KS;
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *actT = (__half *)smem;                         // [2][BM][ROWH]
    __half *wT = actT + 2 * BM * ROWH;                     // [2][BN][ROWH]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware tile order: blocks id and id+8 share an XCD (round-robin dispatch), so the n-tiles of
    // one m-tile are issued 8 apart and re-read their activations from the same L2 (speed only).
    int m_tile, n_tile;
    {
        const int id = blockIdx.x;
        if (a.n_tiles > 1 && (a.m_tiles & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            n_tile = slot % a.n_tiles;
            m_tile = (slot / a.n_tiles) * 8 + xcd;
        } else {
            n_tile = id % a.n_tiles;
            m_tile = id / a.n_tiles;
        }
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int HoWo = a.Ho * a.Wo;

    // per-thread source rows (fixed across K-steps)
    int pix0[NA], iy0[NA], ix0[NA];
    const int part = tid % PARTS;
    constexpr int RPT = 256 / PARTS;              // rows covered per pass of the 256 threads
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + tid / PARTS + RPT * i;
        if (m < a.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            iy0[i] = oy * a.stride - PAD;
            ix0[i] = ox * a.stride - PAD;
            pix0[i] = (b * a.H + iy0[i]) * a.W + ix0[i];
        } else {
            iy0[i] = -100000; ix0[i] = -100000; pix0[i] = 0;
        }
    }
    const int cpt = a.CinPad / BK;       // BK-channel chunks per tap
    const int nsteps = TAPS * cpt;
    const size_t wrow = (size_t)TAPS * a.CinPad;

    u4 ra[NA], rw[NW];
    auto gload = [&](int tap, int cc) {
        const int dy = KS == 1 ? 0 : tap / KS, dx = KS == 1 ? 0 : tap - dy * KS;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && cc * BK + part * 8 < a.Cin;
            ra[i] = u4{0u, 0u, 0u, 0u};
            if (ok) {
                const __half *p = a.in + (size_t)(pix0[i] + dy * a.W + dx) * a.ldi + cc * BK + part * 8;
                ra[i] = *reinterpret_cast<const u4 *>(p);
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int co = n0 + tid / PARTS + RPT * i;
            const __half *p = a.w + (size_t)co * wrow + (size_t)tap * a.CinPad + cc * BK + part * 8;
            rw[i] = *reinterpret_cast<const u4 *>(p);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *reinterpret_cast<u4 *>(actT + (size_t)buf * BM * ROWH + (SWZ ? swz32(tid / PARTS + RPT * i, part) : (tid / PARTS + RPT * i) * ROWH + part * 8)) = ra[i];
#pragma unroll
        for (int i = 0; i < NW; ++i)
            *reinterpret_cast<u4 *>(wT + (size_t)buf * BN * ROWH + (SWZ ? swz32(tid / PARTS + RPT * i, part) : (tid / PARTS + RPT * i) * ROWH + part * 8)) = rw[i];
    };

    f4 acc[BN / 16][WPX / 16];
#pragma unroll
    for (int i = 0; i < BN / 16; ++i)
#pragma unroll
        for (int j = 0; j < WPX / 16; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int tap = 0, cc = 0;
    gload(0, 0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        int ntap = tap, ncc = cc + 1;
        if (ncc == cpt) { ncc = 0; ++ntap; }
        const bool more = s + 1 < nsteps;
        if (more) gload(ntap, ncc);                      // in flight under the MFMAs below
        const __half *ab0 = actT + (size_t)buf * BM * ROWH, *wb0 = wT + (size_t)buf * BN * ROWH;
        const int arow = wv * WPX + (lane & 15), ch = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            h8 bf[WPX / 16];
#pragma unroll
            for (int j = 0; j < WPX / 16; ++j)
                bf[j] = *reinterpret_cast<const h8 *>(ab0 + (SWZ ? swz32(arow + j * 16, ch) : (arow + j * 16) * ROWH + ch * 8 + ks * 32));
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) {
                const h8 af = *reinterpret_cast<const h8 *>(wb0 + (SWZ ? swz32(i * 16 + (lane & 15), ch) : (i * 16 + (lane & 15)) * ROWH + ch * 8 + ks * 32));
#pragma unroll
                for (int j = 0; j < WPX / 16; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        tap = ntap; cc = ncc;
    }

    // epilogue: bias + SiLU in fp32, transpose through LDS, full-vector stores (+ residual)
    constexpr int SROW = BN + 8;                          // halfs per staged pixel row
    __half *stage = (__half *)smem;                       // [BM][SROW], reuses the operand buffers
#pragma unroll
    for (int i = 0; i < BN / 16; ++i) {
        const int co = 16 * i + (lane >> 4) * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(a.bias + n0 + co);
#pragma unroll
        for (int j = 0; j < WPX / 16; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wv * WPX + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;                           // 16-byte chunks per pixel row
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 256) {
        const int row = q / CPR, pc = q - row * CPR;
        const int m = m0 + row, co = n0 + pc * 8;
        if (m < a.M && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + (size_t)m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float2 x = __half22float2(vh[t]), y = __half22float2(rh[t]);
                    vh[t] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + (size_t)m * a.ldo + co) = v;
        }
    }
}

template <int BN, int WPX, int KS, int BK>
constexpr size_t conv_smem()
{
    constexpr size_t op = (size_t)2 * (4 * WPX + BN) * (BK == 32 ? 32 : BK + 8) * 2;
    constexpr size_t st = (size_t)4 * WPX * (BN + 8) * 2;
    return op > st ? op : st;
}

template <int BN, int WPX, int KS, int BK = 32>
hipError_t launch_conv(const ConvArgs &a, hipStream_t s)
{
    if (a.CinPad % BK) return hipErrorInvalidValue;
    constexpr size_t smem = conv_smem<BN, WPX, KS, BK>();
    if (hipError_t e = rva_func_smem((const void *)k_conv_mfma<BN, WPX, KS, BK>, smem); e != hipSuccess) return e;
    k_conv_mfma<BN, WPX, KS, BK><<<a.m_tiles * a.n_tiles, 256, smem, s>>>(a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, row-reuse variant.  A tile is 4*WPX CONSECUTIVE raster pixels of one image, so for
// a fixed vertical tap dy the inputs of all three horizontal taps form ONE contiguous run of BM+2 pixels
// (raster index p0 + (dy-1)*W - 1 ...).  Per K-step (dy, 32-channel chunk) that run is staged once and the
// three dx taps read their B fragments from it at row offset dx; lanes whose ox+dx-1 falls off the image
// row (raster wrap) zero their fragment.  3x fewer activation loads and barriers per MFMA than gathering per
// tap, while keeping the small tiles (high occupancy) the autotuner prefers.
struct RowArgs {
    const __half *in; int ldi;
    const __half *w; const float *bias;
    __half *out; int ldo;
    const __half *res; int ldr;
    int H, W, Cin, CinPad, Cout, act, n_tiles, tiles_per_img;
};

template <int BN, int WPX, bool PF2>
__global__ void __launch_bounds__(256) k_conv3_row(RowArgs a)
{
    constexpr int BM = 4 * WPX;
    constexpr int AROWS = BM + 2;
    constexpr int NA = (AROWS * 4 + 255) / 256;          // activation chunks per thread per step
    constexpr int NW = BN * 12 / 256;                      // weight chunks per thread per step (3 taps x 4 parts)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int AROWS_P = (AROWS + 3) & ~3;              // keep every buffer 4-row aligned for the swizzle
    __half *actT = (__half *)smem;                         // [2][AROWS_P][32]  (swizzled 64-byte rows)
    __half *wT = actT + 2 * AROWS_P * 32;                  // [2][3 dx][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_tile = blockIdx.x % a.n_tiles;
    const int t = blockIdx.x / a.n_tiles;
    const int b = t / a.tiles_per_img, p0 = (t - b * a.tiles_per_img) * BM;
    const int HW = a.H * a.W, n0 = n_tile * BN;
    const int cpt = a.CinPad >> 5;
    const int nsteps = 3 * cpt;
    const size_t wrow = (size_t)9 * a.CinPad;
    const __half *img = a.in + (size_t)b * HW * a.ldi;

    u4 ra[NA], rw[NW], rb[NA], rx[NW];       // second set only used when PF2 (two K-steps of loads in flight)
    auto gload_to = [&](u4 (&ra)[NA], u4 (&rw)[NW], int dy, int cc) {
        const int q0 = p0 + (dy - 1) * a.W - 1;            // raster index of tile row 0
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i, r = c >> 2, part = c & 3;
            const int q = q0 + r;
            ra[i] = u4{0u, 0u, 0u, 0u};
            if (r < AROWS && (unsigned)q < (unsigned)HW && (cc << 5) + part * 8 < a.Cin)
                ra[i] = *reinterpret_cast<const u4 *>(img + (size_t)q * a.ldi + (cc << 5) + part * 8);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + 256 * i, part = c & 3, rowi = c >> 2, dx = rowi % 3, co = rowi / 3;
            rw[i] = *reinterpret_cast<const u4 *>(a.w + (size_t)(n0 + co) * wrow + (size_t)(dy * 3 + dx) * a.CinPad + (cc << 5) + part * 8);
        }
    };
    auto lstore_from = [&](const u4 (&ra)[NA], const u4 (&rw)[NW], int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i, r = c >> 2, part = c & 3;
            if (r < AROWS) *reinterpret_cast<u4 *>(actT + (size_t)buf * AROWS_P * 32 + swz32(r, part)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + 256 * i, part = c & 3, rowi = c >> 2, dxw = rowi % 3, cow = rowi / 3;
            *reinterpret_cast<u4 *>(wT + (size_t)buf * BN * 96 + swz32(dxw * BN + cow, part)) = rw[i];
        }
    };

    // horizontal validity of this lane's pixels for dx = 0 (needs ox >= 1) and dx = 2 (needs ox <= W-2)
    bool okl[WPX / 16], okr[WPX / 16];
#pragma unroll
    for (int j = 0; j < WPX / 16; ++j) {
        const int p = min(p0 + wv * WPX + 16 * j + (lane & 15), HW - 1);
        const int ox = p % a.W;
        okl[j] = ox >= 1;
        okr[j] = ox <= a.W - 2;
    }

    f4 acc[BN / 16][WPX / 16];
#pragma unroll
    for (int i = 0; i < BN / 16; ++i)
#pragma unroll
        for (int j = 0; j < WPX / 16; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    auto compute = [&](int buf) {
        const __half *ab = actT + (size_t)buf * AROWS_P * 32;
        const __half *wb = wT + (size_t)buf * BN * 96;
        const int arow = wv * WPX + (lane & 15), ch = lane >> 4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            h8 bf[WPX / 16];
#pragma unroll
            for (int j = 0; j < WPX / 16; ++j) {
                bf[j] = *reinterpret_cast<const h8 *>(ab + swz32(arow + j * 16 + dx, ch));
                if (dx == 0 && !okl[j]) bf[j] = hz;
                if (dx == 2 && !okr[j]) bf[j] = hz;
            }
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) {
                const h8 af = *reinterpret_cast<const h8 *>(wb + swz32(dx * BN + i * 16 + (lane & 15), ch));
#pragma unroll
                for (int j = 0; j < WPX / 16; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto step_of = [&](int s, int &dy, int &cc) { dy = s / cpt; cc = s - dy * cpt; };
    int dy, cc;
#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 8;
    const bool st_on = tid == 0 && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 8;
    const int st_slot = st_on ? blockIdx.x / st_stride : 0;
    int st_n = 0;
#endif
    if (!PF2) {
        STAMP(0);
        gload_to(ra, rw, 0, 0);
        lstore_from(ra, rw, 0);
        __syncthreads();
        STAMP(1);
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) { step_of(s + 1, dy, cc); gload_to(ra, rw, dy, cc); }
            STAMP(2);
            compute(s & 1);
            STAMP(3);
            if (more) lstore_from(ra, rw, (s + 1) & 1);
            STAMP(4);
            __syncthreads();
            STAMP(5);
        }
    } else {
        // two K-steps of global loads in flight: set A carries the even steps' successors, set B the odd ones
        gload_to(ra, rw, 0, 0);
        lstore_from(ra, rw, 0);
        if (nsteps > 1) { step_of(1, dy, cc); gload_to(ra, rw, dy, cc); }      // step 1 -> set A
        __syncthreads();
        for (int s = 0; s < nsteps; s += 2) {
            // even step s: prefetch s+2 into set B, compute s, store set A (= step s+1)
            if (s + 2 < nsteps) { step_of(s + 2, dy, cc); gload_to(rb, rx, dy, cc); }
            compute(0);
            if (s + 1 < nsteps) lstore_from(ra, rw, 1);
            __syncthreads();
            if (s + 1 >= nsteps) break;
            // odd step s+1: prefetch s+3 into set A, compute s+1, store set B (= step s+2)
            if (s + 3 < nsteps) { step_of(s + 3, dy, cc); gload_to(ra, rw, dy, cc); }
            compute(1);
            if (s + 2 < nsteps) lstore_from(rb, rx, 0);
            __syncthreads();
        }
    }

    constexpr int SROW = BN + 8;
    __half *stage = (__half *)smem;
#pragma unroll
    for (int i = 0; i < BN / 16; ++i) {
        const int co = 16 * i + (lane >> 4) * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(a.bias + n0 + co);
#pragma unroll
        for (int j = 0; j < WPX / 16; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wv * WPX + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
    const size_t mbase = (size_t)b * HW + p0;
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 256) {
        const int row = q / CPR, pc = q - row * CPR;
        const int co = n0 + pc * 8;
        if (p0 + row < HW && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            const size_t m = mbase + row;
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + m * a.ldo + co) = v;
        }
    }
    STAMP(6);
}

template <int BN, int WPX, bool PF2 = false>
hipError_t launch_row(RowArgs &a, int batch, hipStream_t s)
{
    constexpr int BM = 4 * WPX;
    constexpr size_t op = (size_t)2 * ((((BM + 2) + 3) & ~3) + BN * 3) * 32 * 2;
    constexpr size_t st = (size_t)BM * (BN + 8) * 2;
    constexpr size_t smem = op > st ? op : st;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_row<BN, WPX, PF2>, smem); e != hipSuccess) return e;
    a.tiles_per_img = rva_ceil_div(a.H * a.W, BM);
    k_conv3_row<BN, WPX, PF2><<<batch * a.tiles_per_img * a.n_tiles, 256, smem, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Resident-chunk convolution (3x3 stride 1, and 1x1): persistent blocks, one per CU.
//
// A work item is (tile, channel chunk).  A tile = 4*WPX consecutive output pixels (raster order; of one
// image for 3x3, of the flat batch for 1x1) x BN output channels.  For each item ALL operands sit in LDS:
//   3x3: the input rows the strip touches (+ halo row above/below, + zero column left/right) for CK = 32
//        channels, and the weights of all nine taps [9][BN][32];
//   1x1: the tile's pixels for CK = 128 channels and weights [BN][128].
// The MFMA phase of an item therefore issues no global memory instruction and has no barrier; the NEXT
// item's operands (next chunk, or the first chunk of the block's next tile) are prefetched into
// registers at the start of the phase and written to LDS at the item boundary (2 barriers per item).
// Compared with re-gathering activations per tap this moves 9x fewer bytes from L2 to LDS and hides the
// load latency behind TAPS * CK/32 MFMA steps instead of one.
struct ResArgs {
    const __half *in; int ldi;
    const __half *w; const float *bias;
    __half *out; int ldo;
    const __half *res; int ldr;
    int H, W, Cin, CinPad, Cout, act, n_tiles, tiles_per_img, total_tiles, tile_rows, M;
    RVA_DBG_FIELD   // diagnostic build: 1 = skip MFMA phase, 2 = skip prefetch loads, 4 = skip epilogue
};

template <int BN, int WPX, int KS, int CK, int NA>
__global__ void __launch_bounds__(256, 1) k_conv_res(ResArgs a)
{
    constexpr int BM = 4 * WPX;
    constexpr int TAPS = KS * KS;
    constexpr int ROW = CK + 8;               // halfs per LDS row (pad 16 B: conflict-free b128 reads)
    constexpr int PARTS = CK / 8;             // 16-byte parts per row
    constexpr int KSTEPS = CK / 32;
    constexpr int NWT = TAPS * BN * PARTS / 256;   // weight uint4 per thread per item
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W2 = a.W + 2;
    const int act_px = KS == 3 ? a.tile_rows * W2 : BM;
    __half *actT = (__half *)smem;                          // [act_px][ROW]
    __half *wT = actT + (size_t)act_px * ROW;               // [TAPS][BN][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HW = a.H * a.W;
    const int cpt = a.CinPad / CK;
    const size_t wrow = (size_t)TAPS * a.CinPad;

    struct Tile { int n0, b, p0, oy_first, npix; };
    auto decode = [&](int t) {
        Tile T;
        T.n0 = (t % a.n_tiles) * BN;
        const int m = t / a.n_tiles;
        if (KS == 3) {
            T.b = m / a.tiles_per_img;
            T.p0 = (m - T.b * a.tiles_per_img) * BM;
            T.oy_first = T.p0 / a.W;
            const int p_last = min(T.p0 + BM, HW) - 1;
            T.npix = (p_last / a.W - T.oy_first + 3) * W2;
        } else {
            T.b = 0; T.p0 = m * BM; T.oy_first = 0; T.npix = min(BM, a.M - T.p0);
        }
        return T;
    };

    u4 ra[NA], rw[NWT];
    auto prefetch_act = [&](const Tile &T, int cc) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + 256 * i;
            const int px = q / PARTS, part = q - px * PARTS;
            ra[i] = u4{0u, 0u, 0u, 0u};
            if (px < T.npix && cc * CK + part * 8 < a.Cin) {
                if (KS == 3) {
                    const int r = px / W2, c = px - r * W2;
                    const int iy = T.oy_first - 1 + r, ix = c - 1;
                    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                        ra[i] = *reinterpret_cast<const u4 *>(a.in + ((size_t)(T.b * a.H + iy) * a.W + ix) * a.ldi + cc * CK + part * 8);
                } else {
                    ra[i] = *reinterpret_cast<const u4 *>(a.in + (size_t)(T.p0 + px) * a.ldi + cc * CK + part * 8);
                }
            }
        }
    };
    auto store_act = [&](const Tile &T) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + 256 * i;
            const int px = q / PARTS, part = q - px * PARTS;
            if (px < (KS == 3 ? T.npix : BM)) *reinterpret_cast<u4 *>(actT + (size_t)px * ROW + part * 8) = ra[i];
        }
    };
    auto prefetch_w = [&](int n0, int cc) {
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int q = tid + 256 * i;                   // (tap, co, part)
            const int part = q % PARTS, rowi = q / PARTS, co = rowi % BN, tap = rowi / BN;
            rw[i] = *reinterpret_cast<const u4 *>(a.w + (size_t)(n0 + co) * wrow + (size_t)tap * a.CinPad + cc * CK + part * 8);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int q = tid + 256 * i;
            const int part = q % PARTS, rowi = q / PARTS;
            *reinterpret_cast<u4 *>(wT + (size_t)rowi * ROW + part * 8) = rw[i];
        }
    };

    int t = blockIdx.x;
    if (t >= a.total_tiles) return;
    Tile T = decode(t);
    int cc = 0;
    prefetch_act(T, 0);
    prefetch_w(T.n0, 0);
    store_act(T);
    store_w();
    __syncthreads();

    f4 acc[BN / 16][WPX / 16];
#pragma unroll
    for (int i = 0; i < BN / 16; ++i)
#pragma unroll
        for (int j = 0; j < WPX / 16; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    while (true) {
        // ---- next work item: prefetch under this item's MFMA phase
        int nt = t, ncc = cc + 1;
        if (ncc == cpt) { ncc = 0; nt = t + gridDim.x; }
        const bool has_next = nt < a.total_tiles;
        Tile TN = T;
        if (has_next && !RVA_DBG(a, 2)) {
            if (nt != t) TN = decode(nt);
            prefetch_act(TN, ncc);
            prefetch_w(TN.n0, ncc);   // unconditional: a conditional definition keeps the staging array in scratch
        }
        // ---- MFMA phase: everything from LDS
        int bidx[WPX / 16];
#pragma unroll
        for (int j = 0; j < WPX / 16; ++j) {
            const int lp = wv * WPX + 16 * j + (lane & 15);
            if (KS == 3) {
                const int p = min(T.p0 + lp, HW - 1);      // tail pixels recompute the last pixel, never stored
                const int oy = p / a.W, ox = p - oy * a.W;
                bidx[j] = ((oy - T.oy_first) * W2 + ox) * ROW + (lane >> 4) * 8;
            } else {
                bidx[j] = lp * ROW + (lane >> 4) * 8;
            }
        }
        const __half *wb = wT + (size_t)(lane & 15) * ROW + (lane >> 4) * 8;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (RVA_DBG(a, 1)) break;
            const int toff = KS == 3 ? ((tap / 3) * W2 + (tap % 3)) * ROW : 0;
#pragma unroll 1
            for (int ks = 0; ks < KSTEPS; ++ks) {
                h8 bf[WPX / 16];
#pragma unroll
                for (int j = 0; j < WPX / 16; ++j) bf[j] = *reinterpret_cast<const h8 *>(actT + bidx[j] + toff + ks * 32);
#pragma unroll
                for (int i = 0; i < BN / 16; ++i) {
                    const h8 af = *reinterpret_cast<const h8 *>(wb + (size_t)(tap * BN + i * 16) * ROW + ks * 32);
#pragma unroll
                    for (int j = 0; j < WPX / 16; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                   // every wave is done reading this item's operands
        if (cc == cpt - 1 && !RVA_DBG(a, 4)) {
            // ---- epilogue of the tile: bias + SiLU, transpose through LDS, vector stores (+ residual)
            constexpr int SROW = BN + 8;
            __half *stage = (__half *)smem;
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) {
                const int co = 16 * i + (lane >> 4) * 4;
                const float4 bv = *reinterpret_cast<const float4 *>(a.bias + T.n0 + co);
#pragma unroll
                for (int j = 0; j < WPX / 16; ++j) {
                    float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
                    if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                    acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                    const int px = wv * WPX + 16 * j + (lane & 15);
                    __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                    uint2 pk;
                    pk.x = *reinterpret_cast<uint32_t *>(&lo);
                    pk.y = *reinterpret_cast<uint32_t *>(&hi);
                    *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
                }
            }
            __syncthreads();
            constexpr int CPR = BN / 8;
            const size_t mbase = KS == 3 ? (size_t)T.b * HW + T.p0 : (size_t)T.p0;
            const int valid = KS == 3 ? min(BM, HW - T.p0) : min(BM, a.M - T.p0);
#pragma unroll 4
            for (int q = tid; q < BM * CPR; q += 256) {
                const int row = q / CPR, pc = q - row * CPR;
                const int co = T.n0 + pc * 8;
                if (row < valid && co < a.Cout) {
                    uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
                    const size_t m = mbase + row;
                    if (a.res) {
                        const uint4 r = *reinterpret_cast<const uint4 *>(a.res + m * a.ldr + co);
                        __half2 *vh = reinterpret_cast<__half2 *>(&v);
                        const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                            vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                        }
                    }
                    *reinterpret_cast<uint4 *>(a.out + m * a.ldo + co) = v;
                }
            }
            // staging area free again once every wave's LDS reads are done; the global stores stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (!has_next) break;
        store_act(TN);
        store_w();
        __syncthreads();
        T = TN; t = nt; cc = ncc;
    }
}

template <int BN, int WPX, int KS, int CK, int NA>
hipError_t launch_res(ResArgs &a, int batch, int num_cus, hipStream_t s)
{
    constexpr int BM = 4 * WPX;
    constexpr int ROW = CK + 8;
    size_t act_px;
    if (KS == 3) {
        a.tile_rows = (BM - 1 + a.W - 1) / a.W + 3;         // worst-case rows a strip touches, + halo
        act_px = (size_t)a.tile_rows * (a.W + 2);
        a.tiles_per_img = rva_ceil_div(a.H * a.W, BM);
        a.total_tiles = batch * a.tiles_per_img * a.n_tiles;
    } else {
        a.tile_rows = 0;
        act_px = BM;
        a.tiles_per_img = 0;
        a.total_tiles = rva_ceil_div(a.M, BM) * a.n_tiles;
    }
    if (act_px * (CK / 8) > (size_t)NA * 256) return hipErrorInvalidValue;        // staging registers
    const size_t op = (act_px + (size_t)KS * KS * BN) * ROW * 2;
    const size_t st = (size_t)BM * (BN + 8) * 2;
    const size_t smem = op > st ? op : st;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = rva_func_smem((const void *)k_conv_res<BN, WPX, KS, CK, NA>, smem); e != hipSuccess) return e;
    // persistent grid: one block per CU, a multiple of n_tiles so a block keeps its weight columns
    int grid = num_cus - num_cus % a.n_tiles;
    if (grid <= 0) grid = a.n_tiles;
    if (grid > a.total_tiles) grid = a.total_tiles;
    k_conv_res<BN, WPX, KS, CK, NA><<<grid, 256, smem, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Stem: 3x3 stride-2 pad-1 conv on planar [B,3,H,W] fp16 (what K1 writes), Cout <= 64, NHWC out.
struct StemArgs {
    const __half *in; const float *w; const float *bias; __half *out;   // w: [Cout][27] (c, ky, kx)
    int B, H, W, Ho, Wo, Cout, ldo;
    RVA_DBG_FIELD
};

__global__ void __launch_bounds__(256) k_stem(StemArgs a, const __half *__restrict__ gw, const float *__restrict__ gb)
{
    // Persistent blocks; a tile = 8 x 32 output pixels of one image.  Per tile: the 17 x 65 x 3 input patch
    // (prefetched into registers while the previous tile computes) is staged in LDS, each thread writes its
    // pixel's 27 taps (padded to K = 32) as one im2col row, and the 27-deep dot products run as ONE
    // v_mfma_f32_16x16x32_f16 step per 16x16 output block (weights = A operand, loaded once per block;
    // pixels = B operand).  A VALU formulation needs 864 FMAs per pixel and saturated the scalar unit.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RS = 88;                                                      // halfs per patch row: 80 used (x = 64*tx-8 ...) + 8 pad
    __half *tile = (__half *)smem;                                              // [3][17][RS]
    __half *col = (__half *)(smem + 3 * 17 * RS * 2);                           // [256][LDSROW] im2col rows / output stage
    __half *wl = col + 256 * (a.Cout + 8 > LDSROW ? a.Cout + 8 : LDSROW);       // [64][LDSROW] weights (fp16, K padded)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.Wo + 31) >> 5, tiles_y = (a.Ho + 7) >> 3;
    const int tiles_img = tiles_x * tiles_y, total = tiles_img * a.B;
    {   // packed fp16 weights [64][32] (API column order: k = 2j + kx for kx in {0,1}, 18 + j for kx = 2, j = c*3 + ky):
        // one 16-byte load per thread, once per block, scattered into the kernel's own K order
        //   k' = 2j + (kx - 1) for kx in {1,2}  (one aligned dword of the patch row),  k' = 18 + j for kx = 0
        const u4 wv4 = *reinterpret_cast<const u4 *>(gw + tid * 8);
        const unsigned short *wh = reinterpret_cast<const unsigned short *>(&wv4);
        unsigned short *wrow_l = reinterpret_cast<unsigned short *>(wl) + (size_t)(tid >> 2) * LDSROW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = (tid & 3) * 8 + e;
            int kn = k;                                           // k >= 27: zero padding stays in place
            if (k < 18) { const int j = k >> 1; kn = (k & 1) ? 2 * j : 18 + j; }        // kx = 1 -> 2j ; kx = 0 -> 18 + j
            else if (k < 27) kn = 2 * (k - 18) + 1;                                      // kx = 2 -> 2j + 1
            wrow_l[kn] = wh[e];
        }
    }
    const int nblk = (a.Cout + 15) >> 4;                   // 16-channel blocks (<= 4)
    float bv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int co = 16 * i + (lane >> 4) * 4 + u;
            bv[i][u] = co < a.Cout ? gb[co] : 0.f;
        }
    // patch of a tile: 3 planes x 17 rows x 80 halfs starting at x = 64*tx - 8 (so that every row is a run of ten aligned
    // 16-byte chunks; W % 8 == 0).  510 chunks per tile = two 16-byte loads per thread -- the earlier form issued 26
    // two-byte loads per thread and spent a third of the tile time in the issue of those loads (tools/stem_stamps.py).
    u4 pv[2];
    auto prefetch = [&](int t) {
        const int bb = t / tiles_img, r2 = t - bb * tiles_img, tyy = r2 / tiles_x, txx = r2 - tyy * tiles_x;
        const int x00 = txx * 64 - 8, iy0 = tyy * 16 - 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + 256 * k;
            const int row = q / 10, cx = q - row * 10;
            const int c = row / 17, r = row - c * 17;
            const int iy = iy0 + r, x0 = x00 + cx * 8;
            pv[k] = u4{0u, 0u, 0u, 0u};
            if (!RVA_DBG(a, 1) && q < 510 && (unsigned)iy < (unsigned)a.H && (unsigned)x0 < (unsigned)a.W)
                pv[k] = *reinterpret_cast<const u4 *>(a.in + ((size_t)(bb * 3 + c) * a.H + iy) * a.W + x0);
        }
    };
#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 8;
    const bool st_on = tid == 0 && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 8;
    const int st_slot = st_on ? blockIdx.x / st_stride : 0;
    int st_n = 0;
#endif
    int t = blockIdx.x;
    if (t < total) prefetch(t);
    for (; t < total; t += gridDim.x) {
        STAMP(0);
        const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / tiles_x, tx = r2 - ty * tiles_x;
        const int ox0 = tx * 32, oy0 = ty * 8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + 256 * k;
            const int row = q / 10, cx = q - row * 10;
            if (q < 510) *reinterpret_cast<u4 *>(tile + row * RS + cx * 8) = pv[k];
        }
        __syncthreads();
        STAMP(1);
        if (t + (int)gridDim.x < total) prefetch(t + gridDim.x);   // next tile's patch flies under this tile's work
        STAMP(2);
        {   // im2col row of this thread's pixel.  K order (the weights were scattered into it above):
            //   k' = 2*j + (kx - 1) for kx in {1,2}  (j = c*3 + ky): halfs 2*lx+8, 2*lx+9 of the patch row = one aligned dword,
            //   k' = 18 + j for kx = 0 (half 2*lx+7), k' = 27..31 zero.
            const int lx = tid & 31, ly = tid >> 5;
            uint32_t d[16];
            unsigned short sgl[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const __half *p = tile + ((j / 3) * 17 + ly * 2 + (j % 3)) * RS + lx * 2 + 8;
                d[j] = *reinterpret_cast<const uint32_t *>(p);                 // kx = 1, 2
                sgl[j] = *reinterpret_cast<const unsigned short *>(p - 1);     // kx = 0
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) d[9 + j] = (uint32_t)sgl[2 * j] | ((uint32_t)sgl[2 * j + 1] << 16);
            d[13] = sgl[8];
            d[14] = 0; d[15] = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<u4 *>(col + (size_t)tid * LDSROW + q * 8) = u4{d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
        }
        __syncthreads();
        STAMP(3);
        f4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        h8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const h8 *>(col + (size_t)(wv * 64 + 16 * j + (lane & 15)) * LDSROW + (lane >> 4) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < nblk) {
                const h8 af = *reinterpret_cast<const h8 *>(wl + (size_t)(16 * i + (lane & 15)) * LDSROW + (lane >> 4) * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
            }
        }
        STAMP(4);
        __syncthreads();                                    // im2col rows are dead: reuse them as the output stage
        const int srow = a.Cout + 8;
        __half *stage = col;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < nblk) {
                const int co = 16 * i + (lane >> 4) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int px = wv * 64 + 16 * j + (lane & 15);
                    __half2 lo = __floats2half2_rn(silu_f(acc[i][j][0] + bv[i][0]), silu_f(acc[i][j][1] + bv[i][1]));
                    __half2 hi = __floats2half2_rn(silu_f(acc[i][j][2] + bv[i][2]), silu_f(acc[i][j][3] + bv[i][3]));
                    uint2 pk;
                    pk.x = *reinterpret_cast<uint32_t *>(&lo);
                    pk.y = *reinterpret_cast<uint32_t *>(&hi);
                    if (co < a.Cout) *reinterpret_cast<uint2 *>(stage + (size_t)px * srow + co) = pk;
                }
            }
        }
        __syncthreads();
        STAMP(5);
        const int cpr = a.Cout >> 3;
        for (int q = tid; q < 256 * cpr; q += 256) {
            const int px = q / cpr, pc = q - px * cpr;
            const int ox = ox0 + (px & 31), oy = oy0 + (px >> 5);
            if (ox < a.Wo && oy < a.Ho && !RVA_DBG(a, 4))
                *reinterpret_cast<uint4 *>(a.out + ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.ldo + pc * 8) =
                    *reinterpret_cast<const uint4 *>(stage + (size_t)px * srow + pc * 8);
        }
        STAMP(6);
        // the next iteration's first barrier (after the tile fill) also orders these stage reads before the
        // next im2col writes; the tile buffer itself is not touched by the output copy
    }
}

// ---------------------------------------------------------------------------------------------------
// The three chained 5x5 max pools of SPPF in one launch.  pool5(pool5(x)) = pool9(x) and pool5^3(x) = pool13(x)
// (stride 1, -inf padding: the windows just grow), so all three outputs are separable running maxima of the same tile:
// a block holds one image x 8 channels in LDS, does the row pass for window radii 2 / 4 / 6, then the column pass,
// and writes y1, y2, y3 into their channel slices.  Replaces three dependent launches that each re-read 25 taps from L2.
__global__ void __launch_bounds__(256) k_sppf_pool3(const __half *in, int ldi, __half *o1, __half *o2, __half *o3, int ldo,
                                                    int H, int W, int C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W, cg = C >> 3;
    h8 *x = (h8 *)smem;                  // [HW]
    h8 *r5 = x + HW, *r9 = r5 + HW, *r13 = r9 + HW;
    const int b = blockIdx.x / cg, g = blockIdx.x - b * cg;
    const size_t base = (size_t)b * HW;
    for (int p = threadIdx.x; p < HW; p += 256) x[p] = *reinterpret_cast<const h8 *>(in + (base + p) * ldi + g * 8);
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int y = p / W, xx = p - y * W;
        h8 m = x[p];
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            if (xx - d >= 0) m = __builtin_elementwise_max(m, x[p - d]);
            if (xx + d < W) m = __builtin_elementwise_max(m, x[p + d]);
            if (d == 2) r5[p] = m;
            if (d == 4) r9[p] = m;
        }
        r13[p] = m;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int y = p / W;
        h8 m5 = r5[p], m9 = r9[p], m13 = r13[p];
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            if (y - d >= 0) {
                if (d <= 2) m5 = __builtin_elementwise_max(m5, r5[p - d * W]);
                if (d <= 4) m9 = __builtin_elementwise_max(m9, r9[p - d * W]);
                m13 = __builtin_elementwise_max(m13, r13[p - d * W]);
            }
            if (y + d < H) {
                if (d <= 2) m5 = __builtin_elementwise_max(m5, r5[p + d * W]);
                if (d <= 4) m9 = __builtin_elementwise_max(m9, r9[p + d * W]);
                m13 = __builtin_elementwise_max(m13, r13[p + d * W]);
            }
        }
        const size_t o = (base + p) * ldo + g * 8;
        *reinterpret_cast<h8 *>(o1 + o) = m5;
        *reinterpret_cast<h8 *>(o2 + o) = m9;
        *reinterpret_cast<h8 *>(o3 + o) = m13;
    }
}

// The same with G groups of 8 channels per block (round 4): consecutive lanes take the G 16-byte pieces of one pixel and then the
// next pixel, so a wave touches whole 64-byte (G = 4) runs of the NHWC rows instead of one 16-byte piece per 512-byte row, for the
// loads and for the three stores alike.  1024 threads; the 13-wide row maxima replace the input tile in LDS (they are held in
// registers across a barrier), so a block needs 3 x HW x G x 16 bytes.  Used when the launch still fills the chip that way.
template <int G>
__global__ void __launch_bounds__(1024) k_sppf_pool3g(const __half *in, int ldi, __half *o1, __half *o2, __half *o3, int ldo,
                                                      int H, int W, int C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MAXIT = 4;             // items per thread: HW * G <= 4096
    const int HW = H * W, n = HW * G, cgb = C / (8 * G);
    h8 *x = (h8 *)smem;                  // [HW][G]; holds the 13-wide row maxima after the row pass
    h8 *r5 = x + n, *r9 = r5 + n;
    const int b = blockIdx.x / cgb, gb = blockIdx.x - b * cgb;
    const size_t base = (size_t)b * HW;
    const int c0 = gb * 8 * G;
    for (int q = threadIdx.x; q < n; q += 1024) x[q] = *reinterpret_cast<const h8 *>(in + (base + q / G) * ldi + c0 + (q % G) * 8);
    __syncthreads();
    h8 m13[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int q = threadIdx.x + it * 1024;
        if (q < n) {
            const int p = q / G, xx = p % W;
            h8 m = x[q];
#pragma unroll
            for (int d = 1; d <= 6; ++d) {
                if (xx - d >= 0) m = __builtin_elementwise_max(m, x[q - d * G]);
                if (xx + d < W) m = __builtin_elementwise_max(m, x[q + d * G]);
                if (d == 2) r5[q] = m;
                if (d == 4) r9[q] = m;
            }
            m13[it] = m;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int q = threadIdx.x + it * 1024;
        if (q < n) x[q] = m13[it];
    }
    __syncthreads();
    const int WG = W * G;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int q = threadIdx.x + it * 1024;
        if (q < n) {
            const int p = q / G, y = p / W;
            h8 m5 = r5[q], m9 = r9[q], m = x[q];
#pragma unroll
            for (int d = 1; d <= 6; ++d) {
                if (y - d >= 0) {
                    if (d <= 2) m5 = __builtin_elementwise_max(m5, r5[q - d * WG]);
                    if (d <= 4) m9 = __builtin_elementwise_max(m9, r9[q - d * WG]);
                    m = __builtin_elementwise_max(m, x[q - d * WG]);
                }
                if (y + d < H) {
                    if (d <= 2) m5 = __builtin_elementwise_max(m5, r5[q + d * WG]);
                    if (d <= 4) m9 = __builtin_elementwise_max(m9, r9[q + d * WG]);
                    m = __builtin_elementwise_max(m, x[q + d * WG]);
                }
            }
            const size_t o = (base + p) * ldo + c0 + (q % G) * 8;
            *reinterpret_cast<h8 *>(o1 + o) = m5;
            *reinterpret_cast<h8 *>(o2 + o) = m9;
            *reinterpret_cast<h8 *>(o3 + o) = m;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 5x5 stride-1 pad-2 max pool over a channel slice (C % 8 == 0): thread = (pixel, 8 channels)
__global__ void __launch_bounds__(256) k_maxpool5(const __half *in, int ldi, __half *out, int ldo, int B, int H, int W, int C)
{
    const int cg = C >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * H * W * cg) return;
    const int g = (int)(idx % cg);
    const long pix = idx / cg;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    h8 mx;
#pragma unroll
    for (int t = 0; t < 8; ++t) mx[t] = (_Float16)-65504.f;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const h8 v = *reinterpret_cast<const h8 *>(in + ((size_t)(b * H + yy) * W + xx) * ldi + g * 8);
            mx = __builtin_elementwise_max(mx, v);
        }
    }
    *reinterpret_cast<h8 *>(out + (size_t)pix * ldo + g * 8) = mx;
}

// nearest 2x upsample: out pixel (y, x) <- in pixel (y/2, x/2); thread = (out pixel, 8 channels)
__global__ void __launch_bounds__(256) k_upsample2(const __half *in, int ldi, __half *out, int ldo, int B, int H, int W, int C)
{
    const int cg = C >> 3, Ho = H * 2, Wo = W * 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * Ho * Wo * cg) return;
    const int g = (int)(idx % cg);
    const long pix = idx / cg;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
    const uint4 v = *reinterpret_cast<const uint4 *>(in + ((size_t)(b * H + (y >> 1)) * W + (x >> 1)) * ldi + g * 8);
    *reinterpret_cast<uint4 *>(out + (size_t)pix * ldo + g * 8) = v;
}

// ---------------------------------------------------------------------------------------------------
// Detect head: box logits [B, hw, 4*16] + class logits [B, hw, nc] of one level ->
// out[B, 4+nc, A] at anchor offset a0 (xywh * stride, sigmoid scores).  Thread = anchor.
struct HeadArgs {
    const __half *box; int ldb; const __half *cls; int ldc; __half *out;
    int B, h, w, nc, A, a0; float stride;
};

struct Head3Args { HeadArgs lv[3]; int blocks[3]; };     // blocks[l] = 256-anchor blocks of level l

__device__ __forceinline__ void head_anchor(const HeadArgs &a, int i, int b);

__global__ void __launch_bounds__(256) k_head(HeadArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.h * a.w) head_anchor(a, i, blockIdx.y);
}

// the three pyramid levels in one launch: the two small levels no longer run as separate, under-filled kernels
__global__ void __launch_bounds__(256) k_head3(Head3Args a)
{
    int blk = blockIdx.x, l = 0;
    if (blk >= a.blocks[0]) { blk -= a.blocks[0]; l = 1; if (blk >= a.blocks[1]) { blk -= a.blocks[1]; l = 2; } }
    const HeadArgs &h = a.lv[l];
    const int i = blk * 256 + threadIdx.x;
    if (i < h.h * h.w) head_anchor(h, i, blockIdx.y);
}

__device__ __forceinline__ void head_anchor(const HeadArgs &a, int i, int b)
{
    const int hw = a.h * a.w;
    const size_t pix = (size_t)b * hw + i;
    const __half *bp = a.box + pix * a.ldb;
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16], mx = -1e30f;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(bp + s * 16), q1 = *reinterpret_cast<const uint4 *>(bp + s * 16 + 8);
        const __half2 *h0 = reinterpret_cast<const __half2 *>(&q0), *h1 = reinterpret_cast<const __half2 *>(&q1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float2 f = __half22float2(h0[t]); v[2 * t] = f.x; v[2 * t + 1] = f.y;
            f = __half22float2(h1[t]); v[8 + 2 * t] = f.x; v[9 + 2 * t] = f.y;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) mx = fmaxf(mx, v[t]);
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) { const float e = __expf(v[t] - mx); se += e; sw += e * (float)t; }
        d[s] = sw / se;                                  // DFL expectation
    }
    const float ax = (float)(i % a.w) + 0.5f, ay = (float)(i / a.w) + 0.5f;
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    __half *o = a.out + (size_t)b * (4 + a.nc) * a.A + a.a0 + i;
    o[0] = __float2half_rn((x1 + x2) * 0.5f * a.stride);
    o[(size_t)a.A] = __float2half_rn((y1 + y2) * 0.5f * a.stride);
    o[(size_t)2 * a.A] = __float2half_rn((x2 - x1) * a.stride);
    o[(size_t)3 * a.A] = __float2half_rn((y2 - y1) * a.stride);
    const __half *cp = a.cls + pix * a.ldc;
    for (int c = 0; c < a.nc; c += 8) {
        const uint4 q = *reinterpret_cast<const uint4 *>(cp + c);
        const __half2 *hq = reinterpret_cast<const __half2 *>(&q);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float2 f = __half22float2(hq[t]);
            o[(size_t)(4 + c + 2 * t) * a.A] = __float2half_rn(__builtin_amdgcn_rcpf(1.0f + __expf(-f.x)));
            o[(size_t)(5 + c + 2 * t) * a.A] = __float2half_rn(__builtin_amdgcn_rcpf(1.0f + __expf(-f.y)));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, large-tile LDS-DMA variant ("big").
//
// Why: s_memtime stamps of k_conv3_row (tools/row_stamps.py) show that ~45 % of every K-step is spent ISSUING the
// next step's global loads -- the CU's vector-memory path (64-byte row segments, ~30 GB/s per CU) is the limiter,
// not MFMA and not LDS.  The lever is therefore MACs per staged byte: a 256 px x 128 ch tile stages 41 KB per
// (dy, 32-channel) step for 3.1 M MACs (77 MAC/B, vs 38 for the 128 x 64 tile), at the price of one block per CU.
// At that occupancy nothing but the block's own pipeline hides memory latency, so operands are moved by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a THREE-slot ring with two steps in flight, a
// counted s_waitcnt vmcnt(N) and one raw s_barrier per step.
//
// Geometry: the whole batch is one flat raster of M = B*H*W pixels; a tile is BM consecutive raster pixels, so tiles
// cross image borders and there is no per-image tail.  For vertical tap dy the inputs of the three horizontal taps
// are the contiguous run [P0 + (dy-1)*W - 1, +BM+2); source rows are clamped into the tensor and every lane zeroes
// the B fragments whose tap falls outside its own image (9-bit validity mask per pixel), so no zero-fill is needed
// (LDS-DMA cannot write zeros).  LDS image: 1 KiB pieces of 16 rows x 64 B, lane-linear as LDS-DMA requires; the
// swz32 chunk rotation is applied on the per-lane SOURCE address.  Needs Cin % 32 == 0.
struct BigArgs {
    const __half *in; int ldi;
    const __half *w; const float *bias;
    __half *out; int ldo;
    const __half *res; int ldr;
    int H, W, Cin, Cout, CoutPad, act, n_tiles, M, m_tiles;
};

typedef __attribute__((address_space(3))) void *lds_vptr;
typedef const __attribute__((address_space(1))) void *glb_vptr;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One 1 KiB LDS-DMA piece: 64 lanes x 16 bytes from  sbase + voff  (wave-uniform 64-bit base in an SGPR pair, per-lane
// 32-bit byte offset in ONE VGPR) to LDS address m0v + 16 * lane.  Written as asm so that the address really takes the
// saddr form: through the builtin every piece cost a v_mad_i64_i32, three 64-bit VALU adds, a v_readfirstlane and the
// scalar division that recovers (tap, chunk) from the step number -- ~100 issue cycles per piece, 650-830 cycles per K-step
// of a wave (profiles/r01_conv_big.txt), during which its SIMD partner's MFMAs share the VALU port.  In this form a piece
// is two scalar moves and the load; the per-lane offsets are computed once per kernel.
__device__ __forceinline__ void lds_dma16(unsigned voff, const void *sbase, unsigned m0v)
{
    // "s" operands must really be scalar registers: inline asm does not legalise a VGPR into an SGPR operand.  readfirstlane
    // is free for values the compiler already knows to be uniform and repairs the ones it does not.
    const unsigned long long b = (unsigned long long)(size_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    const unsigned m0s = __builtin_amdgcn_readfirstlane(m0v);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(m0s) : "memory");
}
// The same piece with only the lanes of `mask` active: the other lanes neither load nor write their 16 bytes of LDS (what
// k_conv3_run<..., PADO> needs for the pad positions of its padded raster, which are zeroed once and must stay zero).  Only
// from wave-uniform control flow with all 64 lanes active (every issue site of these kernels): EXEC is restored to all ones.
__device__ __forceinline__ void lds_dma16_masked(unsigned voff, const void *sbase, unsigned m0v, unsigned long long mask)
{
    const unsigned long long b = (unsigned long long)(size_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    const unsigned m0s = __builtin_amdgcn_readfirstlane(m0v);
    const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mask), mhi = __builtin_amdgcn_readfirstlane((unsigned)(mask >> 32));
    const unsigned long long ms = ((unsigned long long)mhi << 32) | mlo;
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(sb), "s"(m0s), "s"(ms) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(lds_vptr)p; }
__device__ __forceinline__ void wait_vm_n(int n)
{
    switch (n) {   // wave-uniform
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 10: wait_vm<10>(); break;
    case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break;
    default: wait_vm<12>(); break;         // more than 12 in flight: waiting down to 12 is stricter than asked, never looser
    }
}

template <int BM, int BN, int WGM, int WGN, int NSLOT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NSLOT == 2 ? 4 : 2))) k_conv3_big(BigArgs a)
{
    static_assert(WGM * WGN == 8, "eight waves");
    static_assert(NSLOT == 2 || NSLOT == 3, "ring depth");
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
    constexpr int APIECES = (BM + 2 + 15) / 16;           // 1 KiB pieces (16 rows x 64 B) of the activation run
    constexpr int WPIECES = 3 * BN / 16;                  // weights [3 dx][BN] rows
    constexpr int PIECES = APIECES + WPIECES;
    constexpr int SLOTH = PIECES * 512;                   // halfs per ring slot
    constexpr int NP = (PIECES + 7) / 8;                  // pieces per wave per step (upper bound)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *ring = (__half *)smem;                        // [NSLOT][PIECES][16][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WGM, wn = wv / WGM;
    // XCD-aware tile order: workgroups id and id + 8 land on the same XCD (round-robin dispatch; speed only, never
    // correctness), so the n-tiles of one m-tile -- which stage the same activation rows -- are given ids 8 apart: they
    // share that XCD's L2 instead of pulling the rows across the fabric once per n-tile.  Identity when n_tiles == 1.
    const int n_tile = (blockIdx.x >> 3) % a.n_tiles, m_tile = ((blockIdx.x >> 3) / a.n_tiles) * 8 + (blockIdx.x & 7);
    if (m_tile >= a.m_tiles) return;                      // grid is padded to a multiple of 8 m-tiles
    const int P0 = m_tile * BM, n0 = n_tile * BN;
    const int HW = a.H * a.W;
    const int cpt = a.Cin >> 5;
    const int nsteps = 3 * cpt;
    const int wrow = 9 * a.Cin;

    // this wave's pieces: idx = wv + 8k.  Per-lane byte offsets of each piece, fixed for the whole kernel (see lds_dma16):
    // activation pieces: the lane's source row for dy = 0 / 1 / 2 (clamped into the tensor) + its swizzled chunk, relative to
    // a.in; weight pieces: the lane's weight row (tap dy = 0, chunk 0) + chunk, relative to a.w.  The per-step part of the
    // address -- channel chunk, vertical tap -- is wave-uniform and lives in the scalar base.
    const int lrow = lane >> 2, lp = lane & 3;
    const int my_pieces = (PIECES - wv + 7) / 8;
    unsigned poff0[NP], poff1[NP], poff2[NP];     // three 1-D arrays with compile-time indices: they must stay in registers
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = wv + 8 * k;
        if (idx < APIECES) {
            const int r = idx * 16 + lrow;
            const unsigned c = (unsigned)(((lp - 2 * (r >> 2)) & 3) * 16);
            const int q1 = P0 - 1 + r;
            poff0[k] = (unsigned)min(max(q1 - a.W, 0), a.M - 1) * (unsigned)(a.ldi * 2) + c;
            poff1[k] = (unsigned)min(max(q1, 0), a.M - 1) * (unsigned)(a.ldi * 2) + c;
            poff2[k] = (unsigned)min(max(q1 + a.W, 0), a.M - 1) * (unsigned)(a.ldi * 2) + c;
        } else {
            const int rw = (idx - APIECES) * 16 + lrow;
            const int dx = rw / BN, co = min(n0 + rw - dx * BN, a.CoutPad - 1);
            poff0[k] = (unsigned)(co * wrow + dx * a.Cin) * 2u + (unsigned)(((lp - 2 * (rw >> 2)) & 3) * 16);
            poff1[k] = poff2[k] = poff0[k];
        }
    }
    const unsigned ring0 = lds_addr(ring);
    // the step whose pieces are issued next: every wave issues the steps in order, each exactly once.  Plain scalars and
    // macros on purpose: as lambdas capturing this mutable state by reference the closure went to scratch memory.
    int is_dy = 0, is_cc = 0, is_slot = 0;
    const char *is_abase = (const char *)a.in, *is_wbase = (const char *)a.w;
    unsigned is_lds = ring0;
#define BIG_ISSUE_PIECE(k_)                                                                                   \
    do {                                                                                                      \
        const int idx_ = wv + 8 * (k_);                                                                       \
        if (idx_ < PIECES) {                                                                                  \
            const unsigned m0v_ = is_lds + (unsigned)idx_ * 1024u;                                            \
            if (idx_ < APIECES) {                                                                             \
                unsigned v_ = poff1[k_];                                                                      \
                if (is_dy == 0) v_ = poff0[k_];                                                               \
                if (is_dy == 2) v_ = poff2[k_];                                                               \
                lds_dma16(v_, is_abase, m0v_);                                                                \
            } else {                                                                                          \
                lds_dma16(poff0[k_], is_wbase, m0v_);                                                         \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#define BIG_ISSUE_ADVANCE()                                                                                   \
    do {                                                                                                      \
        ++is_cc;                                                                                              \
        if (is_cc == cpt) { is_cc = 0; ++is_dy; }                                                             \
        ++is_slot;                                                                                            \
        if (is_slot == NSLOT) is_slot = 0;                                                                    \
        is_cc = __builtin_amdgcn_readfirstlane(is_cc);       /* keep the issue state in scalar registers */   \
        is_dy = __builtin_amdgcn_readfirstlane(is_dy);                                                        \
        is_slot = __builtin_amdgcn_readfirstlane(is_slot);                                                    \
        is_abase = (const char *)a.in + is_cc * 64;                                                           \
        is_wbase = (const char *)a.w + (size_t)(is_dy * 3 * a.Cin + is_cc * 32) * 2;                         \
        is_lds = ring0 + (unsigned)is_slot * (unsigned)(SLOTH * 2);                                           \
    } while (0)
#define BIG_ISSUE()                                                                                           \
    do {                                                                                                      \
        _Pragma("unroll") for (int k_i = 0; k_i < NP; ++k_i) BIG_ISSUE_PIECE(k_i);                            \
        BIG_ISSUE_ADVANCE();                                                                                  \
    } while (0)


    f4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    // bias for the epilogue, fetched BEFORE the first LDS-DMA (so the counted vmcnt waits below never include it) and
    // long before its use: a dependent global load at the top of the epilogue costs ~1.5 k cycles per block
    float4 bvs[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i)
        bvs[i] = *reinterpret_cast<const float4 *>(a.bias + min(n0 + wn * TN + 16 * i + (lane >> 4) * 4, a.CoutPad - 4));

#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 8;
    const bool st_on = tid == 0 && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 8;
    const int st_slot = st_on ? blockIdx.x / st_stride : 0;
    int st_n = 0;
#endif
    STAMP(0);
#pragma unroll
    for (int t = 0; t < NSLOT - 1; ++t)
        if (t < nsteps) BIG_ISSUE();
    // 9-bit tap validity per pixel fragment: bit dy*3+dx set when tap (dy,dx) of that pixel lies inside its image (computed
    // here, under the latency of the first burst: see div_s)
    int vm[FM];
    {
        const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)a.W;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int p = P0 + wm * TM + 16 * j + (lane & 15);
            int m = 0;
            if (p < a.M) {
                const int rem = p - div_s(p, HW, inv_hw) * HW, oy = div_s(rem, a.W, inv_w), ox = rem - oy * a.W;
                const int hv = (ox >= 1 ? 1 : 0) | 2 | (ox <= a.W - 2 ? 4 : 0);
                m = (oy >= 1 ? hv : 0) | (hv << 3) | (oy <= a.H - 2 ? hv << 6 : 0);
            }
            vm[j] = m;
        }
    }
    STAMP(1);
    for (int s = 0; s < nsteps; ++s) {
        // retire this wave's pieces of step s (with three slots those of step s+1 stay in flight), then meet the
        // other waves: after the barrier every piece of step s is in LDS and nobody still reads the slot that the
        // pieces of step s+NSLOT-1 -- issued between the MFMAs below -- will overwrite
        wait_vm_n(NSLOT == 3 && s + 1 < nsteps ? my_pieces : 0);
        STAMP(2);
        __builtin_amdgcn_s_barrier();
        STAMP(3);
        const int sn = s + NSLOT - 1;
        const bool more = sn < nsteps;
        // three slots (one block per CU, two waves per SIMD): waves w and w+4 share a SIMD, so the lower four issue
        // their DMA burst before the MFMAs and the upper four after them -- while one wave of a SIMD is busy issuing
        // (~750 cycles), the other one keeps the matrix core fed
        const bool early = wv < 4;
        // two slots: the whole burst goes out right after the barrier too (every wave has left the slot it overwrites); with
        // the pieces interleaved between the MFMA groups the last one was issued two thirds into the step and its ~900-cycle
        // flight showed up as ~430 cycles of vmcnt wait per step
        if (more && (NSLOT == 2 || early)) BIG_ISSUE();
        STAMP(4);
        const int dy = s / cpt;
        const __half *ab = ring + (size_t)(s % NSLOT) * SLOTH;
        const __half *wb = ab + APIECES * 512;
        const int arow = wm * TM + (lane & 15), ch = lane >> 4;
        const int vsh = dy * 3;
        // fragment registers are double-buffered across the three horizontal taps: the ds_reads of tap dx+1 are in
        // flight under the MFMAs of tap dx (otherwise every tap exposes two LDS round trips, ~900 cycles per step)
        constexpr int DB = NSLOT == 3 ? 1 : 0;          // the two-slot variants live within 128 VGPRs: single-buffered
        h8 bf[DB + 1][FM], af[DB + 1][FN];
        auto lfrag = [&](int dx, h8 (&b)[FM], h8 (&w)[FN]) {
#pragma unroll
            for (int j = 0; j < FM; ++j) b[j] = *reinterpret_cast<const h8 *>(ab + swz32(arow + j * 16 + dx, ch));
#pragma unroll
            for (int i = 0; i < FN; ++i) w[i] = *reinterpret_cast<const h8 *>(wb + swz32(dx * BN + wn * TN + i * 16 + (lane & 15), ch));
        };
        if (DB) {
            lfrag(0, bf[0], af[0]);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (dx < 2) lfrag(dx + 1, bf[(dx + 1) & DB], af[(dx + 1) & DB]);
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    if (!((vm[j] >> (vsh + dx)) & 1)) bf[dx & DB][j] = hz;
#pragma unroll
                for (int i = 0; i < FN; ++i) {
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[dx & DB][i], bf[dx & DB][j], acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    bf[0][j] = *reinterpret_cast<const h8 *>(ab + swz32(arow + j * 16 + dx, ch));
                    if (!((vm[j] >> (vsh + dx)) & 1)) bf[0][j] = hz;
                }
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const h8 a1 = *reinterpret_cast<const h8 *>(wb + swz32(dx * BN + wn * TN + i * 16 + (lane & 15), ch));
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf[0][j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (NSLOT == 3 && more && !early) BIG_ISSUE();
        STAMP(5);
    }
    __syncthreads();     // all waves done with the ring: reuse it as the output staging tile

    constexpr int SROW = BN + 8;
    __half *stage = (__half *)smem;                       // [BM][SROW]
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int co = wn * TN + 16 * i + (lane >> 4) * 4;
        const float4 bv = bvs[i];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wm * TM + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 512) {
        const int row = q / CPR, pc = q - row * CPR;
        const int co = n0 + pc * 8;
        const int m = P0 + row;
        if (m < a.M && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + (size_t)m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + (size_t)m * a.ldo + co) = v;
        }
    }
    STAMP(6);
}

#undef BIG_ISSUE
#undef BIG_ISSUE_ADVANCE
#undef BIG_ISSUE_PIECE

template <int BM, int BN, int WGM, int WGN, int NSLOT>
hipError_t launch_big(BigArgs &a, hipStream_t s)
{
    constexpr size_t ring = (size_t)NSLOT * ((BM + 2 + 15) / 16 + 3 * BN / 16) * 1024;
    constexpr size_t st = (size_t)BM * (BN + 8) * 2;
    constexpr size_t smem = ring > st ? ring : st;
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (a.Cin % 32 || a.CoutPad % 4 || a.M >= (1 << 24)) return hipErrorInvalidValue;     // (div_s: raster positions below 2^24)
    // LDS-DMA addresses are a 64-bit scalar base + a 32-bit per-lane byte offset
    if ((size_t)a.M * a.ldi * 2 >= (1ull << 32) || (size_t)a.CoutPad * 9 * a.Cin * 2 >= (1ull << 32)) return hipErrorInvalidValue;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_big<BM, BN, WGM, WGN, NSLOT>, smem); e != hipSuccess) return e;
    a.n_tiles = rva_ceil_div(a.Cout, BN);
    a.m_tiles = rva_ceil_div(a.M, BM);
    k_conv3_big<BM, BN, WGM, WGN, NSLOT><<<rva_ceil_div(a.m_tiles, 8) * 8 * a.n_tiles, 512, smem, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, "long run" LDS-DMA variant.
//
// k_conv3_big stages the activation run of a tile once per (vertical tap, channel chunk): every row travels L2 -> LDS three
// times.  s_memtime stamps after the address rewrite (profiles/r02_conv_stamps.txt) show what is left of a K-step: the
// 24-48 MFMAs of the wave (MFMA-bound while they run, ~62 % of the step) and 0.6-0.7 k cycles of load wait + barrier --
// the per-CU vector-memory path moves 29-41 KB per step and block and is the co-limiter.  Here a 32-channel chunk's run is
// staged ONCE for all three vertical taps: rows [P0 - W - 1, P0 + BM + W + 1) of the flat pixel raster (BM + 2W + 2 rows
// instead of 3 x (BM + 2)), double buffered across chunks; only the weights of a (chunk, dy) step stream per step.
// 128 -> 128 at 40x40 with a 256 x 64 tile: 233 instead of 345 KB per tile (-32 %), and 68 KB of LDS, so two blocks per
// CU overlap each other's epilogues.  K order: chunk-major, then dy, then dx (fp32 accumulation, one rounding at the end
// like every other variant).  Tap validity, swizzle, epilogue as in k_conv3_big.  Needs Cin % 32 == 0 and W <= 160.
struct RunArgs {
    const __half *in; int ldi;
    const __half *w; const float *bias;
    __half *out; int ldo;
    const __half *res; int ldr;
    int H, W, Cin, Cout, CoutPad, act, n_tiles, M, m_tiles, apieces, map_off;
};

// NOSEL: timing-only experiment (private build, -DRVA_EXPERIMENTS: profiles/r04_experiments_not_kept.txt) -- the padding selects
// compiled out, i.e. what a zero-halo activation layout could save at most; border pixels are then wrong.
//
// PADO (round 4): the padding selects leave the MFMA phase by making the tile a run of PADDED positions.  The raster the
// kernel walks has row pitch W + 1 (one pad position after every image row: right pad of row y, left pad of row y + 1) and one
// pad row after every image; a tile is BM consecutive positions of it, the LDS run holds positions [P0 - (W+1) - 1,
// P0 + BM + (W+1) + 1), and every tap of every position is the position at a FIXED offset -- a real pixel or a pad position
// that holds zeros.  The tensor in HBM keeps its layout: the lanes of an LDS-DMA piece whose run rows are real pixels fetch
// those pixels, the lanes whose rows are pad positions are masked out of the piece (EXEC) and their 16 bytes, zeroed once
// before the first step, are never written again.  The price: the MFMAs also run for the pad positions among the tile's
// outputs (1 / W + 1 / H of the work: 5 % at 40 x 40, 2.5 % at 80 x 80, 10 % at 20 x 20), whose rows the epilogue skips (a
// per-row pixel index in LDS).
template <int BM, int BN, int WGM, int WGN, bool NOSEL = false, bool PADO = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BN <= 64 && BM <= 320 ? 4 : 2))) k_conv3_run(RunArgs a)
{
    static_assert(WGM * WGN == 8, "eight waves");
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
    constexpr int WPIECES = 3 * BN / 16;                  // weights [3 dx][BN] rows of one (chunk, dy) step
    constexpr int NPW = (WPIECES + 7) / 8;
    constexpr int MAXA = 5;                               // activation pieces per wave: ceil((BM + 2*160 + 2) / 16 / 8) for BM <= 256
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WGM, wn = wv / WGM;
    const int n_tile = (blockIdx.x >> 3) % a.n_tiles, m_tile = ((blockIdx.x >> 3) / a.n_tiles) * 8 + (blockIdx.x & 7);   // XCD-aware, see k_conv3_big
    if (m_tile >= a.m_tiles) return;
    const int P0 = m_tile * BM, n0 = n_tile * BN;
    const int HW = a.H * a.W;
    const int cpt = a.Cin >> 5;
    const int nsteps = 3 * cpt;
    const int wrow = 9 * a.Cin;
    const int apieces = a.apieces;                        // ceil((BM + 2W + 2) / 16)
    __half *act0 = (__half *)smem;                        // [2][apieces][16][32]
    __half *wt0 = act0 + (size_t)2 * apieces * 512;       // [2][WPIECES][16][32]
    const unsigned lds_act = lds_addr(act0), lds_w = lds_addr(wt0);

    const int lrow = lane >> 2, lp = lane & 3;
    unsigned aoff[MAXA], woff[NPW];
    const int Wp = PADO ? a.W + 1 : a.W, HWp = (a.H + 1) * Wp;      // PADO: pitch of the padded raster
    const float inv_wp = 1.0f / (float)Wp, inv_hwp = 1.0f / (float)HWp;
    unsigned long long amask[PADO ? MAXA : 1];            // PADO: the lanes of piece wv + 8k whose run row is a real pixel
    // PADO: position -> pixel index (or -1 for a pad position / past the tensor)
    auto pixel_of = [&](int pos) {
        if (pos < 0) return -1;
        const int b = div_s(pos, HWp, inv_hwp), rem = pos - b * HWp, yy = div_s(rem, Wp, inv_wp), xx = rem - yy * Wp;
        const int m = (b * a.H + yy) * a.W + xx;
        return (yy < a.H && xx < a.W && m < a.M) ? m : -1;
    };
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
        const int r = (wv + 8 * k) * 16 + lrow;           // row of the run
        int q;
        if (PADO) {
            q = pixel_of(P0 - Wp - 1 + r);
            amask[k] = __builtin_amdgcn_ballot_w64(q >= 0);
            if (q < 0) q = 0;
        } else {
            q = min(max(P0 - a.W - 1 + r, 0), a.M - 1);
        }
        aoff[k] = (unsigned)q * (unsigned)(a.ldi * 2) + (unsigned)(((lp - 2 * (r >> 2)) & 3) * 16);
    }
    int *rowmap = (int *)(smem + a.map_off);              // PADO: pixel index of every output row of the tile, -1 = skip
    if (PADO) {
        for (int r = tid; r < BM; r += 512) rowmap[r] = pixel_of(P0 + r);
        // both run buffers start as zeros: the masked pieces never write the pad positions
        for (int i = tid; i < 2 * apieces * 64; i += 512) reinterpret_cast<uint4 *>(smem)[i] = uint4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int rw = (wv + 8 * k) * 16 + lrow;
        const int dx = rw / BN, co = min(n0 + rw - dx * BN, a.CoutPad - 1);
        woff[k] = (unsigned)(co * wrow + min(dx, 2) * a.Cin) * 2u + (unsigned)(((lp - 2 * (rw >> 2)) & 3) * 16);
    }
    f4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    float4 bvs[FN];                                       // before the first DMA (see k_conv3_big)
#pragma unroll
    for (int i = 0; i < FN; ++i)
        bvs[i] = *reinterpret_cast<const float4 *>(a.bias + min(n0 + wn * TN + 16 * i + (lane >> 4) * 4, a.CoutPad - 4));

#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 8;
    const bool st_on = tid == 0 && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 8;
    const int st_slot = st_on ? blockIdx.x / st_stride : 0;
    int st_n = 0;
#endif
    STAMP(0);
    // prologue: the whole run of chunk 0 and the weights of step 0
#pragma unroll
    for (int k = 0; k < MAXA; ++k)
        if (wv + 8 * k < apieces) {
            if (PADO) lds_dma16_masked(aoff[k], a.in, lds_act + (unsigned)(wv + 8 * k) * 1024u, amask[PADO ? k : 0]);
            else lds_dma16(aoff[k], a.in, lds_act + (unsigned)(wv + 8 * k) * 1024u);
        }
#pragma unroll
    for (int k = 0; k < NPW; ++k)
        if (wv + 8 * k < WPIECES) lds_dma16(woff[k], a.w, lds_w + (unsigned)(wv + 8 * k) * 1024u);
    const int wlane = swz32(wn * TN + (lane & 15), lane >> 4);      // this lane's weight row of fragment 0, tap 0
    // 9-bit tap validity per pixel fragment (under the latency of the prologue burst: see div_s)
    int vm[FM];
    {
        const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)a.W;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int p = P0 + wm * TM + 16 * j + (lane & 15);
            int m = 0;
            if (!PADO && p < a.M) {
                const int rem = p - div_s(p, HW, inv_hw) * HW, oy = div_s(rem, a.W, inv_w), ox = rem - oy * a.W;
                const int hv = (ox >= 1 ? 1 : 0) | 2 | (ox <= a.W - 2 ? 4 : 0);
                m = (oy >= 1 ? hv : 0) | (hv << 3) | (oy <= a.H - 2 ? hv << 6 : 0);
            }
            vm[j] = m;
        }
    }

    int cc = 0, dy = 0;
    STAMP(1);
    for (int s = 0; s < nsteps; ++s) {
        wait_vm<0>();                                     // everything this wave issued a step ago has landed ...
        STAMP(2);
        __builtin_amdgcn_s_barrier();                     // ... and everybody's has; nobody still reads what the next burst overwrites
        STAMP(3);
        // next step's weights, and this step's third of the NEXT chunk's run
        int ncc = cc, ndy = dy + 1;
        if (ndy == 3) { ndy = 0; ++ncc; }
        ncc = __builtin_amdgcn_readfirstlane(ncc);
        ndy = __builtin_amdgcn_readfirstlane(ndy);
        // waves w and w + 4 share a SIMD: the lower four issue their burst before the MFMAs, the upper four after the first
        // horizontal tap -- while one of the pair is stalled in the vector-memory issue, the other keeps the matrix core fed
        // (both bursts still have most of a step to land before the next top-of-step wait)
#define RUN_ISSUE()                                                                                              \
        do {                                                                                                     \
            if (s + 1 < nsteps) {                                                                                \
                const char *wb_ = (const char *)a.w + (size_t)(ndy * 3 * a.Cin + ncc * 32) * 2;                  \
                const unsigned l_ = lds_w + (unsigned)((s + 1) & 1) * (unsigned)(WPIECES * 1024);                \
                _Pragma("unroll") for (int k = 0; k < NPW; ++k)                                                  \
                    if (wv + 8 * k < WPIECES) lds_dma16(woff[k], wb_, l_ + (unsigned)(wv + 8 * k) * 1024u);      \
            }                                                                                                    \
            if (cc + 1 < cpt) {                                                                                  \
                const char *ab_ = (const char *)a.in + (cc + 1) * 64;                                            \
                const unsigned l_ = lds_act + (unsigned)((cc + 1) & 1) * (unsigned)(apieces * 1024);             \
                _Pragma("unroll") for (int k = 0; k < MAXA; ++k)                                                 \
                    if (k % 3 == dy && wv + 8 * k < apieces) {                                                   \
                        if (PADO) lds_dma16_masked(aoff[k], ab_, l_ + (unsigned)(wv + 8 * k) * 1024u, amask[PADO ? k : 0]); \
                        else lds_dma16(aoff[k], ab_, l_ + (unsigned)(wv + 8 * k) * 1024u);                       \
                    }                                                                                            \
            }                                                                                                    \
        } while (0)
        const bool early = wv < 4;
        if (early) RUN_ISSUE();
        STAMP(4);
        const __half *ab = act0 + (size_t)(cc & 1) * apieces * 512;
        const __half *wb = wt0 + (size_t)(s & 1) * WPIECES * 512;
        // Fragment addresses = one swizzled per-lane base per horizontal tap + instruction immediates (16 rows further the
        // rotation of swz32 is the same; weight rows are the lane's base + a multiple of 16 rows): the PMC pass showed 4.4 vector
        // instructions per MFMA in this kernel, and every one of them competes with the MFMAs for the SIMD's issue port.
        const int arow = wm * TM + (lane & 15) + dy * Wp, ch = lane >> 4;
        const int abase[3] = {swz32(arow, ch), swz32(arow + 1, ch), swz32(arow + 2, ch)};
        const int vsh = dy * 3;
        // the fragments of the next horizontal tap are read while the MFMAs of this one run (two register sets); the padding
        // taps are zeroed where a fragment is consumed, a select right behind the load would wait for it at once
        h8 bfx[2][FM], afx[2][FN];
#define RUN_LOAD(S, DX)                                                                                          \
        do {                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                       \
                bfx[S][j] = *reinterpret_cast<const h8 *>(ab + abase[DX] + j * (16 * 32));                       \
            _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                       \
                afx[S][i] = *reinterpret_cast<const h8 *>(wb + wlane + ((DX) * BN + i * 16) * 32);               \
        } while (0)
#define RUN_MFMA(S, DX)                                                                                          \
        do {                                                                                                     \
            if (!NOSEL && !PADO)                                                                                 \
            _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                       \
                if (!((vm[j] >> (vsh + (DX))) & 1)) bfx[S][j] = hz;                                              \
            _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afx[S][i], bfx[S][j], acc[i][j], 0, 0, 0); \
        } while (0)
        RUN_LOAD(0, 0);
        RUN_LOAD(1, 1);
        RUN_MFMA(0, 0);
        if (!early) RUN_ISSUE();
        RUN_LOAD(0, 2);
        RUN_MFMA(1, 1);
        RUN_MFMA(0, 2);
#undef RUN_LOAD
#undef RUN_MFMA
#undef RUN_ISSUE
        cc = ncc; dy = ndy;
        STAMP(5);
    }
    __syncthreads();     // all waves done with the buffers: reuse them as the output staging tile

    constexpr int SROW = BN + 8;
    __half *stage = (__half *)smem;                       // [BM][SROW]
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int co = wn * TN + 16 * i + (lane >> 4) * 4;
        const float4 bv = bvs[i];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wm * TM + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 512) {
        const int row = q / CPR, pc = q - row * CPR;
        const int co = n0 + pc * 8;
        const int m = PADO ? rowmap[row] : P0 + row;
        if ((PADO ? m >= 0 : m < a.M) && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + (size_t)m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + (size_t)m * a.ldo + co) = v;
        }
    }
    STAMP(6);
}

template <int BM, int BN, int WGM, int WGN, bool NOSEL = false, bool PADO = false>
hipError_t launch_run(RunArgs &a, hipStream_t s)
{
    if (a.Cin % 32 || a.CoutPad % 4 || a.W > 160 || a.M >= (1 << 24)) return hipErrorInvalidValue;
    if ((size_t)a.M * a.ldi * 2 >= (1ull << 32) || (size_t)a.CoutPad * 9 * a.Cin * 2 >= (1ull << 32)) return hipErrorInvalidValue;
    const int wp = PADO ? a.W + 1 : a.W;
    a.apieces = rva_ceil_div(BM + 2 * wp + 2, 16);
    if (a.apieces > 8 * 5) return hipErrorInvalidValue;   // MAXA pieces per wave
    const size_t ring = (size_t)(2 * a.apieces + 2 * (3 * BN / 16)) * 1024;
    const size_t st = (size_t)BM * (BN + 8) * 2;
    size_t smem = ring > st ? ring : st;
    long mp = a.M;
    if (PADO) {
        const int batch = a.M / (a.H * a.W);
        mp = (long)batch * (a.H + 1) * (a.W + 1);         // positions of the padded raster (div_s: below 2^24)
        if (a.M % (a.H * a.W) || mp >= (1l << 24)) return hipErrorInvalidValue;
        a.map_off = (int)smem;                            // per-row pixel index behind the rings / the output stage
        smem += (size_t)BM * 4;
    }
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_run<BM, BN, WGM, WGN, NOSEL, PADO>, 160 * 1024); e != hipSuccess) return e;
    a.n_tiles = rva_ceil_div(a.Cout, BN);
    a.m_tiles = (int)((mp + BM - 1) / BM);
    k_conv3_run<BM, BN, WGM, WGN, NOSEL, PADO><<<rva_ceil_div(a.m_tiles, 8) * 8 * a.n_tiles, 512, smem, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// 3x3 STRIDE-2 convolution in the "long run" form (round 4).
//
// The five downsampling convolutions of YOLOv8 ran on the gather kernels (k_conv_gbig): every tap of a K-step fetches its own
// BM rows, nine fetches of the input per output tile -- 0.57-0.70 PFLOP/s against the 0.9 of k_conv3_run on the stride-1 layers.
// Here a tile is BM consecutive positions of the PADDED OUTPUT raster (row pitch Wo + 1: one pad position after every output
// row), and a (32-channel chunk, dy) step stages the input row 2 oy + dy - 1 of every position ONCE, split by column parity:
//   plane E, row t      = input pixel (2 oy + dy - 1, 2 ox)      of position P0 + t        -> the dx = 1 tap of position P0 + t
//   plane O, row t      = input pixel (2 oy + dy - 1, 2 ox + 1)  of position P0 - 1 + t    -> the dx = 0 tap of position P0 + t
//                                                                                           and the dx = 2 tap of P0 + t - 1
// so the three horizontal taps of a fragment are three FIXED row offsets into what the step staged (LDS-DMA takes a source address
// per lane: the stride-2 walk over the input costs nothing), 2 BM + 1 rows per three taps instead of 3 BM.  Padding without a
// select in the MFMA phase, as in k_conv3_run<PADO>: the lanes of a piece whose row is a pad position (the O row before ox = 0 is
// the pad position that ends the previous output row: the left padding) or lies above the image (oy = 0 at dy = 0) are masked out
// of the DMA, and their 16 bytes, zeroed once, are never written -- which is why the steps of each dy have a buffer of their own
// (three activation buffers; weights double-buffered as in k_conv3_run).  Even H and W (no bottom / right padding arises),
// Cin % 32 == 0.  LDS: 3 x (2 BM + 16) x 64 B + 2 x 3 BN x 64 B: 147 KB for 256 x 128, 75 KB for 128 x 64.
template <int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BM <= 128 && BN <= 64 ? 4 : 2))) k_conv3_s2run(RunArgs a)
{
    static_assert(WGM * WGN == 8, "eight waves");
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
    constexpr int WPIECES = 3 * BN / 16, NPW = (WPIECES + 7) / 8;
    constexpr int BMO = (BM + 1 + 15) / 16 * 16;          // rows of plane O (BM + 1 used), plane E follows
    constexpr int AP = (BMO + BM) / 16, MAXA = (AP + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WGM, wn = wv / WGM;
    const int n_tile = (blockIdx.x >> 3) % a.n_tiles, m_tile = ((blockIdx.x >> 3) / a.n_tiles) * 8 + (blockIdx.x & 7);   // XCD-aware, see k_conv3_big
    if (m_tile >= a.m_tiles) return;
    const int P0 = m_tile * BM, n0 = n_tile * BN;
    const int Ho = a.H >> 1, Wo = a.W >> 1, Wp = Wo + 1, HWp = Ho * Wp;
    const int cpt = a.Cin >> 5, nsteps = 3 * cpt, wrow = 9 * a.Cin;
    __half *act0 = (__half *)smem;                        // [3 dy][AP][16][32]
    __half *wt0 = act0 + (size_t)3 * AP * 512;            // [2][WPIECES][16][32]
    const unsigned lds_act = lds_addr(act0), lds_w = lds_addr(wt0);
    const int lrow = lane >> 2, lp = lane & 3;
    const float inv_wp = 1.0f / (float)Wp, inv_hwp = 1.0f / (float)HWp;
    const int batch = a.M / (Ho * Wo);
    // position of the padded output raster -> (image, oy, ox); ox == Wo is the pad position
    unsigned aoff[MAXA], woff[NPW];
    unsigned long long am1[MAXA], am0[MAXA];              // lanes whose row is a real input pixel at dy >= 1 / at dy = 0
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
        const int r = (wv + 8 * k) * 16 + lrow;           // row of the step's buffer
        const bool odd = r < BMO;
        const int pos = odd ? P0 - 1 + r : P0 + (r - BMO);
        bool ok = (odd ? r <= BM : r < BMO + BM) && pos >= 0;
        int b = 0, oy = 0, ox = 0;
        if (ok) {
            b = div_s(pos, HWp, inv_hwp);
            const int rem = pos - b * HWp;
            oy = div_s(rem, Wp, inv_wp); ox = rem - oy * Wp;
            ok = ox < Wo && b < batch;
        }
        am1[k] = __builtin_amdgcn_ballot_w64(ok);
        am0[k] = __builtin_amdgcn_ballot_w64(ok && oy >= 1);
        const int ipix = ok ? (b * a.H + 2 * oy) * a.W + 2 * ox + (odd ? 1 : 0) : 0;       // the dy = 1 row; dy = 0 / 2 shift the base
        aoff[k] = (unsigned)ipix * (unsigned)(a.ldi * 2) + (unsigned)(((lp - 2 * (r >> 2)) & 3) * 16);
    }
    int *rowmap = (int *)(smem + a.map_off);              // output pixel index of every row of the tile, -1 = skip
    for (int r = tid; r < BM; r += 512) {
        const int pos = P0 + r;
        const int b = div_s(pos, HWp, inv_hwp), rem = pos - b * HWp, oy = div_s(rem, Wp, inv_wp), ox = rem - oy * Wp;
        rowmap[r] = (ox < Wo && b < batch) ? (b * Ho + oy) * Wo + ox : -1;
    }
    for (int i = tid; i < 3 * AP * 64; i += 512) reinterpret_cast<uint4 *>(smem)[i] = uint4{0u, 0u, 0u, 0u};     // the masked lanes' bytes stay zero
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int rw = (wv + 8 * k) * 16 + lrow;
        const int dx = rw / BN, co = min(n0 + rw - dx * BN, a.CoutPad - 1);
        woff[k] = (unsigned)(co * wrow + min(dx, 2) * a.Cin) * 2u + (unsigned)(((lp - 2 * (rw >> 2)) & 3) * 16);
    }
    f4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float4 bvs[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i)
        bvs[i] = *reinterpret_cast<const float4 *>(a.bias + min(n0 + wn * TN + 16 * i + (lane >> 4) * 4, a.CoutPad - 4));
    const long long rowb = (long long)a.W * a.ldi * 2;    // bytes of an input row
    // step (cc, dy): activations into buffer dy, weights into buffer (step & 1)
    auto issue = [&](int step, int cc, int dy) {
        const char *wb_ = (const char *)a.w + (size_t)(dy * 3 * a.Cin + cc * 32) * 2;
        const unsigned lw_ = lds_w + (unsigned)(step & 1) * (unsigned)(WPIECES * 1024);
#pragma unroll
        for (int k = 0; k < NPW; ++k)
            if (wv + 8 * k < WPIECES) lds_dma16(woff[k], wb_, lw_ + (unsigned)(wv + 8 * k) * 1024u);
        const char *ab_ = (const char *)a.in + (long long)cc * 64 + (long long)(dy - 1) * rowb;
        const unsigned la_ = lds_act + (unsigned)dy * (unsigned)(AP * 1024);
#pragma unroll
        for (int k = 0; k < MAXA; ++k)
            if (wv + 8 * k < AP) lds_dma16_masked(aoff[k], ab_, la_ + (unsigned)(wv + 8 * k) * 1024u, dy == 0 ? am0[k] : am1[k]);
    };
    issue(0, 0, 0);
    const int wlane = swz32(wn * TN + (lane & 15), lane >> 4);      // this lane's weight row of fragment 0, tap 0
    const int t0 = wm * TM + (lane & 15), ch = lane >> 4;
    const int abase[3] = {swz32(t0, ch), swz32(BMO + t0, ch), swz32(t0 + 1, ch)};      // dx = 0: O[t], dx = 1: E[t], dx = 2: O[t + 1]
    int cc = 0, dy = 0;
    for (int s = 0; s < nsteps; ++s) {
        wait_vm<0>();                                     // what this wave issued a step ago has landed ...
        __builtin_amdgcn_s_barrier();                     // ... and everybody's has; nobody still reads what the next burst overwrites
        int ncc = cc, ndy = dy + 1;
        if (ndy == 3) { ndy = 0; ++ncc; }
        ncc = __builtin_amdgcn_readfirstlane(ncc);
        ndy = __builtin_amdgcn_readfirstlane(ndy);
        const bool early = wv < 4;                        // waves w and w + 4 share a SIMD: see k_conv3_run
        if (early && s + 1 < nsteps) issue(s + 1, ncc, ndy);
        const __half *ab = act0 + (size_t)dy * AP * 512;
        const __half *wb = wt0 + (size_t)(s & 1) * WPIECES * 512;
        h8 bfx[2][FM], afx[2][FN];
#define S2_LOAD(S, DX)                                                                                           \
        do {                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                       \
                bfx[S][j] = *reinterpret_cast<const h8 *>(ab + abase[DX] + j * (16 * 32));                       \
            _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                       \
                afx[S][i] = *reinterpret_cast<const h8 *>(wb + wlane + ((DX) * BN + i * 16) * 32);               \
        } while (0)
#define S2_MFMA(S)                                                                                               \
        do {                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afx[S][i], bfx[S][j], acc[i][j], 0, 0, 0); \
        } while (0)
        S2_LOAD(0, 0);
        S2_LOAD(1, 1);
        S2_MFMA(0);
        if (!early && s + 1 < nsteps) issue(s + 1, ncc, ndy);
        S2_LOAD(0, 2);
        S2_MFMA(1);
        S2_MFMA(0);
#undef S2_LOAD
#undef S2_MFMA
        cc = ncc; dy = ndy;
    }
    __syncthreads();     // all waves done with the buffers: reuse them as the output staging tile

    constexpr int SROW = BN + 8;
    __half *stage = (__half *)smem;                       // [BM][SROW]
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int co = wn * TN + 16 * i + (lane >> 4) * 4;
        const float4 bv = bvs[i];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wm * TM + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 512) {
        const int row = q / CPR, pc = q - row * CPR;
        const int co = n0 + pc * 8;
        const int m = rowmap[row];
        if (m >= 0 && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + (size_t)m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + (size_t)m * a.ldo + co) = v;
        }
    }
}

// a.H / a.W: INPUT size (even), a.M: output pixels
template <int BM, int BN, int WGM, int WGN>
hipError_t launch_s2run(RunArgs &a, hipStream_t s)
{
    constexpr int BMO = (BM + 1 + 15) / 16 * 16, AP = (BMO + BM) / 16;
    if (a.Cin % 32 || a.CoutPad % 4 || (a.H & 1) || (a.W & 1)) return hipErrorInvalidValue;
    const int Ho = a.H / 2, Wo = a.W / 2;
    if (Ho <= 0 || Wo <= 0 || a.M % (Ho * Wo)) return hipErrorInvalidValue;
    const long batch = a.M / (Ho * Wo);
    const long mp = batch * Ho * (Wo + 1);                // positions of the padded output raster (div_s: below 2^24)
    if (mp >= (1l << 24)) return hipErrorInvalidValue;
    // LDS-DMA addresses: a 64-bit scalar base + a 32-bit per-lane byte offset
    if ((size_t)batch * a.H * a.W * a.ldi * 2 >= (1ull << 32) || (size_t)a.CoutPad * 9 * a.Cin * 2 >= (1ull << 32)) return hipErrorInvalidValue;
    const size_t ring = (size_t)(3 * AP + 2 * (3 * BN / 16)) * 1024;
    const size_t st = (size_t)BM * (BN + 8) * 2;
    size_t smem = ring > st ? ring : st;
    a.map_off = (int)smem;
    smem += (size_t)BM * 4;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_s2run<BM, BN, WGM, WGN>, 160 * 1024); e != hipSuccess) return e;
    a.n_tiles = rva_ceil_div(a.Cout, BN);
    a.m_tiles = (int)((mp + BM - 1) / BM);
    k_conv3_s2run<BM, BN, WGM, WGN><<<rva_ceil_div(a.m_tiles, 8) * 8 * a.n_tiles, 512, smem, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, "whole chunk per barrier" LDS-DMA variant (round 4) for layers with FEW pixels.
//
// Where k_conv3_run pays off it is because a 256 x 128 tile has ~1.5 k cycles of MFMA work per (chunk, dy) step to set
// against the ~1 k cycles of wait -> barrier -> issue that every step costs.  The small-M layers have no such tiles: at
// batch 4 (one GPU's share of BASELINE configs[3]) YOLOv8m's 192 -> 192 convolution at 40 x 40 is 6 400 pixels -- 150
// workgroups of 128 x 64, each walking 54 (chunk, dy) steps of 12 MFMAs per wave at ~600 cycles per step: 18.4 us per
// launch against a 1.7 us MFMA floor, seventeen times per forward pass (profiles/r04_m4_conv_tuning.txt).  What bounds
// such a launch is steps x step latency, so this kernel makes the step fat: ONE barrier per 32-channel chunk.  A step
// stages the chunk's activation run (as k_conv3_run does, once for all nine taps) AND the chunk's weights for all nine
// taps ([9][BN] rows of 64 B), both double-buffered across chunks, and runs dy x dx = 9 taps of MFMAs between two
// barriers: Cin / 32 steps instead of 3 Cin / 32.  LDS: 2 x (run + 9 BN / 16) KiB -- 92-150 KB, one workgroup per CU,
// which costs nothing where the grid has fewer workgroups than the chip has CUs.  BN = 96 exists for YOLOv8m, whose widths
// (48 / 96 / 192 / 288 / 384 / 576) are multiples of 96, not of 128.  Same K order (chunk, dy, dx), same fp32 accumulation, same
// epilogue as k_conv3_run.  Needs Cin % 32 == 0 and W <= 160.
template <int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) k_conv3_chunk(RunArgs a)
{
    static_assert(WGM * WGN == 8, "eight waves");
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "whole fragments per wave");
    constexpr int WPIECES = 9 * BN / 16;                  // weights [9 taps][BN] rows of one chunk
    constexpr int NPW = (WPIECES + 7) / 8;
    constexpr int MAXA = 5;                               // activation pieces per wave: ceil((BM + 2*160 + 2) / 16 / 8) for BM <= 256
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WGM, wn = wv / WGM;
    const int n_tile = (blockIdx.x >> 3) % a.n_tiles, m_tile = ((blockIdx.x >> 3) / a.n_tiles) * 8 + (blockIdx.x & 7);   // XCD-aware, see k_conv3_big
    if (m_tile >= a.m_tiles) return;
    const int P0 = m_tile * BM, n0 = n_tile * BN;
    const int HW = a.H * a.W;
    const int cpt = a.Cin >> 5;
    const int wrow = 9 * a.Cin;
    const int apieces = a.apieces;                        // ceil((BM + 2W + 2) / 16)
    __half *act0 = (__half *)smem;                        // [2][apieces][16][32]
    __half *wt0 = act0 + (size_t)2 * apieces * 512;       // [2][WPIECES][16][32]
    const unsigned lds_act = lds_addr(act0), lds_w = lds_addr(wt0);

    const int lrow = lane >> 2, lp = lane & 3;
    unsigned aoff[MAXA], woff[NPW];
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
        const int r = (wv + 8 * k) * 16 + lrow;           // row of the run
        const int q = min(max(P0 - a.W - 1 + r, 0), a.M - 1);
        aoff[k] = (unsigned)q * (unsigned)(a.ldi * 2) + (unsigned)(((lp - 2 * (r >> 2)) & 3) * 16);
    }
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int rw = (wv + 8 * k) * 16 + lrow;          // row of the [9][BN] weight image
        const int tap = min(rw / BN, 8), co = min(n0 + rw - tap * BN, a.CoutPad - 1);
        woff[k] = (unsigned)(co * wrow + tap * a.Cin) * 2u + (unsigned)(((lp - 2 * (rw >> 2)) & 3) * 16);
    }
    f4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    float4 bvs[FN];                                       // before the first DMA (see k_conv3_big)
#pragma unroll
    for (int i = 0; i < FN; ++i)
        bvs[i] = *reinterpret_cast<const float4 *>(a.bias + min(n0 + wn * TN + 16 * i + (lane >> 4) * 4, a.CoutPad - 4));

#define CHUNK_ISSUE(CC)                                                                                          \
    do {                                                                                                         \
        const char *ab_ = (const char *)a.in + (CC) * 64;                                                        \
        const char *wb_ = (const char *)a.w + (CC) * 64;                                                         \
        const unsigned la_ = lds_act + (unsigned)((CC) & 1) * (unsigned)(apieces * 1024);                        \
        const unsigned lw_ = lds_w + (unsigned)((CC) & 1) * (unsigned)(WPIECES * 1024);                          \
        _Pragma("unroll") for (int k = 0; k < NPW; ++k)                                                          \
            if (wv + 8 * k < WPIECES) lds_dma16(woff[k], wb_, lw_ + (unsigned)(wv + 8 * k) * 1024u);             \
        _Pragma("unroll") for (int k = 0; k < MAXA; ++k)                                                         \
            if (wv + 8 * k < apieces) lds_dma16(aoff[k], ab_, la_ + (unsigned)(wv + 8 * k) * 1024u);             \
    } while (0)
    CHUNK_ISSUE(0);
    // 9-bit tap validity per pixel fragment (under the latency of the first burst: see div_s)
    int vm[FM];
    {
        const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)a.W;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int p = P0 + wm * TM + 16 * j + (lane & 15);
            int m = 0;
            if (p < a.M) {
                const int rem = p - div_s(p, HW, inv_hw) * HW, oy = div_s(rem, a.W, inv_w), ox = rem - oy * a.W;
                const int hv = (ox >= 1 ? 1 : 0) | 2 | (ox <= a.W - 2 ? 4 : 0);
                m = (oy >= 1 ? hv : 0) | (hv << 3) | (oy <= a.H - 2 ? hv << 6 : 0);
            }
            vm[j] = m;
        }
    }
    // per-lane fragment bases, fixed for the whole kernel: the run row of the lane's first pixel for each of the nine taps
    // (swizzled; 16 rows further the rotation repeats, so the other fragments are immediates) and its first weight row
    const int ch = lane >> 4;
    int abase[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) abase[t] = swz32(wm * TM + (lane & 15) + (t / 3) * a.W + (t % 3), ch);
    const int wlane = swz32(wn * TN + (lane & 15), ch);

    for (int cc = 0; cc < cpt; ++cc) {
        wait_vm<0>();                                     // this wave's pieces of chunk cc have landed ...
        __builtin_amdgcn_s_barrier();                     // ... and everybody's; nobody still reads the buffers of chunk cc - 1
        // waves w and w + 4 share a SIMD: the lower four issue the next chunk's burst before their MFMAs, the upper four
        // after their first tap (see k_conv3_run)
        const bool early = wv < 4;
        if (early && cc + 1 < cpt) CHUNK_ISSUE(cc + 1);
        const __half *ab = act0 + (size_t)(cc & 1) * apieces * 512;
        const __half *wb = wt0 + (size_t)(cc & 1) * WPIECES * 512;
        h8 bfx[2][FM], afx[2][FN];
#define CHUNK_LOAD(S, T)                                                                                         \
        do {                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                       \
                bfx[S][j] = *reinterpret_cast<const h8 *>(ab + abase[T] + j * (16 * 32));                        \
            _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                       \
                afx[S][i] = *reinterpret_cast<const h8 *>(wb + wlane + ((T) * BN + i * 16) * 32);                \
        } while (0)
#define CHUNK_MFMA(S, T)                                                                                         \
        do {                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                       \
                if (!((vm[j] >> (T)) & 1)) bfx[S][j] = hz;                                                       \
            _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < FM; ++j)                                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afx[S][i], bfx[S][j], acc[i][j], 0, 0, 0); \
        } while (0)
        CHUNK_LOAD(0, 0);
        CHUNK_LOAD(1, 1);
        CHUNK_MFMA(0, 0);
        if (!early && cc + 1 < cpt) CHUNK_ISSUE(cc + 1);
        CHUNK_LOAD(0, 2); CHUNK_MFMA(1, 1);
        CHUNK_LOAD(1, 3); CHUNK_MFMA(0, 2);
        CHUNK_LOAD(0, 4); CHUNK_MFMA(1, 3);
        CHUNK_LOAD(1, 5); CHUNK_MFMA(0, 4);
        CHUNK_LOAD(0, 6); CHUNK_MFMA(1, 5);
        CHUNK_LOAD(1, 7); CHUNK_MFMA(0, 6);
        CHUNK_LOAD(0, 8); CHUNK_MFMA(1, 7);
        CHUNK_MFMA(0, 8);
#undef CHUNK_LOAD
#undef CHUNK_MFMA
    }
#undef CHUNK_ISSUE
    __syncthreads();     // all waves done with the buffers: reuse them as the output staging tile

    constexpr int SROW = BN + 8;
    __half *stage = (__half *)smem;                       // [BM][SROW]
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int co = wn * TN + 16 * i + (lane >> 4) * 4;
        const float4 bv = bvs[i];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wm * TM + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 512) {
        const int row = q / CPR, pc = q - row * CPR;
        const int co = n0 + pc * 8;
        const int m = P0 + row;
        if (m < a.M && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + (size_t)m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + (size_t)m * a.ldo + co) = v;
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
hipError_t launch_chunk(RunArgs &a, hipStream_t s)
{
    if (a.Cin % 32 || a.CoutPad % 4 || a.W > 160 || a.M >= (1 << 24)) return hipErrorInvalidValue;
    if ((size_t)a.M * a.ldi * 2 >= (1ull << 32) || (size_t)a.CoutPad * 9 * a.Cin * 2 >= (1ull << 32)) return hipErrorInvalidValue;
    a.apieces = rva_ceil_div(BM + 2 * a.W + 2, 16);
    if (a.apieces > 8 * 5) return hipErrorInvalidValue;   // MAXA pieces per wave
    const size_t ring = (size_t)(2 * a.apieces + 2 * (9 * BN / 16)) * 1024;
    const size_t st = (size_t)BM * (BN + 8) * 2;
    const size_t smem = ring > st ? ring : st;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_chunk<BM, BN, WGM, WGN>, 160 * 1024); e != hipSuccess) return e;
    a.n_tiles = rva_ceil_div(a.Cout, BN);
    a.m_tiles = rva_ceil_div(a.M, BM);
    k_conv3_chunk<BM, BN, WGM, WGN><<<rva_ceil_div(a.m_tiles, 8) * 8 * a.n_tiles, 512, smem, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Large-tile LDS-DMA kernel for the gathered cases: 1x1 convolutions and 3x3 stride-2 (or stride-1) convolutions.
// Same ring / counted-vmcnt / raw-barrier structure as k_conv3_big, but a K-step is (tap, 64 channels): every staged
// row is one full 128-byte line of a pixel (or of a weight row), a 1 KiB piece is 8 rows, and the bank-conflict-free
// image is the XOR swizzle  physical chunk = chunk ^ (row & 7)  (applied on the per-lane source address; for
// ds_read_b128 fragment reads the lane's xor term is the constant lane & 7).  BK = 32 (layers with Cin = 32 / 96)
// uses 64-byte rows with the swz32 rotation instead, 16 rows per piece.  Every wave owns BM/64 activation pieces
// and BN/64 weight pieces per step, so the vmcnt count is a compile-time constant.  Taps that fall outside the image
// read a clamped address and are zeroed per lane in the B fragment (9-bit mask per pixel).  Needs Cin % 64 == 0.
// SUB > 1 (round 4, two-slot rings only): SUB consecutive K-steps share ONE barrier -- the ring has NSLOT * SUB sub-slots, a
// burst issues SUB steps, the MFMAs of SUB steps run between two barriers.  For the layers with few pixels (stride-2 and 1x1
// layers at 20 x 20 / 40 x 40, batch 4: 13-78 workgroups walking 18-54 steps) the launch time is steps x step latency, and this
// divides the steps; the LDS it costs (one workgroup per CU) is free where the grid is smaller than the chip.
template <int BM, int BN, int WGM, int WGN, int NSLOT, int KS, int BK, bool UP = false, bool HEAD = false, int SUB = 1>
__global__ void __launch_bounds__(512)
    __attribute__((amdgpu_waves_per_eu(NSLOT * SUB * (BM + BN) * BK * 2 <= 80 * 1024 ? 4 : 2))) k_conv_gbig(ConvArgs a)
{
    static_assert(WGM * WGN == 8, "eight waves");
    static_assert(BK == 32 || BK == 64, "K-step of 32 or 64 channels");
    static_assert(SUB == 1 || NSLOT == 2, "several K-steps per barrier: two-slot ring");
    static_assert(!UP || (KS == 1 && BK == 64), "the upsample + concat source form exists for 1x1 convolutions");
    constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
    constexpr int RPP = 512 / BK;                                  // rows per 1 KiB piece (8 rows of 128 B / 16 of 64 B)
    constexpr int LPR = BK / 8;                                    // lanes (16-byte chunks) per row
    constexpr int APIECES = BM / RPP, WPIECES = BN / RPP;
    constexpr int NA = (APIECES + 7) / 8, NW = (WPIECES + 7) / 8;  // pieces per wave per step (upper bounds)
    constexpr int SLOTH = (BM + BN) * BK;                          // halfs per ring slot
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *ring = (__half *)smem;                                 // [NSLOT * SUB][BM + BN][BK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WGM, wn = wv / WGM;
    // XCD-aware tile order (see k_conv3_big): the n-tiles of one m-tile get workgroup ids 8 apart
    const int n_tile = (blockIdx.x >> 3) % a.n_tiles, m_tile = ((blockIdx.x >> 3) / a.n_tiles) * 8 + (blockIdx.x & 7);
    if (m_tile >= a.m_tiles) return;
    const int P0 = m_tile * BM, n0 = n_tile * BN;
    const int HoWo = a.Ho * a.Wo;
    const int cpt = a.Cin / BK;
    const int nsteps = TAPS * cpt;
    const int wrow = TAPS * a.Cin;

    // per-lane constants of this wave's pieces (piece index = wv + 8k within the activation / weight region)
    const int lrow = lane / LPR, lp = lane % LPR;
    // source chunk of this lane = inverse of the LDS swizzle (BK 64: chunk ^ (row & 7); BK 32: swz32's rotation)
    const int c8 = (BK == 64 ? (lp ^ lrow) : ((lp - 2 * (lrow >> 2)) & 3)) * 8;
    // LDS-DMA addressing (lds_dma16): per piece ONE per-lane byte offset, computed here once -- the lane's source row at
    // the CENTRE tap (always inside the image) plus its swizzled chunk -- and a 9-bit mask of the taps that stay inside the
    // image.  Per K-step the tap displacement and the channel chunk are wave-uniform: the displacement is added to the
    // offset of lanes whose tap is valid (the others re-read the centre row; their B fragments are zeroed below), the chunk
    // sits in the scalar base.  UP: two sources, one offset each (low-res pixel / full-res pixel).
    unsigned actr[NA], actr2[UP ? NA : 1];
    int avm[NA];
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;      // div_s: raster positions below 2^24 (launch_gbig1)
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int m = P0 + (wv + 8 * k) * RPP + lrow;
        actr[k] = (unsigned)(c8 * 2);
        avm[k] = 0;
        if (UP) actr2[k] = (unsigned)(c8 * 2);
        if (wv + 8 * k < APIECES && m < a.M) {
            const int b = div_s(m, HoWo, inv_howo), rem = m - b * HoWo, oy = div_s(rem, a.Wo, inv_wo), ox = rem - oy * a.Wo;
            const int iy0 = oy * a.stride - PAD, ix0 = ox * a.stride - PAD;
            const unsigned pix = (unsigned)((b * a.H + iy0 + PAD) * a.W + ix0 + PAD);
            actr[k] = pix * (unsigned)((UP ? a.ldi2 : a.ldi) * 2) + (unsigned)(c8 * 2);
            if (UP) actr2[k] = (unsigned)((b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * (unsigned)(a.ldi * 2) + (unsigned)(c8 * 2);
            int msk = KS == 1 ? 1 : 0;
            if (KS == 3) {
                const int vy = ((unsigned)iy0 < (unsigned)a.H ? 1 : 0) | ((unsigned)(iy0 + 1) < (unsigned)a.H ? 2 : 0) | ((unsigned)(iy0 + 2) < (unsigned)a.H ? 4 : 0);
                const int vx = ((unsigned)ix0 < (unsigned)a.W ? 1 : 0) | ((unsigned)(ix0 + 1) < (unsigned)a.W ? 2 : 0) | ((unsigned)(ix0 + 2) < (unsigned)a.W ? 4 : 0);
                msk = ((vy & 1) ? vx : 0) | ((vy & 2) ? vx << 3 : 0) | ((vy & 4) ? vx << 6 : 0);
            }
            avm[k] = msk;
        }
    }
    unsigned woff[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) woff[k] = (unsigned)(min(n0 + (wv + 8 * k) * RPP + lrow, a.CoutPad - 1) * wrow + c8) * 2u;
    const int my_pieces = max(0, (APIECES - wv + 7) / 8) + max(0, (WPIECES - wv + 7) / 8);
    const unsigned ring0 = lds_addr(ring);
    // issue state of the next step (scalar registers; macros, not lambdas: see k_conv3_big)
    int is_tap = 0, is_cc = 0, is_slot = 0;
    int is_toff = KS == 3 ? -(a.W + 1) * a.ldi * 2 : 0;            // byte displacement of the tap from the centre row
    const char *is_abase = (const char *)a.in, *is_wbase = (const char *)a.w;
    bool is_low = UP;                                              // UP: this chunk comes from the low-res tensor
    unsigned is_lds = ring0;
#define GB_ISSUE()                                                                                                  \
    do {                                                                                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < NA; ++k_) {                                                         \
            if (wv + 8 * k_ < APIECES) {                                                                            \
                unsigned v_ = actr[k_];                                                                             \
                if (KS == 3 && ((avm[k_] >> is_tap) & 1)) v_ += (unsigned)is_toff;                                  \
                if (UP && is_low) v_ = actr2[k_];                                                                   \
                lds_dma16(v_, is_abase, is_lds + (unsigned)(wv + 8 * k_) * 1024u);                                  \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int k_ = 0; k_ < NW; ++k_) {                                                         \
            if (wv + 8 * k_ < WPIECES) lds_dma16(woff[k_], is_wbase, is_lds + (unsigned)(BM * BK * 2) + (unsigned)(wv + 8 * k_) * 1024u); \
        }                                                                                                           \
        ++is_cc;                                                                                                    \
        if (is_cc == cpt) { is_cc = 0; ++is_tap; }                                                                  \
        ++is_slot;                                                                                                  \
        if (is_slot == NSLOT * SUB) is_slot = 0;                                                                    \
        is_cc = __builtin_amdgcn_readfirstlane(is_cc);                                                              \
        is_tap = __builtin_amdgcn_readfirstlane(is_tap);                                                            \
        is_slot = __builtin_amdgcn_readfirstlane(is_slot);                                                          \
        if (KS == 3) {                                                                                              \
            const int dy_ = is_tap / 3, dx_ = is_tap - dy_ * 3;                                                     \
            is_toff = ((dy_ - 1) * a.W + (dx_ - 1)) * a.ldi * 2;                                                    \
        }                                                                                                           \
        if (UP) {                                                                                                   \
            const int ch_ = is_cc * BK;                                                                             \
            is_low = ch_ < a.c_split;                                                                               \
            is_abase = is_low ? (const char *)a.in + ch_ * 2 : (const char *)a.in2 + (ch_ - a.c_split) * 2;         \
        } else {                                                                                                    \
            is_abase = (const char *)a.in + is_cc * (BK * 2);                                                       \
        }                                                                                                           \
        is_wbase = (const char *)a.w + (size_t)(is_tap * a.Cin + is_cc * BK) * 2;                                   \
        is_lds = ring0 + (unsigned)is_slot * (unsigned)(SLOTH * 2);                                                 \
    } while (0)


    f4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    float4 bvs[FN];                                                // before the first DMA: see k_conv3_big
#pragma unroll
    for (int i = 0; i < FN; ++i)
        bvs[i] = *reinterpret_cast<const float4 *>(a.bias + min(n0 + wn * TN + 16 * i + (lane >> 4) * 4, a.CoutPad - 4));

#pragma unroll
    for (int t = 0; t < (NSLOT - 1) * SUB; ++t)
        if (t < nsteps) GB_ISSUE();
    // tap validity of this lane's pixels (bit = tap index); 1x1: every tap valid.  Computed here, under the latency of the first
    // burst (see div_s)
    int vm[FM];
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        int msk = KS == 1 ? 1 : 0;
        if (KS == 3) {
            const int m = P0 + wm * TM + 16 * j + (lane & 15);
            if (m < a.M) {
                const int b = div_s(m, HoWo, inv_howo), rem = m - b * HoWo, oy = div_s(rem, a.Wo, inv_wo), ox = rem - oy * a.Wo;
                const int iy0 = oy * a.stride - 1, ix0 = ox * a.stride - 1;
                const int vy = ((unsigned)iy0 < (unsigned)a.H ? 1 : 0) | ((unsigned)(iy0 + 1) < (unsigned)a.H ? 2 : 0) | ((unsigned)(iy0 + 2) < (unsigned)a.H ? 4 : 0);
                const int vx = ((unsigned)ix0 < (unsigned)a.W ? 1 : 0) | ((unsigned)(ix0 + 1) < (unsigned)a.W ? 2 : 0) | ((unsigned)(ix0 + 2) < (unsigned)a.W ? 4 : 0);
                msk = ((vy & 1) ? vx : 0) | ((vy & 2) ? vx << 3 : 0) | ((vy & 4) ? vx << 6 : 0);
            }
        }
        vm[j] = msk;
    }
    const int xr = lane & 7;                                       // BK 64: (row & 7) of every fragment row this lane reads
    const int nfat = (nsteps + SUB - 1) / SUB;                     // barriers: one per SUB K-steps
    for (int sf = 0; sf < nfat; ++sf) {
        // steps s+1 .. s+NSLOT-2 may stay in flight (their pieces were issued after step s's)
        wait_vm_n(NSLOT >= 3 ? min(max(nsteps - 1 - sf, 0), NSLOT - 2) * my_pieces : 0);
        __builtin_amdgcn_s_barrier();
        const int sn = (sf + NSLOT - 1) * SUB;                     // first K-step of the burst issued now
        const bool more = sn < nsteps;
        const bool early = NSLOT == 2 || wv < 4;
        if (early) {
#pragma unroll
            for (int u = 0; u < SUB; ++u)
                if (sn + u < nsteps) GB_ISSUE();
        }
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        const int s = sf * SUB + u;
        if (SUB > 1 && s >= nsteps) break;
        const int tap = s / cpt;
        const __half *ab = ring + (size_t)(s % (NSLOT * SUB)) * SLOTH;
        const __half *wb = ab + BM * BK;
        const int arow = wm * TM + (lane & 15), wr = wn * TN + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            h8 bf[FM];
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int r = arow + 16 * j;
                bf[j] = *reinterpret_cast<const h8 *>(ab + (BK == 64 ? r * 64 + ((ks * 4 + (lane >> 4)) ^ xr) * 8 : swz32(r, lane >> 4)));
                if (KS == 3 && !((vm[j] >> tap) & 1)) bf[j] = hz;
            }
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int r = wr + 16 * i;
                const h8 af = *reinterpret_cast<const h8 *>(wb + (BK == 64 ? r * 64 + ((ks * 4 + (lane >> 4)) ^ xr) * 8 : swz32(r, lane >> 4)));
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
            }
        }
      }
        if (more && !early) GB_ISSUE();                            // (three-slot rings: SUB == 1)
    }
#undef GB_ISSUE
    __syncthreads();     // ring drained and no longer read: reuse it as the output staging tile

    constexpr int SROW = BN + 8;
    __half *stage = (__half *)smem;                                // [BM][SROW]
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int co = wn * TN + 16 * i + (lane >> 4) * 4;
        const float4 bv = bvs[i];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
            const int px = wm * TM + 16 * j + (lane & 15);
            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t *>(&lo);
            pk.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
        }
    }
    __syncthreads();
    if constexpr (HEAD) {
        // thread = pixel row of the staged tile (values rounded to fp16 exactly as the stand-alone path stores them);
        // same operation order as head_anchor, so the result is bit-identical to conv -> k_head
        for (int r = tid; r < BM; r += 512) {
            const int m = P0 + r;
            if (m >= a.M) continue;
            const int b = m / a.hHW, i = m - b * a.hHW;
            __half *o = a.hout + (size_t)b * (4 + a.hnc) * a.hA + a.ha0 + i;
            const __half *row = stage + (size_t)r * SROW;
            if (a.hmode == 1) {
                float d[4];
#pragma unroll
                for (int sd = 0; sd < 4; ++sd) {
                    float v[16], mx = -1e30f;
                    const uint4 q0 = *reinterpret_cast<const uint4 *>(row + sd * 16), q1 = *reinterpret_cast<const uint4 *>(row + sd * 16 + 8);
                    const __half2 *h0 = reinterpret_cast<const __half2 *>(&q0), *h1 = reinterpret_cast<const __half2 *>(&q1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float2 f = __half22float2(h0[t]); v[2 * t] = f.x; v[2 * t + 1] = f.y;
                        f = __half22float2(h1[t]); v[8 + 2 * t] = f.x; v[9 + 2 * t] = f.y;
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) mx = fmaxf(mx, v[t]);
                    float se = 0.f, sw = 0.f;
#pragma unroll
                    for (int t = 0; t < 16; ++t) { const float e = __expf(v[t] - mx); se += e; sw += e * (float)t; }
                    d[sd] = sw / se;
                }
                const float ax = (float)(i % a.hW) + 0.5f, ay = (float)(i / a.hW) + 0.5f;
                const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
                o[0] = __float2half_rn((x1 + x2) * 0.5f * a.hstride);
                o[(size_t)a.hA] = __float2half_rn((y1 + y2) * 0.5f * a.hstride);
                o[(size_t)2 * a.hA] = __float2half_rn((x2 - x1) * a.hstride);
                o[(size_t)3 * a.hA] = __float2half_rn((y2 - y1) * a.hstride);
            } else {
                const int cend = min(BN, a.Cout - n0);
                for (int c = 0; c < cend; c += 8) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(row + c);
                    const __half2 *hq = reinterpret_cast<const __half2 *>(&q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float2 f = __half22float2(hq[t]);
                        o[(size_t)(4 + n0 + c + 2 * t) * a.hA] = __float2half_rn(__builtin_amdgcn_rcpf(1.0f + __expf(-f.x)));
                        o[(size_t)(5 + n0 + c + 2 * t) * a.hA] = __float2half_rn(__builtin_amdgcn_rcpf(1.0f + __expf(-f.y)));
                    }
                }
            }
        }
        return;
    }
    constexpr int CPR = BN / 8;
#pragma unroll 4
    for (int q = tid; q < BM * CPR; q += 512) {
        const int row = q / CPR, pc = q - row * CPR;
        const int co = n0 + pc * 8;
        const int m = P0 + row;
        if (m < a.M && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)row * SROW + pc * 8);
            if (a.res) {
                const uint4 r = *reinterpret_cast<const uint4 *>(a.res + (size_t)m * a.ldr + co);
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
            }
            *reinterpret_cast<uint4 *>(a.out + (size_t)m * a.ldo + co) = v;
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int NSLOT, int KS, int BK, bool UP = false, bool HEAD = false, int SUB = 1>
hipError_t launch_gbig1(ConvArgs &a, hipStream_t s)
{
    constexpr size_t ring = (size_t)NSLOT * SUB * (BM + BN) * BK * 2;
    constexpr size_t st = (size_t)BM * (BN + 8) * 2;
    constexpr size_t smem = ring > st ? ring : st;
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (hipError_t e = rva_func_smem((const void *)k_conv_gbig<BM, BN, WGM, WGN, NSLOT, KS, BK, UP, HEAD, SUB>, smem); e != hipSuccess) return e;
    a.n_tiles = rva_ceil_div(a.Cout, BN);
    // LDS-DMA addresses are a 64-bit scalar base + a 32-bit per-lane byte offset
    if ((size_t)a.H * a.W * (size_t)(a.M / (a.Ho * a.Wo) + 1) * (UP ? a.ldi2 : a.ldi) * 2 >= (1ull << 32) ||
        (size_t)a.CoutPad * KS * KS * a.Cin * 2 >= (1ull << 32))
        return hipErrorInvalidValue;
    if (a.M >= (1 << 24)) return hipErrorInvalidValue;             // div_s: raster positions below 2^24
    a.m_tiles = rva_ceil_div(a.M, BM);
    k_conv_gbig<BM, BN, WGM, WGN, NSLOT, KS, BK, UP, HEAD, SUB><<<rva_ceil_div(a.m_tiles, 8) * 8 * a.n_tiles, 512, smem, s>>>(a);
    return hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN, int NSLOT, int BK = 64, int SUB = 1>
hipError_t launch_gbig(ConvArgs &a, int ksize, hipStream_t s)
{
    if (a.Cin % BK || a.H > 2000 || a.W > 2000) return hipErrorInvalidValue;
    return ksize == 1 ? launch_gbig1<BM, BN, WGM, WGN, NSLOT, 1, BK, false, false, SUB>(a, s)
                      : launch_gbig1<BM, BN, WGM, WGN, NSLOT, 3, BK, false, false, SUB>(a, s);
}

// ---------------------------------------------------------------------------------------------------
// The first downsampling convolution (3x3 stride 2, Cin = 32, Cout <= 64) as a patch kernel.  It is the one strided
// layer whose weights (9 x 64 x 32 halfs = 36 KB) fit LDS for the whole launch, and at 320x320 the gather kernels
// re-fetch every input pixel ~2.25 times through the per-CU vector-memory path (96-100 us against an HBM roofline
// of 52 us).  Persistent blocks, two per CU: weights resident; per 4 x 32 output tile the 9 x 65 input patch is
// brought in ONCE by LDS-DMA (one patch buffer, reused as the output stage; the two blocks of a CU overlap each other).
// The patch is stored as two column-parity planes, LDS row = ((py*2 + (px & 1)) * 33 + (px >> 1)), so that the 16
// lanes of a B fragment (consecutive output x, i.e. input x of stride 2) read CONSECUTIVE 64-byte rows (swz32).
struct S2Args {
    const __half *in; int ldi;
    const __half *w; const float *bias;
    __half *out; int ldo;
    const __half *res; int ldr;
    int B, H, W, Ho, Wo, Cout, CoutPad, act, tiles_x, tiles_y, total, n_tiles;
};

template <int STRIDE, int CIN, int CO, int NWV, int NBUF, int RPW = 1>
__global__ void __launch_bounds__(NWV * 64) k_conv3_patch(S2Args a)
{
    // output tile = NWV*RPW rows x 32 pixels (RPW rows per wave: RPW = 2 halves the LDS reads per MFMA); STRIDE 2: patch (2*NWV+1) x 65 as two column-parity planes,
    // LDS row = (py*2 + (px & 1)) * 33 + (px >> 1);  STRIDE 1: patch (NWV+2) x 34, LDS row = py * 34 + px.
    // CIN 32: 64-byte rows (swz32), 16 rows per 1 KiB piece;  CIN 64: 128-byte rows (chunk ^ (row & 7)), 8 rows per piece.
    // NBUF 2: the next tile's patch is fetched under this tile's MFMAs and epilogue (one block per CU);  NBUF 1: one
    // patch buffer, several blocks per CU overlap each other instead.
    static_assert(CIN == 32 || CIN == 64, "input channels");
    constexpr int TH = NWV * RPW, TW = 32, NT = NWV * 64, FMW = 2 * RPW;
    constexpr int PH = STRIDE == 2 ? 2 * TH + 1 : TH + 2;
    constexpr int CW = STRIDE == 2 ? TW + 1 : TW + 2;                      // columns per (parity) plane
    constexpr int PROWS = STRIDE == 2 ? PH * 2 * CW : PH * CW;
    constexpr int RPP = 512 / CIN, LPR = CIN / 8;                          // rows per piece, lanes per row
    constexpr int PPIECES = (PROWS + RPP - 1) / RPP;
    constexpr int WPIECES = 9 * CO / RPP;
    constexpr int FN = CO / 16, KCH = CIN / 32;
    constexpr int SROW = CO + 8;
    static_assert(TH * TW * SROW * 2 <= PPIECES * 1024, "the output stage reuses a patch buffer");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *wl = (__half *)smem;                           // [9 taps][CO][CIN]   (swizzled rows)
    __half *patch0 = wl + WPIECES * 512;                   // [NBUF][PPIECES * RPP][CIN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane / LPR, lp = lane % LPR;
    const int tiles_img = a.tiles_x * a.tiles_y;
    // output channels are split into n_tiles groups of CO (each block keeps ITS group's weights resident): the layers whose
    // whole weight tensor leaves room for one block per CU only run as two half-width blocks per CU that overlap each
    // other's load / MFMA / SiLU / store phases
    const int n0 = (int)(blockIdx.x % a.n_tiles) * CO;
    const int bid = blockIdx.x / a.n_tiles, nblk = gridDim.x / a.n_tiles;
    // source chunk (x8 halfs) of LDS row `row`, physical chunk lp: inverse of the swizzle the fragment reads apply
    auto src_c8 = [&](int row) { return (CIN == 64 ? (lp ^ (row & 7)) : ((lp - 2 * (row >> 2)) & 3)) * 8; };
    // half offset of (row, 16-byte chunk c) inside a swizzled buffer
    auto lds_off = [&](int row, int c) { return CIN == 64 ? row * 64 + ((c ^ (row & 7)) << 3) : swz32(row, c); };

    float4 bvs[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bvs[i] = *reinterpret_cast<const float4 *>(a.bias + min(n0 + 16 * i + (lane >> 4) * 4, a.CoutPad - 4));

    // weights: once per block
#pragma unroll
    for (int k = 0; k < (WPIECES + NWV - 1) / NWV; ++k) {
        const int idx = wv + NWV * k;
        if (idx < WPIECES) {
            const int row = idx * RPP + lrow;                       // row = tap * CO + co
            const int tap = row / CO, co = min(n0 + row - tap * CO, a.CoutPad - 1);
            const __half *src = a.w + (size_t)(co * 9 + tap) * CIN + src_c8(row);
            __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(wl + idx * 512), 16, 0, 0);
        }
    }
    // the patch position of every LDS row this lane fills is the same for all tiles: decode it once
    constexpr int NPP = (PPIECES + NWV - 1) / NWV;
    int pdy[NPP], pdx[NPP], pc8[NPP];
#pragma unroll
    for (int k = 0; k < NPP; ++k) {
        const int row = (wv + NWV * k) * RPP + lrow;
        const int rc = min(row, PROWS - 1);
        const int pr = rc / CW, c = rc - pr * CW;                   // STRIDE 2: pr = py * 2 + parity
        pdy[k] = STRIDE == 2 ? pr >> 1 : pr;
        pdx[k] = STRIDE == 2 ? 2 * c + (pr & 1) : c;
        pc8[k] = src_c8(row);
    }
    auto issue_patch = [&](int t, int pb) {
        const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
        const int iy_base = STRIDE * ty * TH - 1, ix_base = STRIDE * tx * TW - 1;
        __half *dst = patch0 + (size_t)pb * PPIECES * 512;
        const int img = b * a.H;
#pragma unroll
        for (int k = 0; k < NPP; ++k) {
            const int idx = wv + NWV * k;
            if (idx < PPIECES) {
                const int iy = iy_base + pdy[k], ix = ix_base + pdx[k];
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int q = ok ? (img + iy) * a.W + ix : 0;
                const __half *src = a.in + (size_t)q * a.ldi + pc8[k];
                __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(dst + idx * 512), 16, 0, 0);
            }
        }
    };

    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 8;
    const bool st_on = tid == 0 && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 8;
    const int st_slot = st_on ? blockIdx.x / st_stride : 0;
    int st_n = 0;
#endif
    int pb = 0;
    if (NBUF == 2 && bid < a.total) issue_patch(bid, 0);
    for (int t = bid; t < a.total; t += nblk) {
        STAMP(0);
        if (NBUF == 1) issue_patch(t, 0);
        wait_vm<0>();
        STAMP(1);
        __builtin_amdgcn_s_barrier();        // this tile's patch (and the weights) are in LDS; the previous tile is done
        STAMP(2);
        if (NBUF == 2 && t + nblk < a.total) issue_patch(t + nblk, pb ^ 1);
        STAMP(3);
        const __half *patch = patch0 + (size_t)pb * PPIECES * 512;
        __half *stage = patch0 + (size_t)pb * PPIECES * 512;       // [TH*TW][SROW], once the patch is dead
        const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
        // tap validity of this lane's pixels (bit = dy * 3 + dx); fragment j = row (j >> 1) of this wave, half (j & 1)
        int vm[FMW];
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
            const int oy = ty * TH + wv * RPW + (j >> 1);
            const int ox = tx * TW + 16 * (j & 1) + (lane & 15);
            int m = 0;
            if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
                    if ((unsigned)(STRIDE * oy - 1 + tp / 3) < (unsigned)a.H && (unsigned)(STRIDE * ox - 1 + tp % 3) < (unsigned)a.W) m |= 1 << tp;
            }
            vm[j] = m;
        }
        f4 acc[FN][FMW];
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FMW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int dy = tp / 3, dx = tp % 3;
            // LDS row of this lane's pixel in wave-row r: STRIDE 2 -> patch row 2*(wv*RPW + r) + dy, parity plane dx & 1
            const int rb = (STRIDE == 2 ? ((2 * wv * RPW + dy) * 2 + (dx & 1)) * CW + (dx >> 1) : (wv * RPW + dy) * CW + dx) + (lane & 15);
            constexpr int ROWSTEP = STRIDE == 2 ? 4 * CW : CW;      // LDS rows between consecutive output rows
#pragma unroll
            for (int ks = 0; ks < KCH; ++ks) {
                h8 bf[FMW];
#pragma unroll
                for (int j = 0; j < FMW; ++j) {
                    bf[j] = *reinterpret_cast<const h8 *>(patch + lds_off(rb + (j >> 1) * ROWSTEP + 16 * (j & 1), ks * 4 + (lane >> 4)));
                    if (!((vm[j] >> tp) & 1)) bf[j] = hz;
                }
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const h8 af = *reinterpret_cast<const h8 *>(wl + lds_off(tp * CO + 16 * i + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
                    for (int j = 0; j < FMW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        STAMP(4);
        __syncthreads();                     // every wave is done reading the patch: reuse it as the output stage
        STAMP(5);
        // epilogue: bias + SiLU -> stage -> 16-byte row stores
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int co = 16 * i + (lane >> 4) * 4;
            const float4 bv = bvs[i];
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
                if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                const int px = (wv * RPW + (j >> 1)) * TW + 16 * (j & 1) + (lane & 15);
                __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                uint2 pk;
                pk.x = *reinterpret_cast<uint32_t *>(&lo);
                pk.y = *reinterpret_cast<uint32_t *>(&hi);
                *reinterpret_cast<uint2 *>(stage + (size_t)px * SROW + co) = pk;
            }
        }
        STAMP(6);
        __syncthreads();
        STAMP(7);
        const int cpr = min(CO, a.Cout - n0) >> 3;
        for (int q = tid; q < TH * TW * cpr; q += NT) {
            const int px = q / cpr, pc = q - px * cpr;
            const int yy = ty * TH + px / TW, xx = tx * TW + px % TW;
            if (yy < a.Ho && xx < a.Wo) {
                uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)px * SROW + pc * 8);
                const size_t m = (size_t)(b * a.Ho + yy) * a.Wo + xx;
                if (a.res) {
                    const uint4 r = *reinterpret_cast<const uint4 *>(a.res + m * a.ldr + n0 + pc * 8);
                    __half2 *vh = reinterpret_cast<__half2 *>(&v);
                    const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                        vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                    }
                }
                *reinterpret_cast<uint4 *>(a.out + m * a.ldo + n0 + pc * 8) = v;
            }
        }
        STAMP(8);
        if (NBUF == 1) __syncthreads();      // stage fully read before the next patch lands on it
        else pb ^= 1;                        // two buffers: the next top-of-loop barrier orders the stage reads
    }
}

template <int STRIDE, int CIN, int CO, int NWV, int NBUF, int RPW = 1>
hipError_t launch_patch(S2Args &g, int num_cus, hipStream_t s)
{
    constexpr int RPP = 512 / CIN, TH = NWV * RPW;
    constexpr int PROWS = STRIDE == 2 ? (2 * TH + 1) * 2 * 33 : (TH + 2) * 34;
    constexpr size_t smem = (size_t)(9 * CO / RPP + NBUF * ((PROWS + RPP - 1) / RPP)) * 1024;
    static_assert(smem <= 160 * 1024, "LDS budget");   // 163,840 B per CU
    constexpr int per_cu = smem <= 32 * 1024 ? 4 : smem <= 53 * 1024 ? 3 : smem <= 80 * 1024 ? 2 : 1;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_patch<STRIDE, CIN, CO, NWV, NBUF, RPW>, smem); e != hipSuccess) return e;
    g.tiles_x = rva_ceil_div(g.Wo, 32); g.tiles_y = rva_ceil_div(g.Ho, TH);
    g.total = g.tiles_x * g.tiles_y * g.B;
    g.n_tiles = rva_ceil_div(g.Cout, CO);
    int per_n = per_cu * num_cus / g.n_tiles;                  // persistent blocks per channel group
    if (per_n > g.total) per_n = g.total;
    if (per_n < 1) per_n = 1;
    const int grid = per_n * g.n_tiles;
    k_conv3_patch<STRIDE, CIN, CO, NWV, NBUF, RPW><<<grid, NWV * 64, smem, s>>>(g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The bottleneck of the first C2f of YOLOv8s -- y = x + SiLU(conv3x3(SiLU(conv3x3(x)))), 32 channels at 160 x 160 -- in ONE launch
// (round 4).  Layer by layer the 160 x 160 stage is bandwidth-bound: the two patch launches move 122 MB each (patch halo included)
// at what a device copy reaches.  Here the 32-channel intermediate never leaves LDS: per 4 x 32 output tile the (4 + 4) x 36 input
// patch comes in once by LDS-DMA (pixels outside the image fetch a zero line: the first convolution's padding), the first
// convolution runs over the (4 + 2) x 34 intermediate pixels the tile's outputs touch -- laid out with the INPUT patch's row pitch
// (36), so that a fragment of 16 consecutive intermediate rows reads 16 consecutive input rows at a fixed offset per tap; two junk
// columns per row are the price -- and writes them, SiLU applied and zero outside the image (the second convolution's padding), into
// a second LDS image; the second convolution reads its taps from there, adds the shortcut from the centre of the input patch
// and stores.  Reads 2.25 x 64 B + writes 64 B per output pixel instead of (1.33 + 1 + 1.33 + 1) x 64 B.  Same operations in the
// same order as the two patch launches (taps in order, fp32 accumulation, bias, SiLU, one rounding; shortcut added to the rounded
// value): bit-identical to them.  81 KB of LDS, two blocks per CU, four waves each.
__device__ __attribute__((aligned(64))) unsigned g_zero_line[16];

struct PairArgs {
    const __half *in; int ldi;
    const __half *w1; const float *b1; const __half *w2; const float *b2;
    __half *out; int ldo;
    int B, H, W, tiles_x, tiles_y, total;
};
constexpr int PR_TH = 4, PR_TW = 32, PR_PITCH = 36;
constexpr int PR_IPIECES = 19;            // input patch rows: (TH + 4) x 36 = 288 (+ what the junk intermediate rows read: < 304)
constexpr int PR_MFRAGS = 14;             // intermediate rows: (TH + 2) x 36 = 216 -> 14 fragments of 16
constexpr int PR_SROW = 32 + 8;
constexpr size_t PR_SMEM = (size_t)2 * 18 * 1024 + (size_t)PR_IPIECES * 1024 + (size_t)PR_MFRAGS * 1024 + (size_t)PR_TH * PR_TW * PR_SROW * 2;
static_assert(2 * PR_SMEM <= 160 * 1024, "two blocks per CU");

__global__ void __launch_bounds__(256) k_c2f_pair32(PairArgs a)
{
    constexpr int TH = PR_TH, TW = PR_TW, PITCH = PR_PITCH, NWV = 4, SROW = PR_SROW;
    constexpr int NIP = (PR_IPIECES + NWV - 1) / NWV, NMF = (PR_MFRAGS + NWV - 1) / NWV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *wl1 = (__half *)smem;                         // [9 taps][32][32], swizzled rows
    __half *wl2 = wl1 + 18 * 512;
    __half *inp = wl2 + 18 * 512;                         // [304][32]   row = py * 36 + px
    __half *mid = inp + PR_IPIECES * 512;                 // [224][32]   row = my * 36 + mx
    __half *stage = mid + PR_MFRAGS * 512;                // [128][SROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4, lrow = lane >> 2, lp = lane & 3;
    const int tiles_img = a.tiles_x * a.tiles_y;
    // both weight tensors, once per block (36 pieces)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int idx = wv + NWV * k, sub = idx % 18;
        const int row = sub * 16 + lrow, tap = row >> 5, co = row & 31;
        const __half *src = (idx < 18 ? a.w1 : a.w2) + (size_t)(co * 9 + tap) * 32 + ((lp - 2 * (row >> 2)) & 3) * 8;
        __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)((idx < 18 ? wl1 : wl2) + sub * 512), 16, 0, 0);
    }
    float4 bv1[2], bv2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bv1[i] = *reinterpret_cast<const float4 *>(a.b1 + 16 * i + q * 4);
        bv2[i] = *reinterpret_cast<const float4 *>(a.b2 + 16 * i + q * 4);
    }
    // the patch position of every input row this lane fills: the same for all tiles
    int ppy[NIP], ppx[NIP], pc8[NIP];
#pragma unroll
    for (int k = 0; k < NIP; ++k) {
        const int row = (wv + NWV * k) * 16 + lrow;
        ppy[k] = row < (TH + 4) * PITCH ? row / PITCH : 1 << 20;          // rows behind the patch: never inside an image
        ppx[k] = row % PITCH;
        pc8[k] = ((lp - 2 * (row >> 2)) & 3) * 8;
    }
    // intermediate position of this lane's row in each of the wave's fragments
    int mmy[NMF], mmx[NMF];
#pragma unroll
    for (int k = 0; k < NMF; ++k) {
        const int row = (wv + NWV * k) * 16 + n;
        mmy[k] = row / PITCH; mmx[k] = row % PITCH;
    }
    for (int t = blockIdx.x; t < a.total; t += gridDim.x) {
        const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        // ---- the input patch: rows y0 - 2 .. y0 + TH + 1, columns x0 - 2 .. x0 + 33
#pragma unroll
        for (int k = 0; k < NIP; ++k) {
            const int idx = wv + NWV * k;
            if (idx < PR_IPIECES) {
                const int iy = y0 - 2 + ppy[k], ix = x0 - 2 + ppx[k];
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const __half *src = ok ? a.in + ((size_t)(b * a.H + iy) * a.W + ix) * a.ldi + pc8[k] : reinterpret_cast<const __half *>(g_zero_line);
                __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(inp + idx * 512), 16, 0, 0);
            }
        }
        wait_vm<0>();
        __syncthreads();                                  // the patch (and, the first time, the weights) are in LDS
        // ---- first convolution over the intermediate rows of this wave's fragments
        f4 acc1[NMF][2];
#pragma unroll
        for (int k = 0; k < NMF; ++k) acc1[k][0] = acc1[k][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int toff = (tp / 3) * PITCH + tp % 3;
            h8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const h8 *>(wl1 + swz32(tp * 32 + 16 * i + n, q));
#pragma unroll
            for (int k = 0; k < NMF; ++k) {
                if (wv + NWV * k < PR_MFRAGS) {
                    const h8 bf = *reinterpret_cast<const h8 *>(inp + swz32((wv + NWV * k) * 16 + n + toff, q));
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc1[k][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc1[k][i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NMF; ++k) {
            if (wv + NWV * k < PR_MFRAGS) {
                const int row = (wv + NWV * k) * 16 + n;
                const int iy = y0 - 1 + mmy[k], ix = x0 - 1 + mmx[k];
                const bool inimg = mmx[k] < TW + 2 && mmy[k] < TH + 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 bv = bv1[i];
                    const float v0 = silu_f(acc1[k][i][0] + bv.x), v1 = silu_f(acc1[k][i][1] + bv.y), v2 = silu_f(acc1[k][i][2] + bv.z), v3 = silu_f(acc1[k][i][3] + bv.w);
                    __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                    uint2 pk;
                    pk.x = inimg ? *reinterpret_cast<uint32_t *>(&lo) : 0u;
                    pk.y = inimg ? *reinterpret_cast<uint32_t *>(&hi) : 0u;
                    *reinterpret_cast<uint2 *>(mid + swz32(row, 2 * i + (q >> 1)) + 4 * (q & 1)) = pk;
                }
            }
        }
        __syncthreads();                                  // the intermediate image is complete
        // ---- second convolution: wave wv = output row wv, two fragments of 16 pixels
        f4 acc2[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[j][0] = acc2[j][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int rb = (wv + tp / 3) * PITCH + tp % 3 + n;
            h8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const h8 *>(wl2 + swz32(tp * 32 + 16 * i + n, q));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const h8 bf = *reinterpret_cast<const h8 *>(mid + swz32(rb + 16 * j, q));
#pragma unroll
                for (int i = 0; i < 2; ++i) acc2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc2[j][i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 bv = bv2[i];
                const float v0 = silu_f(acc2[j][i][0] + bv.x), v1 = silu_f(acc2[j][i][1] + bv.y), v2 = silu_f(acc2[j][i][2] + bv.z), v3 = silu_f(acc2[j][i][3] + bv.w);
                __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                uint2 pk;
                pk.x = *reinterpret_cast<uint32_t *>(&lo);
                pk.y = *reinterpret_cast<uint32_t *>(&hi);
                *reinterpret_cast<uint2 *>(stage + (size_t)(wv * TW + 16 * j + n) * SROW + 16 * i + q * 4) = pk;
            }
        __syncthreads();
        // ---- + shortcut (the centre of the input patch), 16-byte row stores
        for (int e = tid; e < TH * TW * 4; e += 256) {
            const int px = e >> 2, pc = e & 3;
            const int ry = px / TW, rx = px - ry * TW;
            const int yy = y0 + ry, xx = x0 + rx;
            if (yy < a.H && xx < a.W) {
                uint4 v = *reinterpret_cast<const uint4 *>(stage + (size_t)px * SROW + pc * 8);
                const uint4 r = *reinterpret_cast<const uint4 *>(inp + swz32((ry + 2) * PITCH + rx + 2, pc));
                __half2 *vh = reinterpret_cast<__half2 *>(&v);
                const __half2 *rh = reinterpret_cast<const __half2 *>(&r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                    vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                }
                *reinterpret_cast<uint4 *>(a.out + ((size_t)(b * a.H + yy) * a.W + xx) * a.ldo + pc * 8) = v;
            }
        }
        __syncthreads();                                  // patch and stage are free for the next tile
    }
}

// ---------------------------------------------------------------------------------------------------
// Patch kernel, two wave sets half a tile period apart (3x3 stride 1, Cin = 64, Cout <= 64; weights resident in LDS).
//
// k_conv3_patch<1,64,64,8,1,2> runs its MFMA phase at 95 % of the matrix pipe's time, but per 16 x 32 tile that phase is
// 9.7 k of 27 k cycles: patch load (5 k), barriers (6 k), SiLU + stage (3 k) and stores (3 k) are all serial around it
// (tools/patch_stamps.py) -- the weights leave room for one block per CU only, so no second block overlaps them.  Here the
// block's waves are two SETS (SW waves each, RPW output rows per wave, 8 x 32 tiles); each set owns a patch buffer and walks
// its own tiles.  Both sets run the same loop
//     MFMAs of the tile | barrier | issue the next patch, epilogue of the tile, wait for the patch | barrier
// but set 1 starts one barrier later: whenever a SIMD's set-0 waves are in their MFMA phase the set-1 waves are in their load /
// epilogue phase and vice versa, so the matrix pipe is fed in both halves of the period and every barrier does double duty
// (patch free for one set, patch landed for the other).  The epilogue goes from registers to 16-byte stores
// (v_permlane16_swap between the two 16-pixel halves of a row), there is no LDS left for a stage.  Instantiated as 2 x 8 waves
// of 128 registers, one row per wave: an MFMA phase fed from LDS needs TWO waves per SIMD to run near the pipe's rate (one
// reaches ~55 % however its loads are software-pipelined: the 2 x 4 waves x two rows form measured that).
template <int CIN, int CO, int SW, int RPW>
__global__ void __launch_bounds__(2 * SW * 64) __attribute__((amdgpu_waves_per_eu(SW / 2, SW / 2))) k_conv3_patch2(S2Args a)
{
    static_assert(CIN == 64 && CO == 64, "64 -> 64 form");
    static_assert(SW * RPW == 8 && (SW == 4 || SW == 8), "8-row tiles: four waves x two rows or eight waves x one row per set");
    constexpr int NWAVES = 2 * SW, TH = SW * RPW, TW = 32, FMW = 2 * RPW, FN = CO / 16, KCH = CIN / 32;
    constexpr int PH = TH + 2, CW = TW + 2, PROWS = PH * CW;
    constexpr int RPP = 512 / CIN, LPR = CIN / 8;
    constexpr int PPIECES = (PROWS + RPP - 1) / RPP, NPP = (PPIECES + SW - 1) / SW;
    constexpr int WPIECES = 9 * CO / RPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *wl = (__half *)smem;                           // [9 taps][CO][CIN]   (swizzled rows)
    __half *patch0 = wl + WPIECES * 512;                   // [2 sets][PPIECES * RPP][CIN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wv / SW, ws = wv % SW;                 // wave set, wave within the set
    const int n = lane & 15, q = lane >> 4;
    const int lrow = lane / LPR, lp = lane % LPR;
    const int tiles_img = a.tiles_x * a.tiles_y;
    auto src_c8 = [&](int row) { return (lp ^ (row & 7)) * 8; };
    auto lds_off = [&](int row, int c) { return row * 64 + ((c ^ (row & 7)) << 3); };

    // weights: once per block, all eight waves
#pragma unroll
    for (int k = 0; k < (WPIECES + NWAVES - 1) / NWAVES; ++k) {
        const int idx = wv + NWAVES * k;
        if (idx < WPIECES) {
            const int row = idx * RPP + lrow;                       // row = tap * CO + co
            const int tap = row / CO, co = min(row - tap * CO, a.CoutPad - 1);
            const __half *src = a.w + (size_t)(co * 9 + tap) * CIN + src_c8(row);
            __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(wl + idx * 512), 16, 0, 0);
        }
    }
    // Patch pieces through lds_dma16: a wave-uniform 64-bit base (the patch's top-left pixel) + one 32-bit byte offset per lane,
    // tile-invariant and computed once; per tile only the validity of the lane's pixel is tested (lanes outside the image
    // re-read the tile's first output pixel, their taps are masked in the B fragments).  Through the builtin a piece cost
    // ~25 vector instructions of address arithmetic, issued beside the other set's MFMA phase.
    unsigned poff[NPP];
    int pgeo[NPP];                                         // dy << 8 | dx of the lane's patch pixel
    const unsigned ldi2 = (unsigned)a.ldi * 2u;
#pragma unroll
    for (int k = 0; k < NPP; ++k) {
        const int row = (ws + SW * k) * RPP + lrow;
        const int rc = min(row, PROWS - 1);
        const int dy = rc / CW, dx = rc - dy * CW;
        pgeo[k] = (dy << 8) | dx;
        poff[k] = (unsigned)(dy * a.W + dx) * ldi2 + (unsigned)src_c8(row) * 2u;
    }
    __half *patch = patch0 + (size_t)set * PPIECES * 512;  // this set's buffer
    const unsigned lds_patch = lds_addr(patch);
    auto issue_patch = [&](int t) {
        const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
        const int iy_base = ty * TH - 1, ix_base = tx * TW - 1;
        const char *base = (const char *)a.in + ((long long)(b * a.H + iy_base) * a.W + ix_base) * (long long)ldi2;
        const unsigned centre = (unsigned)(a.W + 1) * ldi2;
#pragma unroll
        for (int k = 0; k < NPP; ++k) {
            const int idx = ws + SW * k;
            if (idx < PPIECES) {
                const bool ok = (unsigned)(iy_base + (pgeo[k] >> 8)) < (unsigned)a.H && (unsigned)(ix_base + (pgeo[k] & 255)) < (unsigned)a.W;
                lds_dma16(ok ? poff[k] : centre, base, lds_patch + (unsigned)idx * 1024u);
            }
        }
    };
    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
    // Fragment addresses as a few per-lane bases + instruction immediates: every vector instruction beside the MFMAs costs
    // issue slots the matrix pipe's feeding competes for (an MFMA holds the port 8 of its 16 cycles), and a swizzled offset
    // computed per fragment was ~30 of them per 16 MFMAs.  B: the patch row of pixel n for wave-row r in 0 .. RPW+1 and
    // dx in 0..2 (the +16-pixel half is +1024 halfs, same swizzle; the second 32-channel half is the offset ^ 32);
    // A: row tap * CO + 16 i + n has (row & 7) = n & 7, so tap and i are immediates.
    int boff[RPW + 2][3];
#pragma unroll
    for (int r = 0; r < RPW + 2; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) boff[r][dx] = lds_off((ws * RPW + r) * CW + dx + n, q);
    const int aoff = lds_off(n, q);
    float *bias_l = reinterpret_cast<float *>(patch0 + (size_t)2 * PPIECES * 512);      // [CO]: registers are scarce, a global read per tile would expose its latency
    if (tid < CO) bias_l[tid] = a.bias[min(tid, a.CoutPad - 1)];
    // tiles of this block: t = blockIdx.x + j * gridDim.x, j = 0, 1, ...; set s takes j = s, s + 2, ...
    const int nblk_tiles = (int)blockIdx.x < a.total ? (a.total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int nmine = (nblk_tiles - set + 1) / 2;          // tiles of this set
    const int niter = (nblk_tiles + 1) / 2;                // iterations both sets run (set 0 has the larger share)
#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 4;
    const bool st_on = (tid == 0 || tid == SW * 64) && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 4;
    const int st_slot = st_on ? (blockIdx.x / st_stride) * 2 + (tid >= SW * 64 ? 1 : 0) : 0;
    int st_n = 0;
#endif
    if (nmine > 0) issue_patch(blockIdx.x + set * gridDim.x);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();                          // weights and both first patches are in LDS
    if (set == 1) __builtin_amdgcn_s_barrier();            // half a period behind set 0
    for (int it = 0; it < niter; ++it) {
        const bool live = it < nmine;
        const int t = blockIdx.x + (2 * it + set) * gridDim.x;
        const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
        STAMP(0);
        f4 acc[FN][FMW];
        if (live) {
            int vm[FMW];
#pragma unroll
            for (int j = 0; j < FMW; ++j) {
                const int oy = ty * TH + ws * RPW + (j >> 1), ox = tx * TW + 16 * (j & 1) + n;
                int m = 0;
                if (oy < a.Ho && ox < a.Wo) {
                    const int hv = (ox >= 1 ? 1 : 0) | 2 | (ox <= a.W - 2 ? 4 : 0);
                    m = (oy >= 1 ? hv : 0) | (hv << 3) | (oy <= a.H - 2 ? hv << 6 : 0);
                }
                vm[j] = m;
            }
            // taps that some lane of this wave has to zero (wave-uniform): most tiles need none, the selects are skipped
            int vall = 0x1ff;
#pragma unroll
            for (int j = 0; j < FMW; ++j) vall &= vm[j];
            int need = 0;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                if (__builtin_amdgcn_ballot_w64(!((vall >> tp) & 1)) != 0) need |= 1 << tp;
            need = __builtin_amdgcn_readfirstlane(need);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FMW; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
            // group g = (tap, 32-channel half): FMW B + FN A fragments, FMW x FN MFMAs.  This wave is the only one of its SIMD in
            // the MFMA phase: the next group's fragments are read while this group's MFMAs run, and the padding taps are
            // zeroed where a fragment is consumed (a select behind the load would wait for it at once).
            constexpr int NG = 9 * KCH;
            h8 bfa[FMW], afa[FN], bfb[FMW], afb[FN];
#define P2_LOAD(BF, AF, G)                                                                                          \
            do {                                                                                                    \
                constexpr int tp_ = (G) / KCH, ks_ = (G) % KCH, dy_ = tp_ / 3, dx_ = tp_ % 3;                       \
                _Pragma("unroll") for (int j = 0; j < FMW; ++j)                                                     \
                    BF[j] = *reinterpret_cast<const h8 *>(patch + (boff[dy_ + (j >> 1)][dx_] ^ (ks_ * 32)) + (j & 1) * 1024); \
                _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                      \
                    AF[i] = *reinterpret_cast<const h8 *>(wl + (aoff ^ (ks_ * 32)) + (tp_ * CO + 16 * i) * 64);     \
            } while (0)
#define P2_MFMA(BF, AF, G)                                                                                          \
            do {                                                                                                    \
                constexpr int tp_ = (G) / KCH;                                                                      \
                if ((need >> tp_) & 1) {                                                                            \
                    _Pragma("unroll") for (int j = 0; j < FMW; ++j)                                                 \
                        if (!((vm[j] >> tp_) & 1)) BF[j] = hz;                                                      \
                }                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < FN; ++i)                                                      \
                    _Pragma("unroll") for (int j = 0; j < FMW; ++j)                                                 \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[i], BF[j], acc[i][j], 0, 0, 0);       \
            } while (0)
            static_assert(NG == 18, "the group loop below is written out for nine taps x two channel halves");
#define P2_PAIR(G)   P2_LOAD(bfb, afb, (G) + 1); P2_MFMA(bfa, afa, (G)); P2_LOAD(bfa, afa, (G) + 2); P2_MFMA(bfb, afb, (G) + 1)
#define P2_ONE(G)    P2_LOAD(bfa, afa, (G)); P2_MFMA(bfa, afa, (G))
            if constexpr (RPW == 2) {
                // one MFMA wave of this set per SIMD: the next group's fragments are read under this group's MFMAs
                P2_LOAD(bfa, afa, 0);
                P2_PAIR(0); P2_PAIR(2); P2_PAIR(4); P2_PAIR(6); P2_PAIR(8); P2_PAIR(10); P2_PAIR(12); P2_PAIR(14);
                P2_LOAD(bfb, afb, 17); P2_MFMA(bfa, afa, 16); P2_MFMA(bfb, afb, 17);
            } else {
                // two MFMA waves of this set per SIMD cover each other's fragment latency; 128 registers per wave
                P2_ONE(0); P2_ONE(1); P2_ONE(2); P2_ONE(3); P2_ONE(4); P2_ONE(5); P2_ONE(6); P2_ONE(7); P2_ONE(8);
                P2_ONE(9); P2_ONE(10); P2_ONE(11); P2_ONE(12); P2_ONE(13); P2_ONE(14); P2_ONE(15); P2_ONE(16); P2_ONE(17);
                (void)bfb; (void)afb;
            }
#undef P2_ONE
#undef P2_PAIR
#undef P2_LOAD
#undef P2_MFMA
        }
        STAMP(1);
        __builtin_amdgcn_s_barrier();        // this set has left its patch (the other set: its next patch has landed)
        STAMP(2);
        if (it + 1 < nmine) issue_patch(t + 2 * gridDim.x);
        STAMP(3);
        if (live) {
            // bias + SiLU in registers, one v_permlane16_swap per packed dword between the two 16-pixel halves of a row: lane
            // (n, q) ends up with eight consecutive channels of pixel n of half (q & 1), channels 16 i + 8 (q >> 1) ..
            const int qh = q & 1, qc = q >> 1;
            const int ox = tx * TW + 16 * qh + n;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int oy = ty * TH + ws * RPW + r;
                const bool pv = oy < a.Ho && ox < a.Wo;
                const size_t m = (size_t)(b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias_l + 16 * i + q * 4);
                    uint32_t w2[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f4 c = acc[i][2 * r + h];
                        float v0 = c[0] + bv.x, v1 = c[1] + bv.y, v2 = c[2] + bv.z, v3 = c[3] + bv.w;
                        if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3); }
                        __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                        w2[h][0] = *reinterpret_cast<uint32_t *>(&lo);
                        w2[h][1] = *reinterpret_cast<uint32_t *>(&hi);
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(w2[0][0], w2[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(w2[0][1], w2[1][1], false, false);
                    uint4 v = {s0[0], s1[0], s0[1], s1[1]};
                    const int co = 16 * i + 8 * qc;
                    if (pv && co < a.Cout) {
                        if (a.res) {
                            const uint4 rr = *reinterpret_cast<const uint4 *>(a.res + m * a.ldr + co);
                            __half2 *vh = reinterpret_cast<__half2 *>(&v);
                            const __half2 *rh = reinterpret_cast<const __half2 *>(&rr);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float2 x = __half22float2(vh[u]), y = __half22float2(rh[u]);
                                vh[u] = __floats2half2_rn(x.x + y.x, x.y + y.y);
                            }
                        }
                        *reinterpret_cast<uint4 *>(a.out + m * a.ldo + co) = v;
                    }
                }
            }
        }
        STAMP(4);
        wait_vm<0>();                        // the next patch (issued above) has landed; the stores have left too
        STAMP(5);
        __builtin_amdgcn_s_barrier();        // ... for every wave of this set (the other set: it has left ITS patch)
    }
    if (set == 0) __builtin_amdgcn_s_barrier();            // the barrier set 1 took ahead of its loop
}

template <int CIN, int CO, int SW, int RPW>
hipError_t launch_patch2(S2Args &g, int num_cus, hipStream_t s)
{
    constexpr int RPP = 512 / CIN, TH = 8;
    constexpr int PROWS = (TH + 2) * 34;
    constexpr size_t smem = (size_t)(9 * CO / RPP + 2 * ((PROWS + RPP - 1) / RPP)) * 1024 + CO * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (g.Cout > CO || g.Cout % 8) return hipErrorInvalidValue;
    if (hipError_t e = rva_func_smem((const void *)k_conv3_patch2<CIN, CO, SW, RPW>, smem); e != hipSuccess) return e;
    g.tiles_x = rva_ceil_div(g.Wo, 32); g.tiles_y = rva_ceil_div(g.Ho, TH);
    g.total = g.tiles_x * g.tiles_y * g.B;
    g.n_tiles = 1;
    int grid = num_cus;
    if (grid > g.total) grid = g.total;
    k_conv3_patch2<CIN, CO, SW, RPW><<<grid, 2 * SW * 64, smem, s>>>(g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Stem + first downsampling convolution in ONE launch, for the YOLOv8s widths (3 -> 32 -> 64, both 3x3 stride 2, SiLU).
//
// As two launches the pair moves 78 MB in, 210 MB out (stem), 210 MB in again and 105 MB out: 600 MB per 32-frame batch,
// 160 us of an HBM-bound 1.9 ms tick.  Fused, the 32-channel half-resolution tensor never leaves the CU.  Persistent
// blocks (one per CU, sixteen waves) walk 4 x 32 output tiles as a two-stage pipeline of wave groups, two waves of each
// group per SIMD, ONE workgroup barrier per tile:
//   waves 0-7 (stage S, vector-ALU bound: 585 x 32 SiLUs per tile)   tile k+1
//      the 3 x 19 x 136 planar input patch (prefetched into registers one tile ahead) sits in LDS; the 9 x 65 stem pixels
//      under the tile are computed by MFMA straight from it -- no im2col buffer: the K order is  k' = 4 * (c*3 + ky) + s
//      with s = 0 a zero-weight slot and s = 1..3 = kx 0..2, so a pixel's four slots of a (c, ky) row are two aligned
//      dwords of the patch row and a lane's eight K values are two such rows (K = 36 padded to 64: a second MFMA whose only
//      live row is (c, ky) = (2, 2)); bias + SiLU, stem pixels outside the image written as zeros (they are the second
//      convolution's padding), into the column-parity patch layout of k_conv3_patch<2, 32, 64>;
//   waves 8-15 (stage C, MFMA bound: 36 MFMAs per wave)             tile k
//      the second convolution on the patch the other group finished one tile ago, exactly as k_conv3_patch does it
//      (weights resident, tap-major), then bias + SiLU -> a wave-private stage -> 16-byte row stores.
// Patch and input buffers are double-buffered between the groups.
struct Stem2Args {
    const __half *in; const __half *w1; const float *b1; const __half *w2; const float *b2; __half *out; int ldo;
    int B, H, W, H1, W1, Ho, Wo, tiles_x, tiles_y, total;
};

constexpr int S2_TH = 4, S2_TW = 32;
constexpr int S2_PROWS = (2 * S2_TH + 1) * 2 * (S2_TW + 1);                  // 9 x 65 stem pixels as two column-parity planes
constexpr int S2_IR = 4 * S2_TH + 3, S2_IC = 136;                            // input patch: rows per plane, halfs per row
constexpr int S2_SROW = 64 + 8;
constexpr size_t S2_SMEM = (size_t)36 * 1024 + 2 * (size_t)(S2_PROWS + 1) * 64 + 2 * (size_t)3 * S2_IR * S2_IC * 2 + (size_t)S2_TH * S2_TW * S2_SROW * 2;
static_assert(S2_SMEM <= 160 * 1024, "LDS budget");

__global__ void __launch_bounds__(1024) k_stem2(Stem2Args a)
{
    constexpr int TH = S2_TH, TW = S2_TW, CW = TW + 1, PROWS = S2_PROWS, WPIECES = 9 * 64 / 16;
    constexpr int IR = S2_IR, IC = S2_IC;
    constexpr int NCH = 3 * IR * (IC / 8), NPV = (NCH + 511) / 512;           // 16-byte chunks of the input patch, per thread of group S
    constexpr int SPX = (2 * TH + 1) * (2 * TW + 1), NFR = (SPX + 15) / 16;   // stem pixels / fragments of a patch
    constexpr int SROW = S2_SROW, FN = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *wl = (__half *)smem;                           // [9 taps][64][32]   second convolution, swizzled rows
    __half *patch0 = wl + WPIECES * 512;                   // [2][PROWS + 1][32] stem output (+ a dump row for lanes without a pixel)
    __half *inp0 = patch0 + 2 * (PROWS + 1) * 32;                // [2][3][IR][IC]     planar input patch
    __half *stage0 = inp0 + 2 * 3 * IR * IC;               // [TH*TW][SROW]      output stage, rows private to a wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int nk = ((int)a.total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this block

    // second convolution's weights: once per block, by LDS-DMA (row = tap * 64 + co)
    {
        const int lrow = lane >> 2, lp = lane & 3;
#pragma unroll
        for (int k = 0; k < (WPIECES + 15) / 16; ++k) {
            const int idx = wv + 16 * k;
            if (idx < WPIECES) {
                const int row = idx * 16 + lrow;
                const int tap = row >> 6, co = row & 63;
                const __half *src = a.w2 + (size_t)(co * 9 + tap) * 32 + ((lp - 2 * (row >> 2)) & 3) * 8;
                __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)(wl + idx * 512), 16, 0, 0);
            }
        }
    }
#ifdef RVA_ROW_STAMPS
    const int st_stride = gridDim.x / 8;
    const bool st_on = (tid == 0 || tid == 512) && st_stride > 0 && blockIdx.x % st_stride == 0 && blockIdx.x / st_stride < 4;
    const int st_slot = st_on ? (blockIdx.x / st_stride) * 2 + (tid >> 9) : 0;
    int st_n = 0;
#endif

    if (wv < 8) {
        // ---------------- stage S: input patch -> stem pixels -> patch[k & 1] ----------------
        // stem weights as A fragments in registers.  API column order of w1 (see rva_stem_conv_f16): k = 2j + kx for kx in
        // {0,1}, 18 + j for kx = 2, j = c*3 + ky.  The bias is the accumulator's initial value.
        h8 af1[2], af2[2];
        f4 bias1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const __half *wr = a.w1 + (size_t)(16 * i + n) * 32;
            const __half zero = __float2half(0.f);
            __half t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 2 * q + (e >> 2), sl = e & 3, kx = sl - 1;
                t1[e] = sl == 0 ? zero : wr[kx < 2 ? 2 * j + kx : 18 + j];
                t2[e] = (q == 0 && e >= 1 && e < 4) ? wr[e - 1 < 2 ? 16 + (e - 1) : 26] : zero;
            }
            af1[i] = *reinterpret_cast<const h8 *>(t1);
            af2[i] = *reinterpret_cast<const h8 *>(t2);
#pragma unroll
            for (int u = 0; u < 4; ++u) bias1[i][u] = a.b1[16 * i + 4 * q + u];
        }
        // this lane's two (c, ky) rows of the first K step, as half offsets into the input patch
        const int ro0 = (((2 * q) / 3) * IR + (2 * q) % 3) * IC, ro1 = (((2 * q + 1) / 3) * IR + (2 * q + 1) % 3) * IC;
        constexpr int ro8 = (2 * IR + 2) * IC;
        u4 pv[NPV];
        auto prefetch = [&](int t) {
            const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
            const int iy0 = 4 * ty * TH - 3, x00 = 4 * tx * TW - 8;
#pragma unroll
            for (int k = 0; k < NPV; ++k) {
                const int c16 = tid + 512 * k;
                const int row = c16 / 17, cx = c16 - row * 17;
                const int c = row / IR, r = row - c * IR;
                const int iy = iy0 + r, x0 = x00 + cx * 8;
                pv[k] = u4{0u, 0u, 0u, 0u};
                if (c16 < NCH && (unsigned)iy < (unsigned)a.H && (unsigned)x0 < (unsigned)a.W)
                    pv[k] = *reinterpret_cast<const u4 *>(a.in + ((size_t)(b * 3 + c) * a.H + iy) * a.W + x0);
            }
        };
        auto stage_input = [&](int buf) {
            __half *inp = inp0 + (size_t)buf * 3 * IR * IC;
#pragma unroll
            for (int k = 0; k < NPV; ++k) {
                const int c16 = tid + 512 * k;
                if (c16 < NCH) *reinterpret_cast<u4 *>(inp + (size_t)(c16 / 17) * IC + (c16 % 17) * 8) = pv[k];
            }
        };
        // geometry of this lane's pixel in each of the wave's fragments (the same for every tile): read offset into the input
        // patch, (py << 8 | px), and the write offset of channel group 0 in the stem patch (group 1 = that ^ 16)
        constexpr int FPW = (NFR + 7) / 8, UNR = FPW;
        static_assert(FPW % UNR == 0, "fragments per wave");
        int frd[FPW], fpp[FPW], fw[FPW];
#pragma unroll
        for (int e = 0; e < FPW; ++e) {
            const int f = wv + 8 * e;
            const int p = min(16 * f + n, SPX - 1);
            const int py = p / (2 * TW + 1), px = p - py * (2 * TW + 1);
            const int prow = (py * 2 + (px & 1)) * CW + (px >> 1);
            frd[e] = (2 * py) * IC + 2 * px + 4;
            fpp[e] = (py << 8) | px;
            fw[e] = 16 * f + n < SPX ? swz32(prow, q >> 1) + 4 * (q & 1) : PROWS * 32 + 4 * (q & 1);      // no pixel: the dump row behind the patch
        }
        if (nk > 0) { prefetch(blockIdx.x); stage_input(0); }
        __syncthreads();                                       // input patch of the first tile visible to the group
        for (int k = 0; k <= nk; ++k) {                        // iteration k: this group works on tile k, the other on tile k - 1
            STAMP(0);
            if (k < nk) {
                const int t = blockIdx.x + k * gridDim.x;
                const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
                (void)b;
                if (k + 1 < nk) prefetch(t + gridDim.x);
                const __half *inp = inp0 + (size_t)(k & 1) * 3 * IR * IC;
                __half *patch = patch0 + (size_t)(k & 1) * (PROWS + 1) * 32;
                const int sy0 = 2 * ty * TH - 1, sx0 = 2 * tx * TW - 1;
                // fragments in pairs: the chain  LDS read -> 2 dependent MFMAs -> exp / rcp -> LDS write  of one fragment is
                // ~700 cycles of latency (tools/stem2_stamps.py); independent chains overlap
                const int pyb = -sy0, pxb = -sx0;                     // first in-image patch row / column
                const int pye = a.H1 - sy0, pxe = a.W1 - sx0;         // one past the last
#pragma unroll
                for (int g = 0; g < FPW / UNR; ++g) {
                    h8 bf1[UNR], bf2[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const __half *base = inp + frd[g * UNR + u];
                        const uint32_t *d0 = reinterpret_cast<const uint32_t *>(base + ro0), *d1 = reinterpret_cast<const uint32_t *>(base + ro1);
                        const uint32_t *d8 = reinterpret_cast<const uint32_t *>(base + ro8);
                        uint32_t b1w[4] = {d0[0], d0[1], d1[0], d1[1]};
                        uint32_t b2w[4] = {d8[0], d8[1], 0u, 0u};       // lanes q > 0 carry zero weights in af2: their (finite) data is ignored
                        bf1[u] = *reinterpret_cast<const h8 *>(b1w);
                        bf2[u] = *reinterpret_cast<const h8 *>(b2w);
                    }
                    f4 sacc[UNR][2];
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int i = 0; i < 2; ++i) sacc[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[i], bf1[u], bias1[i], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int i = 0; i < 2; ++i) sacc[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af2[i], bf2[u], sacc[u][i], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int py = fpp[g * UNR + u] >> 8, px = fpp[g * UNR + u] & 255;
                        const bool inimg = py >= pyb && py < pye && px >= pxb && px < pxe;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const float v0 = silu_f(sacc[u][i][0]), v1 = silu_f(sacc[u][i][1]), v2 = silu_f(sacc[u][i][2]), v3 = silu_f(sacc[u][i][3]);
                            __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                            uint2 pk;
                            pk.x = inimg ? *reinterpret_cast<uint32_t *>(&lo) : 0u;
                            pk.y = inimg ? *reinterpret_cast<uint32_t *>(&hi) : 0u;
                            *reinterpret_cast<uint2 *>(patch + (fw[g * UNR + u] ^ (16 * i))) = pk;     // lanes without a pixel: the dump row
                        }
                    }
                }
                STAMP(1);
                if (k + 1 < nk) stage_input((k + 1) & 1);      // its last readers finished in iteration k - 1
            } else { STAMP(1); }
            STAMP(2);
            __syncthreads();
            STAMP(3);
        }
    } else {
        // ---------------- stage C: patch[(k - 1) & 1] -> second convolution -> output tile ----------------
        const int wb = wv - 8, orow = wb >> 1, ohalf = wb & 1;          // output row of the tile / 16-pixel half of it
        float4 bvs[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) bvs[i] = *reinterpret_cast<const float4 *>(a.b2 + 16 * i + q * 4);
        wait_vm<0>();                                          // this wave's share of the weights
        __syncthreads();
        __half *stage = stage0 + (size_t)wb * 16 * SROW;
        for (int k = 0; k <= nk; ++k) {
            STAMP(0);
            if (k > 0) {
                const int t = blockIdx.x + (k - 1) * gridDim.x;
                const int b = t / tiles_img, r2 = t - b * tiles_img, ty = r2 / a.tiles_x, tx = r2 - ty * a.tiles_x;
                const __half *patch = patch0 + (size_t)((k - 1) & 1) * (PROWS + 1) * 32;
                f4 acc[FN];
#pragma unroll
                for (int i = 0; i < FN; ++i) acc[i] = f4{bvs[i].x, bvs[i].y, bvs[i].z, bvs[i].w};
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int dy = tp / 3, dx = tp % 3;
                    const int rb = ((2 * orow + dy) * 2 + (dx & 1)) * CW + (dx >> 1) + 16 * ohalf + n;
                    const h8 bf = *reinterpret_cast<const h8 *>(patch + swz32(rb, q));
#pragma unroll
                    for (int i = 0; i < FN; ++i) {
                        const h8 af = *reinterpret_cast<const h8 *>(wl + swz32(tp * 64 + 16 * i + n, q));
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[i], 0, 0, 0);
                    }
                }
                STAMP(1);
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const int co = 16 * i + q * 4;
                    const float v0 = silu_f(acc[i][0]), v1 = silu_f(acc[i][1]), v2 = silu_f(acc[i][2]), v3 = silu_f(acc[i][3]);
                    __half2 lo = __floats2half2_rn(v0, v1), hi = __floats2half2_rn(v2, v3);
                    uint2 pk;
                    pk.x = *reinterpret_cast<uint32_t *>(&lo);
                    pk.y = *reinterpret_cast<uint32_t *>(&hi);
                    *reinterpret_cast<uint2 *>(stage + (size_t)n * SROW + co) = pk;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the stage rows are this wave's own: no barrier, just ordering
                const int yy = ty * TH + orow;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int c16 = lane + 64 * it;
                    const int opx = c16 >> 3, pc = c16 & 7;
                    const int xx = tx * TW + 16 * ohalf + opx;
                    if (yy < a.Ho && xx < a.Wo)
                        *reinterpret_cast<uint4 *>(a.out + ((size_t)(b * a.Ho + yy) * a.Wo + xx) * a.ldo + pc * 8) =
                            *reinterpret_cast<const uint4 *>(stage + (size_t)opx * SROW + pc * 8);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // stage reads done before the next tile's stage writes
            } else { STAMP(1); }
            STAMP(2);
            __syncthreads();
            STAMP(3);
        }
    }
}

}  // namespace

#define RVA_CONV_VARIANTS 89
#ifdef RVA_EXPERIMENTS
#define RVA_CONV_VARIANTS_MAX 99      // 96..: timing-only experiment kernels of a private build (tools/exp_build.py), never in librva.so
#else
#define RVA_CONV_VARIANTS_MAX RVA_CONV_VARIANTS
#endif

extern "C" {
#ifdef RVA_ROW_STAMPS
int rva_dbg_read_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * 256); }
#endif

// variant: 0 = heuristic choice; otherwise an explicit kernel (used by the plan's per-layer autotune):
//   1..4  gather kernel  <BN,WPX> = <64,64> <64,32> <128,64> <128,32>
//   5..8  resident kernel <BN,WPX> = <64,64> <64,32> <128,64> <128,32>   (stride 1 only)
//   9..12 row-reuse kernel, same tile order                               (3x3 stride 1 only)
//   13..16 gather kernel with 64-channel K-steps, same tile order         (Cin padded to 64)
//   17..20 row-reuse kernel with two K-steps of loads in flight           (3x3 stride 1 only)
//   21..24 large-tile LDS-DMA kernel <BM,BN> = <256,128> <128,128> <256,64> <128,64>, 3-slot ring (3x3 stride 1, Cin % 32 == 0)
//   25..32 the same with a 2-slot ring, <192,128> <128,128> <256,64> <128,64> <224,128> <160,128> <384,64> <320,64>:
//          two or three blocks per CU; the extra tile heights exist so that the tile count can fit whole rounds of the 256 CUs
//   33..38 large-tile LDS-DMA gather kernel with 64-channel K-steps (1x1; 3x3 stride 1 or 2; Cin % 64 == 0):
//          <256,128> <128,128> <256,64> <128,64> 3-slot, <128,128> <256,64> <192,128> 2-slot
//   40..42 the same with 32-channel K-steps (Cin % 32 == 0): <256,64> 3-slot, <128,64> 3-slot, <256,64> 2-slot
//   43..45 patch kernels for Cin = 32 (weights resident, input patch staged once per tile): 3x3 stride 2 with Cout <= 64;
//          3x3 stride 1 with Cout <= 32, 4- and 8-row tiles
//   46..51 patch kernels for 3x3 stride 1, Cin = 64, Cout <= 64 (double-buffered 4-row tile, 8-row tile, 4-row tile,
//          double-buffered 8-row tile, and two forms with two output rows per wave)
//   52..60 "long run" kernels: a 32-channel chunk's activation run staged once for the three vertical taps (3x3 stride 1)
//   61..63 patch kernels for Cin = 64 with the output channels in two resident groups of 32
//   64..65 LDS-DMA gather kernel with 256 x 256 tiles (64 MACs per staged byte; one block per CU): wave tile 64 x 128 / 128 x 64
//   66     patch kernel with two wave sets half a tile period apart (3x3 stride 1, Cin = 64, Cout <= 64): 2 x 8 waves, one output row per wave
//   67..73 "whole chunk per barrier" kernels for layers with few pixels (3x3 stride 1, Cin % 32 == 0): <BM,BN> = <128,64> <64,64>
//          <256,64> <64,96> <128,96> <192,64> <256,96>; one barrier per 32-channel chunk (nine taps), one block per CU
//   74..79 LDS-DMA gather kernel (as 33..39, two-slot ring) with SUB 64-channel K-steps per barrier: <BM,BN>xSUB = <128,128>x2 <128,64>x2
//          <128,64>x3 <64,64>x4 <64,128>x3 <64,64>x2
//   80..85 "long run" kernels on the padded raster (k_conv3_run<..., PADO>: no padding selects): <256,64> <256,128> <224,128> <320,64>
//          <192,128> <128,64>
int rva_conv2d_nhwc_f16_v(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias, void *out,
                          int ldo, const void *residual, int ldr, int batch, int H, int W, int Cin, int Cout, int ksize,
                          int stride, int act, int variant, rva_stream_t stream_);

int rva_conv2d_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias, void *out,
                        int ldo, const void *residual, int ldr, int batch, int H, int W, int Cin, int Cout, int ksize,
                        int stride, int act, rva_stream_t stream_)
{
    return rva_conv2d_nhwc_f16_v(ctx, in, ldi, weights, bias, out, ldo, residual, ldr, batch, H, W, Cin, Cout, ksize, stride,
                                 act, 0, stream_);
}

int rva_conv2d_nhwc_f16_v(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias, void *out,
                          int ldo, const void *residual, int ldr, int batch, int H, int W, int Cin, int Cout, int ksize,
                          int stride, int act, int variant, rva_stream_t stream_)
{
    if (!ctx) return RVA_ERR_ARG;
    hipStream_t s = (hipStream_t)stream_;
    if (!in || !weights || !bias || !out || batch <= 0 || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3) ||
        (stride != 1 && stride != 2) || Cin % 8 || Cout % 8 || ldi % 8 || ldo % 8 || (residual && ldr % 8) || variant < 0 || variant > RVA_CONV_VARIANTS_MAX ||
        ((uintptr_t)in | (uintptr_t)out | (uintptr_t)weights | (uintptr_t)residual) % 16)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_conv2d_nhwc_f16: unsupported shape/alignment (Cin%%8, Cout%%8, ld%%8, 16-byte pointers)");
    ConvArgs a{};
    a.in = (const __half *)in; a.ldi = ldi; a.w = (const __half *)weights; a.bias = bias;
    a.out = (__half *)out; a.ldo = ldo; a.res = (const __half *)residual; a.ldr = ldr;
    a.H = H; a.W = W; a.Cin = Cin; a.CinPad = rva_ceil_div(Cin, 32) * 32; a.Cout = Cout; a.stride = stride; a.act = act;
    const int pad = ksize / 2;
    a.Ho = (H + 2 * pad - ksize) / stride + 1;
    a.Wo = (W + 2 * pad - ksize) / stride + 1;
    a.M = batch * a.Ho * a.Wo;
    // tile choice: weights are padded to a multiple of 64 output channels by the caller (rva_conv_cout_pad)
    const int cpad = rva_ceil_div(Cout, 64) * 64;
    const bool bn128 = cpad % 128 == 0;
    if (!ctx->num_cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
        if (ctx->num_cus <= 0) ctx->num_cus = 256;
    }
    const int num_cus = ctx->num_cus;
    if (variant == 0) {
        // heuristic (callers that do not autotune): the LDS-DMA kernels wherever their channel constraints hold,
        // tile picked from the autotune tables of the YOLOv8 layers (tools/show_tuning.py)
        int pick = 0;
        if (ksize == 3 && stride == 1 && Cin % 32 == 0)
            pick = cpad % 128 == 0 ? ((long)a.M * Cout >= 20000000L ? 25 : 21) : (a.M >= 100000 ? 31 : 23);
        else if (Cin % 64 == 0 && (ksize == 3 || stride == 1))
            pick = 37;
        if (pick) {
            const int rc = rva_conv2d_nhwc_f16_v(ctx, in, ldi, weights, bias, out, ldo, residual, ldr, batch, H, W, Cin, Cout, ksize,
                                                 stride, act, pick, stream_);
            if (rc == RVA_OK) return rc;
        }
    }
    if (variant >= 86 && variant <= 89) {
        // stride-2 "long run" kernels on the padded output raster: see k_conv3_s2run
        hipError_t ev = hipErrorInvalidValue;
        if (ksize == 3 && stride == 2) {
            RunArgs g{};
            g.in = a.in; g.ldi = ldi; g.w = a.w; g.bias = bias; g.out = a.out; g.ldo = ldo; g.res = a.res; g.ldr = ldr;
            g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CoutPad = rva_ceil_div(Cout, 64) * 64; g.act = act; g.M = a.M;
            switch (variant) {
            case 86: ev = launch_s2run<256, 128, 4, 2>(g, s); break;     // 147 KB
            case 87: ev = launch_s2run<256, 64, 4, 2>(g, s); break;      // 123 KB
            case 88: ev = launch_s2run<128, 128, 2, 4>(g, s); break;     // 99 KB
            default: ev = launch_s2run<128, 64, 2, 4>(g, s); break;      // 75 KB: two blocks per CU
            }
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 80 && variant <= 85) {
        // "long run" kernels on the padded raster (no padding selects in the MFMA phase): see k_conv3_run<..., PADO>
        hipError_t ev = hipErrorInvalidValue;
        if (ksize == 3 && stride == 1) {
            RunArgs g{};
            g.in = a.in; g.ldi = ldi; g.w = a.w; g.bias = bias; g.out = a.out; g.ldo = ldo; g.res = a.res; g.ldr = ldr;
            g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CoutPad = rva_ceil_div(Cout, 64) * 64; g.act = act; g.M = a.M;
            switch (variant) {
            case 80: ev = launch_run<256, 64, 4, 2, false, true>(g, s); break;
            case 81: ev = launch_run<256, 128, 4, 2, false, true>(g, s); break;
            case 82: ev = launch_run<224, 128, 2, 4, false, true>(g, s); break;
            case 83: ev = launch_run<320, 64, 4, 2, false, true>(g, s); break;
            case 84: ev = launch_run<192, 128, 4, 2, false, true>(g, s); break;
            default: ev = launch_run<128, 64, 2, 4, false, true>(g, s); break;
            }
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
#ifdef RVA_EXPERIMENTS
    if (variant >= 96) {
        hipError_t ev = hipErrorInvalidValue;
        if (ksize == 3 && stride == 1) {
            RunArgs g{};
            g.in = a.in; g.ldi = ldi; g.w = a.w; g.bias = bias; g.out = a.out; g.ldo = ldo; g.res = a.res; g.ldr = ldr;
            g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CoutPad = rva_ceil_div(Cout, 64) * 64; g.act = act; g.M = a.M;
            const char *nosel = getenv("RVA_NOSEL");      // the select-free kernels give wrong border pixels: opt-in per process
            if (variant == 96) { if (nosel && nosel[0] == '1') ev = launch_run<256, 128, 4, 2, true>(g, s); }       // run<256,128> without the padding selects
            else if (variant == 97) { if (nosel && nosel[0] == '1') ev = launch_run<256, 64, 4, 2, true>(g, s); }   // run<256,64> without the padding selects
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, RVA_ERR_ARG, "experiment variant %d not applicable here", variant);
    }
#endif
    if (variant >= 74) {
        // LDS-DMA gather kernel with several 64-channel K-steps per barrier (1x1, 3x3 of either stride; Cin % 64 == 0): the small-M layers
        a.CoutPad = cpad;
        if (ksize == 1 && stride != 1) return rva_fail(ctx, RVA_ERR_ARG, "conv variant %d not applicable here", variant);
        hipError_t ev;
        switch (variant) {
        case 74: ev = launch_gbig<128, 128, 2, 4, 2, 64, 2>(a, ksize, s); break;   // 128 KB
        case 75: ev = launch_gbig<128, 64, 2, 4, 2, 64, 2>(a, ksize, s); break;    // 96 KB
        case 76: ev = launch_gbig<128, 64, 2, 4, 2, 64, 3>(a, ksize, s); break;    // 144 KB
        case 77: ev = launch_gbig<64, 64, 2, 4, 2, 64, 4>(a, ksize, s); break;     // 128 KB
        case 78: ev = launch_gbig<64, 128, 2, 4, 2, 64, 3>(a, ksize, s); break;    // 144 KB
        default: ev = launch_gbig<64, 64, 2, 4, 2, 64, 2>(a, ksize, s); break;     // 64 KB: two blocks per CU
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 67) {
        // "whole chunk per barrier" kernels (3x3 stride 1, Cin % 32 == 0): the small-M layers, see k_conv3_chunk
        hipError_t ev = hipErrorInvalidValue;
        if (ksize == 3 && stride == 1) {
            RunArgs g{};
            g.in = a.in; g.ldi = ldi; g.w = a.w; g.bias = bias; g.out = a.out; g.ldo = ldo; g.res = a.res; g.ldr = ldr;
            g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CoutPad = rva_ceil_div(Cout, 64) * 64; g.act = act; g.M = a.M;
            switch (variant) {
            case 67: ev = launch_chunk<128, 64, 4, 2>(g, s); break;
            case 68: ev = launch_chunk<64, 64, 2, 4>(g, s); break;
            case 69: ev = launch_chunk<256, 64, 4, 2>(g, s); break;
            case 70: ev = launch_chunk<64, 96, 4, 2>(g, s); break;
            case 71: ev = launch_chunk<128, 96, 4, 2>(g, s); break;
            case 72: ev = launch_chunk<192, 64, 4, 2>(g, s); break;
            default: ev = launch_chunk<256, 96, 4, 2>(g, s); break;
            }
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 66) {
        // patch kernel with two wave sets half a tile period apart (3x3 stride 1, Cin = 64, Cout <= 64)
        hipError_t ev = hipErrorInvalidValue;
        S2Args g{a.in, ldi, a.w, bias, a.out, ldo, a.res, ldr, batch, H, W, a.Ho, a.Wo, Cout, cpad, act, 0, 0, 0};
        // sixteen waves: two sets of eight, one output row per wave -- two MFMA waves per SIMD in every phase.  (The 2 x 4 waves x
        // two rows form, one 256-register MFMA wave per SIMD, measured 27.0 us against 25.6 us at 80 x 80 and was dropped.)
        if (ksize == 3 && stride == 1 && Cin == 64 && Cout <= 64) ev = launch_patch2<64, 64, 8, 1>(g, num_cus, s);
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 64) {
        // LDS-DMA gather kernel with 256-channel output tiles: 43-64 MACs per staged byte against 32 of the 128 x 128 tile.  The CU's
        // vector-memory path moves 64 B/clk, its MFMAs 4096 MAC/clk: below 64 MAC/B the staging, not the matrix pipe, caps a
        // 1x1 convolution (a plain GEMM, no tap reuse).  One block per CU.
        a.CoutPad = cpad;
        if (ksize == 1 && stride != 1) return rva_fail(ctx, RVA_ERR_ARG, "conv variant %d not applicable here", variant);
        hipError_t ev;
        // (measured and dropped: <128,256> tiles, three- and four-slot rings with 32-channel steps -- more bytes in flight per CU
        //  did not help the memory-latency-bound 1x1 layers, profiles/r02_conv_tuning.txt)
        if (variant == 64) ev = launch_gbig<256, 256, 4, 2, 2>(a, ksize, s);     // 128 KB ring, wave tile 64 x 128
        else ev = launch_gbig<256, 256, 2, 4, 2>(a, ksize, s);                   // 128 KB ring, wave tile 128 x 64
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 52 && variant <= 60) {
        // "long run" LDS-DMA kernels (3x3 stride 1, Cin % 32 == 0): a chunk's activation run staged once for all three dy
        hipError_t ev = hipErrorInvalidValue;
        if (ksize == 3 && stride == 1) {
            RunArgs g{};
            g.in = a.in; g.ldi = ldi; g.w = a.w; g.bias = bias; g.out = a.out; g.ldo = ldo; g.res = a.res; g.ldr = ldr;
            g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CoutPad = rva_ceil_div(Cout, 64) * 64; g.act = act; g.M = a.M;
            switch (variant) {
            case 52: ev = launch_run<256, 64, 4, 2>(g, s); break;
            case 53: ev = launch_run<128, 64, 2, 4>(g, s); break;
            case 54: ev = launch_run<384, 64, 8, 1>(g, s); break;
            case 55: ev = launch_run<192, 128, 4, 2>(g, s); break;
            case 56: ev = launch_run<256, 128, 4, 2>(g, s); break;
            case 57: ev = launch_run<128, 128, 2, 4>(g, s); break;
            case 58: ev = launch_run<224, 128, 2, 4>(g, s); break;    // tile heights that fit whole rounds of the 256 CUs better
            case 59: ev = launch_run<160, 128, 2, 4>(g, s); break;
            default: ev = launch_run<320, 64, 4, 2>(g, s); break;
            }
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 43) {
        // patch kernels (weights resident in LDS): Cin = 32: 43 = 3x3 stride 2, Cout <= 64 (the first downsampling
        // convolution); 44 / 45 = 3x3 stride 1, Cout <= 32, tiles of 4 / 8 rows (the 32 -> 32 bottleneck convolutions)
        hipError_t ev = hipErrorInvalidValue;
        S2Args g{a.in, ldi, a.w, bias, a.out, ldo, a.res, ldr, batch, H, W, a.Ho, a.Wo, Cout, cpad, act, 0, 0, 0};
        if (ksize == 3 && Cin == 32) {
            if (variant == 43 && stride == 2 && Cout <= 64) ev = launch_patch<2, 32, 64, 4, 1>(g, num_cus, s);
            else if (variant == 44 && stride == 1 && Cout <= 32) ev = launch_patch<1, 32, 32, 4, 1>(g, num_cus, s);
            else if (variant == 45 && stride == 1 && Cout <= 32) ev = launch_patch<1, 32, 32, 8, 1>(g, num_cus, s);
        } else if (ksize == 3 && Cin == 64 && stride == 1 && Cout <= 64) {
            // 64 -> 64: 72 KB of weights resident, one block per CU
            if (variant == 46) ev = launch_patch<1, 64, 64, 4, 2>(g, num_cus, s);         // 124 KB, two patch buffers
            else if (variant == 47) ev = launch_patch<1, 64, 64, 8, 1>(g, num_cus, s);    // 116 KB, eight waves, one buffer
            else if (variant == 48) ev = launch_patch<1, 64, 64, 4, 1>(g, num_cus, s);    // 98 KB
            else if (variant == 49) ev = launch_patch<1, 64, 64, 8, 2>(g, num_cus, s);    // 158 KB: eight waves, two patch buffers
            else if (variant == 50) ev = launch_patch<1, 64, 64, 4, 2, 2>(g, num_cus, s); // 158 KB: four waves x two rows, two buffers
            else if (variant == 51) ev = launch_patch<1, 64, 64, 8, 1, 2>(g, num_cus, s); // 149 KB: 16-row tile, eight waves x two rows
            // output channels in two groups of 32: 36 KB of weights per block, two blocks per CU
            else if (variant == 61) ev = launch_patch<1, 64, 32, 8, 1, 1>(g, num_cus, s); // 79 KB: 8-row tile
            else if (variant == 62) ev = launch_patch<1, 64, 32, 4, 1, 1>(g, num_cus, s); // 62 KB: 4-row tile
            else if (variant == 63) ev = launch_patch<1, 64, 32, 4, 1, 2>(g, num_cus, s); // 79 KB: 8-row tile, four waves x two rows
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 33) {
        // large-tile LDS-DMA gather kernel, 64-channel K-steps (1x1, and 3x3 of either stride; Cin % 64 == 0)
        a.CoutPad = cpad;
        if (ksize == 1 && stride != 1) return rva_fail(ctx, RVA_ERR_ARG, "conv variant %d not applicable here", variant);
        hipError_t ev;
        switch (variant) {
        case 33: ev = launch_gbig<256, 128, 4, 2, 3>(a, ksize, s); break;    // 144 KB, one block per CU
        case 34: ev = launch_gbig<128, 128, 2, 4, 3>(a, ksize, s); break;    // 96 KB
        case 35: ev = launch_gbig<256, 64, 4, 2, 3>(a, ksize, s); break;     // 120 KB
        case 36: ev = launch_gbig<128, 64, 2, 4, 3>(a, ksize, s); break;     // 72 KB: two blocks per CU
        case 37: ev = launch_gbig<128, 128, 2, 4, 2>(a, ksize, s); break;    // 64 KB: two blocks per CU
        case 38: ev = launch_gbig<256, 64, 4, 2, 2>(a, ksize, s); break;     // 80 KB: two blocks per CU
        case 39: ev = launch_gbig<192, 128, 4, 2, 2>(a, ksize, s); break;    // 80 KB: two blocks per CU
        // 32-channel K-steps (64-byte rows, swz32): the layers with Cin = 32 / 96
        case 40: ev = launch_gbig<256, 64, 4, 2, 3, 32>(a, ksize, s); break; // 60 KB: two blocks per CU
        case 41: ev = launch_gbig<128, 64, 2, 4, 3, 32>(a, ksize, s); break; // 36 KB: four blocks per CU
        default: ev = launch_gbig<256, 64, 4, 2, 2, 32>(a, ksize, s); break; // 40 KB: three blocks per CU
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant >= 21) {
        // large-tile LDS-DMA kernel (3x3 stride 1, Cin % 32 == 0)
        hipError_t ev = hipErrorInvalidValue;
        if (ksize == 3 && stride == 1) {
            BigArgs g{};
            g.in = a.in; g.ldi = ldi; g.w = a.w; g.bias = bias; g.out = a.out; g.ldo = ldo; g.res = a.res; g.ldr = ldr;
            g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.CoutPad = rva_ceil_div(Cout, 64) * 64; g.act = act; g.M = a.M;
            switch (variant) {
            case 21: ev = launch_big<256, 128, 4, 2, 3>(g, s); break;
            case 22: ev = launch_big<128, 128, 2, 4, 3>(g, s); break;
            case 23: ev = launch_big<256, 64, 4, 2, 3>(g, s); break;
            case 24: ev = launch_big<128, 64, 2, 4, 3>(g, s); break;
            case 25: ev = launch_big<192, 128, 4, 2, 2>(g, s); break;     // 74 KB: two blocks per CU
            case 26: ev = launch_big<128, 128, 2, 4, 2>(g, s); break;     // 66 KB: two blocks per CU
            case 27: ev = launch_big<256, 64, 4, 2, 2>(g, s); break;      // 58 KB: two blocks per CU
            case 28: ev = launch_big<128, 64, 2, 4, 2>(g, s); break;      // 42 KB: three blocks per CU
            case 29: ev = launch_big<224, 128, 2, 4, 2>(g, s); break;     // 78 KB: two blocks per CU
            case 30: ev = launch_big<160, 128, 2, 4, 2>(g, s); break;     // 70 KB
            case 31: ev = launch_big<384, 64, 8, 1, 2>(g, s); break;      // 74 KB
            default: ev = launch_big<320, 64, 4, 2, 2>(g, s); break;      // 66 KB
            }
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (variant) {
        const int v = (variant - 1) & 3;                      // 0:<64,64> 1:<64,32> 2:<128,64> 3:<128,32>
        const int vbn = v >= 2 ? 128 : 64, vwpx = (v & 1) ? 32 : 64;
        if (vbn == 128 && !bn128) return rva_fail(ctx, RVA_ERR_ARG, "variant needs Cout padded to 128");
        hipError_t ev = hipErrorInvalidValue;
        if (variant <= 4) {
            a.n_tiles = cpad / vbn;
            a.m_tiles = rva_ceil_div(a.M, 4 * vwpx);
#define RVA_V(BN_, WPX_) (ksize == 1 ? launch_conv<BN_, WPX_, 1>(a, s) : launch_conv<BN_, WPX_, 3>(a, s))
            ev = v == 0 ? RVA_V(64, 64) : v == 1 ? RVA_V(64, 32) : v == 2 ? RVA_V(128, 64) : RVA_V(128, 32);
#undef RVA_V
        } else if (variant >= 17) {
            if (ksize == 3 && stride == 1) {
                RowArgs rr{};
                rr.in = a.in; rr.ldi = ldi; rr.w = a.w; rr.bias = bias; rr.out = a.out; rr.ldo = ldo; rr.res = a.res; rr.ldr = ldr;
                rr.H = H; rr.W = W; rr.Cin = Cin; rr.CinPad = a.CinPad; rr.Cout = Cout; rr.act = act;
                rr.n_tiles = cpad / vbn;
                ev = v == 0 ? launch_row<64, 64, true>(rr, batch, s) : v == 1 ? launch_row<64, 32, true>(rr, batch, s)
                   : v == 2 ? launch_row<128, 64, true>(rr, batch, s) : launch_row<128, 32, true>(rr, batch, s);
            }
        } else if (variant >= 13) {
            a.n_tiles = cpad / vbn;
            a.m_tiles = rva_ceil_div(a.M, 4 * vwpx);
#define RVA_V64(BN_, WPX_) (ksize == 1 ? launch_conv<BN_, WPX_, 1, 64>(a, s) : launch_conv<BN_, WPX_, 3, 64>(a, s))
            ev = v == 0 ? RVA_V64(64, 64) : v == 1 ? RVA_V64(64, 32) : v == 2 ? RVA_V64(128, 64) : RVA_V64(128, 32);
#undef RVA_V64
        } else if (variant >= 9) {
            if (ksize == 3 && stride == 1) {
                RowArgs rr{};
                rr.in = a.in; rr.ldi = ldi; rr.w = a.w; rr.bias = bias; rr.out = a.out; rr.ldo = ldo; rr.res = a.res; rr.ldr = ldr;
                rr.H = H; rr.W = W; rr.Cin = Cin; rr.CinPad = a.CinPad; rr.Cout = Cout; rr.act = act;
                rr.n_tiles = cpad / vbn;
                ev = v == 0 ? launch_row<64, 64>(rr, batch, s) : v == 1 ? launch_row<64, 32>(rr, batch, s)
                   : v == 2 ? launch_row<128, 64>(rr, batch, s) : launch_row<128, 32>(rr, batch, s);
            }
        } else if (stride == 1) {
            ResArgs ra{};
            ra.in = a.in; ra.ldi = ldi; ra.w = a.w; ra.bias = bias; ra.out = a.out; ra.ldo = ldo; ra.res = a.res; ra.ldr = ldr;
            ra.H = H; ra.W = W; ra.Cin = Cin; ra.CinPad = a.CinPad; ra.Cout = Cout; ra.act = act; ra.M = a.M;
            ra.n_tiles = cpad / vbn;
            if (ksize == 3)
                ev = v == 0 ? launch_res<64, 64, 3, 32, 9>(ra, batch, num_cus, s) : v == 1 ? launch_res<64, 32, 3, 32, 11>(ra, batch, num_cus, s)
                   : v == 2 ? launch_res<128, 64, 3, 32, 9>(ra, batch, num_cus, s) : launch_res<128, 32, 3, 32, 11>(ra, batch, num_cus, s);
            else if (a.CinPad % 128 == 0)
                ev = v == 0 ? launch_res<64, 64, 1, 128, 16>(ra, batch, num_cus, s) : v == 1 ? launch_res<64, 32, 1, 128, 8>(ra, batch, num_cus, s)
                   : v == 2 ? launch_res<128, 64, 1, 128, 16>(ra, batch, num_cus, s) : hipErrorInvalidValue;
            else
                ev = v == 0 ? launch_res<64, 64, 1, 32, 4>(ra, batch, num_cus, s) : v == 1 ? launch_res<64, 32, 1, 32, 2>(ra, batch, num_cus, s)
                   : hipErrorInvalidValue;
        }
        if (ev == hipSuccess) return RVA_OK;
        (void)hipGetLastError();
        return rva_fail(ctx, ev == hipErrorInvalidValue ? RVA_ERR_ARG : RVA_ERR_HIP, "conv variant %d not applicable here", variant);
    }
    if (stride == 1) {
        ResArgs ra{};
        ra.in = a.in; ra.ldi = ldi; ra.w = a.w; ra.bias = bias; ra.out = a.out; ra.ldo = ldo; ra.res = a.res; ra.ldr = ldr;
        ra.H = H; ra.W = W; ra.Cin = Cin; ra.CinPad = a.CinPad; ra.Cout = Cout; ra.act = act; ra.M = a.M;
#ifdef RVA_ROW_STAMPS
        { const char *e = getenv("RVA_CONV_DBG"); ra.dbg = e ? atoi(e) : 0; }
#endif
        const bool bn128 = cpad % 128 == 0;
        hipError_t e2 = hipErrorInvalidValue;
        if (ksize == 3) {
            // tile = 256 px x 128 ch when that yields enough tiles for every CU, else smaller tiles
            const long t256 = (long)batch * rva_ceil_div(H * W, 256), t128 = (long)batch * rva_ceil_div(H * W, 128);
            if (bn128 && t256 * (cpad / 128) >= 2 * num_cus) { ra.n_tiles = cpad / 128; e2 = launch_res<128, 64, 3, 32, 9>(ra, batch, num_cus, s); }
            if (e2 == hipErrorInvalidValue && t256 * (cpad / 64) >= 2 * num_cus) { ra.n_tiles = cpad / 64; e2 = launch_res<64, 64, 3, 32, 9>(ra, batch, num_cus, s); }
            if (e2 == hipErrorInvalidValue && bn128 && t128 * (cpad / 128) >= 2 * num_cus) { ra.n_tiles = cpad / 128; e2 = launch_res<128, 32, 3, 32, 11>(ra, batch, num_cus, s); }
            if (e2 == hipErrorInvalidValue) { ra.n_tiles = cpad / 64; e2 = launch_res<64, 32, 3, 32, 11>(ra, batch, num_cus, s); }
        } else if (a.CinPad % 128 == 0) {
            const long t256 = rva_ceil_div(a.M, 256);
            if (bn128 && t256 * (cpad / 128) >= 2 * num_cus) { ra.n_tiles = cpad / 128; e2 = launch_res<128, 64, 1, 128, 16>(ra, batch, num_cus, s); }
            if (e2 == hipErrorInvalidValue && t256 * (cpad / 64) >= 2 * num_cus) { ra.n_tiles = cpad / 64; e2 = launch_res<64, 64, 1, 128, 16>(ra, batch, num_cus, s); }
            if (e2 == hipErrorInvalidValue) { ra.n_tiles = cpad / 64; e2 = launch_res<64, 32, 1, 128, 8>(ra, batch, num_cus, s); }
        } else {
            const long t256 = rva_ceil_div(a.M, 256);
            if (t256 * (cpad / 64) >= 2 * num_cus) { ra.n_tiles = cpad / 64; e2 = launch_res<64, 64, 1, 32, 4>(ra, batch, num_cus, s); }
            else { ra.n_tiles = cpad / 64; e2 = launch_res<64, 32, 1, 32, 2>(ra, batch, num_cus, s); }
        }
        if (e2 == hipSuccess) return RVA_OK;
        if (e2 != hipErrorInvalidValue) return rva_fail(ctx, RVA_ERR_HIP, "resident conv launch failed: %s", hipGetErrorString(e2));
        (void)hipGetLastError();   // geometry does not fit the resident kernel: fall through to the gather kernel
    }
    a.n_tiles = cpad / (bn128 ? 128 : 64);
    // small problems: 128-pixel tiles keep more CUs busy
    const bool small = (long)rva_ceil_div(a.M, 256) * a.n_tiles < 512;
    const int BM = small ? 128 : 256;
    a.m_tiles = rva_ceil_div(a.M, BM);
    hipError_t e;
#define RVA_CONV(BN_, WPX_)                                                  \
    (ksize == 1 ? launch_conv<BN_, WPX_, 1>(a, s) : launch_conv<BN_, WPX_, 3>(a, s))
    if (bn128) e = small ? RVA_CONV(128, 32) : RVA_CONV(128, 64);
    else e = small ? RVA_CONV(64, 32) : RVA_CONV(64, 64);
#undef RVA_CONV
    if (e != hipSuccess) return rva_fail(ctx, RVA_ERR_HIP, "conv launch failed: %s", hipGetErrorString(e));
    return RVA_OK;
}

int rva_conv1x1_upcat_f16(rva_ctx *ctx, const void *low, int ld_low, int c_low, const void *skip, int ld_skip, int c_skip,
                          const void *weights, const float *bias, void *out, int ldo, int batch, int H, int W, int Cout,
                          int act, int variant, rva_stream_t stream_)
{
    if (!ctx) return RVA_ERR_ARG;
    if (!low || !skip || !weights || !bias || !out || batch <= 0 || H <= 0 || W <= 0 || (H | W) & 1 || c_low <= 0 || c_skip <= 0 ||
        c_low % 64 || c_skip % 64 || Cout % 8 || ld_low % 8 || ld_skip % 8 || ldo % 8 ||
        ((uintptr_t)low | (uintptr_t)skip | (uintptr_t)out | (uintptr_t)weights) % 16 || H > 2000 || W > 2000)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_conv1x1_upcat_f16: unsupported shape/alignment (even H, W; channel counts %% 64; ld %% 8)");
    ConvArgs a{};
    a.in = (const __half *)low; a.ldi = ld_low; a.in2 = (const __half *)skip; a.ldi2 = ld_skip; a.c_split = c_low;
    a.w = (const __half *)weights; a.bias = bias; a.out = (__half *)out; a.ldo = ldo; a.res = nullptr; a.ldr = 0;
    a.H = H; a.W = W; a.Cin = c_low + c_skip; a.CinPad = a.Cin; a.Cout = Cout; a.stride = 1; a.act = act;
    a.Ho = H; a.Wo = W; a.M = batch * H * W; a.CoutPad = rva_ceil_div(Cout, 64) * 64;
    hipStream_t s = (hipStream_t)stream_;
    hipError_t ev;
    switch (variant) {
    case 33: ev = launch_gbig1<256, 128, 4, 2, 3, 1, 64, true>(a, s); break;
    case 34: ev = launch_gbig1<128, 128, 2, 4, 3, 1, 64, true>(a, s); break;
    case 35: ev = launch_gbig1<256, 64, 4, 2, 3, 1, 64, true>(a, s); break;
    case 36: ev = launch_gbig1<128, 64, 2, 4, 3, 1, 64, true>(a, s); break;
    case 0:
    case 37: ev = launch_gbig1<128, 128, 2, 4, 2, 1, 64, true>(a, s); break;
    case 38: ev = launch_gbig1<256, 64, 4, 2, 2, 1, 64, true>(a, s); break;
    case 39: ev = launch_gbig1<192, 128, 4, 2, 2, 1, 64, true>(a, s); break;
    default: return rva_fail(ctx, RVA_ERR_ARG, "rva_conv1x1_upcat_f16: variant %d not applicable (0 or 33..39)", variant);
    }
    if (ev != hipSuccess) return rva_fail(ctx, RVA_ERR_HIP, "rva_conv1x1_upcat_f16: launch failed: %s", hipGetErrorString(ev));
    return RVA_OK;
}

int rva_conv1x1_head_f16(rva_ctx *ctx, const void *in, int ldi, const void *weights, const float *bias, int batch, int H, int W,
                         int Cin, int Cout, int mode, void *out, int nc, int anchors_total, int anchor_offset, float stride_px,
                         int variant, rva_stream_t stream_)
{
    if (!ctx) return RVA_ERR_ARG;
    if (!in || !weights || !bias || !out || batch <= 0 || H <= 0 || W <= 0 || Cin % 64 || Cout % 8 || ldi % 8 || nc % 8 ||
        (mode != 1 && mode != 2) || (mode == 1 && Cout != 64) || (mode == 2 && Cout != nc) ||
        ((uintptr_t)in | (uintptr_t)weights) % 16 || H > 2000 || W > 2000)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_conv1x1_head_f16: unsupported shape (Cin %% 64; box branch Cout = 64; class branch Cout = nc)");
    ConvArgs a{};
    a.in = (const __half *)in; a.ldi = ldi; a.w = (const __half *)weights; a.bias = bias; a.out = nullptr; a.ldo = 0;
    a.H = H; a.W = W; a.Cin = Cin; a.CinPad = Cin; a.Cout = Cout; a.stride = 1; a.act = 0; a.Ho = H; a.Wo = W; a.M = batch * H * W;
    a.CoutPad = rva_ceil_div(Cout, 64) * 64;
    a.hout = (__half *)out; a.hmode = mode; a.hnc = nc; a.hA = anchors_total; a.ha0 = anchor_offset; a.hW = W; a.hHW = H * W;
    a.hstride = stride_px;
    hipStream_t s = (hipStream_t)stream_;
    hipError_t ev;
    switch (variant) {
    case 33: ev = launch_gbig1<256, 128, 4, 2, 3, 1, 64, false, true>(a, s); break;
    case 34: ev = launch_gbig1<128, 128, 2, 4, 3, 1, 64, false, true>(a, s); break;
    case 35: ev = launch_gbig1<256, 64, 4, 2, 3, 1, 64, false, true>(a, s); break;
    case 36: ev = launch_gbig1<128, 64, 2, 4, 3, 1, 64, false, true>(a, s); break;
    case 0:
    case 37: ev = launch_gbig1<128, 128, 2, 4, 2, 1, 64, false, true>(a, s); break;
    case 38: ev = launch_gbig1<256, 64, 4, 2, 2, 1, 64, false, true>(a, s); break;
    case 39: ev = launch_gbig1<192, 128, 4, 2, 2, 1, 64, false, true>(a, s); break;
    default: return rva_fail(ctx, RVA_ERR_ARG, "rva_conv1x1_head_f16: variant %d not applicable (0 or 33..39)", variant);
    }
    if (ev != hipSuccess) return rva_fail(ctx, RVA_ERR_HIP, "rva_conv1x1_head_f16: launch failed: %s", hipGetErrorString(ev));
    return RVA_OK;
}

int rva_conv_cout_pad(int Cout) { return rva_ceil_div(Cout, 64) * 64; }

int rva_conv_num_variants(void) { return RVA_CONV_VARIANTS_MAX; }

int rva_stem_conv_f16(rva_ctx *ctx, const void *in_planar, const void *weights, const float *bias, void *out, int ldo,
                      int batch, int H, int W, int Cout, rva_stream_t stream_)
{
    if (!ctx || !in_planar || !weights || !bias || !out || Cout % 8 || Cout > 64 || ldo % 8)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_stem_conv_f16: bad argument");
    StemArgs a{(const __half *)in_planar, nullptr, bias, (__half *)out, batch, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, Cout, ldo};
#ifdef RVA_ROW_STAMPS
    { const char *e = getenv("RVA_STEM_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    const int total_tiles = rva_ceil_div(a.Wo, 32) * rva_ceil_div(a.Ho, 8) * batch;
    int grid = 256 * 4;                                     // persistent: ~4 blocks per CU (LDS-limited)
    if (grid > total_tiles) grid = total_tiles;
    if (W % 8 || ((uintptr_t)in_planar & 15)) return rva_fail(ctx, RVA_ERR_ARG, "rva_stem_conv_f16: W %% 8 == 0 and a 16-byte aligned input are required");
    k_stem<<<grid, 256, (size_t)(3 * 17 * 88 * 2) + (size_t)(256 * (Cout + 8 > LDSROW ? Cout + 8 : LDSROW) + 64 * LDSROW) * 2, (hipStream_t)stream_>>>(a, (const __half *)weights, bias);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_c2f_pair32_f16(rva_ctx *ctx, const void *in, int ldi, const void *w1, const float *b1, const void *w2, const float *b2, void *out,
                       int ldo, int batch, int H, int W, rva_stream_t stream_)
{
    if (!ctx || !in || !w1 || !b1 || !w2 || !b2 || !out || batch <= 0 || H <= 0 || W <= 0 || ldi % 8 || ldo % 8 || ldi < 32 || ldo < 32)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_c2f_pair32_f16: bad argument");
    if ((((size_t)in) | ((size_t)out) | ((size_t)w1) | ((size_t)w2)) & 15) return rva_fail(ctx, RVA_ERR_ARG, "rva_c2f_pair32_f16: 16-byte aligned tensors");
    if ((size_t)batch * H * W >= (1ull << 31)) return rva_fail(ctx, RVA_ERR_ARG, "rva_c2f_pair32_f16: tensor too large");
    if (!ctx->num_cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
        if (ctx->num_cus <= 0) ctx->num_cus = 256;
    }
    PairArgs a{(const __half *)in, ldi, (const __half *)w1, b1, (const __half *)w2, b2, (__half *)out, ldo, batch, H, W,
               rva_ceil_div(W, PR_TW), rva_ceil_div(H, PR_TH), 0};
    a.total = a.tiles_x * a.tiles_y * batch;
    RVA_HIP(ctx, rva_func_smem((const void *)k_c2f_pair32, PR_SMEM));
    int grid = 2 * ctx->num_cus;
    if (grid > a.total) grid = a.total;
    k_c2f_pair32<<<grid, 256, PR_SMEM, (hipStream_t)stream_>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_stem2_f16(rva_ctx *ctx, const void *in_planar, const void *w1, const float *b1, const void *w2, const float *b2,
                  void *out, int ldo, int batch, int H, int W, rva_stream_t stream_)
{
    if (!ctx || !in_planar || !w1 || !b1 || !w2 || !b2 || !out || ldo % 8 || ldo < 64 || batch <= 0 || H < 4 || W < 4)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_stem2_f16: bad argument");
    if (W % 8 || ((uintptr_t)in_planar & 15)) return rva_fail(ctx, RVA_ERR_ARG, "rva_stem2_f16: W %% 8 == 0 and a 16-byte aligned input are required");
    Stem2Args a{(const __half *)in_planar, (const __half *)w1, b1, (const __half *)w2, b2, (__half *)out, ldo, batch, H, W};
    a.H1 = (H - 1) / 2 + 1; a.W1 = (W - 1) / 2 + 1;
    a.Ho = (a.H1 - 1) / 2 + 1; a.Wo = (a.W1 - 1) / 2 + 1;
    a.tiles_x = rva_ceil_div(a.Wo, S2_TW); a.tiles_y = rva_ceil_div(a.Ho, S2_TH);
    a.total = a.tiles_x * a.tiles_y * batch;
    constexpr size_t smem = S2_SMEM;
    RVA_HIP(ctx, rva_func_smem((const void *)k_stem2, smem));
    if (!ctx->num_cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
        if (ctx->num_cus <= 0) ctx->num_cus = 256;
    }
    int grid = ctx->num_cus;                               // persistent: one block per CU (159 KB of LDS each)
    if (grid > a.total) grid = a.total;
    k_stem2<<<grid, 1024, smem, (hipStream_t)stream_>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_maxpool5_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, void *out, int ldo, int batch, int H, int W, int C,
                          rva_stream_t stream_)
{
    if (!ctx || !in || !out || C % 8 || ldi % 8 || ldo % 8) return rva_fail(ctx, RVA_ERR_ARG, "rva_maxpool5_nhwc_f16: bad argument");
    const long n = (long)batch * H * W * (C / 8);
    k_maxpool5<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>((const __half *)in, ldi, (__half *)out, ldo, batch, H, W, C);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_sppf_pool3_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, void *out1, void *out2, void *out3, int ldo, int batch,
                            int H, int W, int C, rva_stream_t stream_)
{
    if (!ctx || !in || !out1 || !out2 || !out3 || C % 8 || ldi % 8 || ldo % 8 || batch <= 0 || H <= 0 || W <= 0)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_sppf_pool3_nhwc_f16: bad argument");
    const size_t smem = (size_t)H * W * 16 * 4;
    if (smem > 150 * 1024) return rva_fail(ctx, RVA_ERR_ARG, "rva_sppf_pool3_nhwc_f16: H*W too large for one LDS tile (use rva_maxpool5_nhwc_f16 x3)");
    // whole 64-byte runs per wave where that still fills the chip (32 frames of YOLOv8s: 256 blocks of 1024 threads)
    if (C % 32 == 0 && (long)batch * (C / 32) >= 192 && (size_t)H * W * 4 <= 4096 && (size_t)H * W * 4 * 16 * 3 <= 150 * 1024 && !getenv("RVA_SPPF_G1")) {
        RVA_HIP(ctx, rva_func_smem((const void *)k_sppf_pool3g<4>, 150 * 1024));
        k_sppf_pool3g<4><<<batch * (C / 32), 1024, (size_t)H * W * 4 * 16 * 3, (hipStream_t)stream_>>>((const __half *)in, ldi, (__half *)out1,
                                                                                              (__half *)out2, (__half *)out3, ldo, H, W, C);
        RVA_HIP(ctx, hipGetLastError());
        return RVA_OK;
    }
    RVA_HIP(ctx, rva_func_smem((const void *)k_sppf_pool3, 150 * 1024));
    k_sppf_pool3<<<batch * (C / 8), 256, smem, (hipStream_t)stream_>>>((const __half *)in, ldi, (__half *)out1, (__half *)out2,
                                                                       (__half *)out3, ldo, H, W, C);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_upsample2x_nhwc_f16(rva_ctx *ctx, const void *in, int ldi, void *out, int ldo, int batch, int H, int W, int C,
                            rva_stream_t stream_)
{
    if (!ctx || !in || !out || C % 8 || ldi % 8 || ldo % 8) return rva_fail(ctx, RVA_ERR_ARG, "rva_upsample2x_nhwc_f16: bad argument");
    const long n = (long)batch * H * 2 * W * 2 * (C / 8);
    k_upsample2<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>((const __half *)in, ldi, (__half *)out, ldo, batch, H, W, C);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_yolo_head3_f16(rva_ctx *ctx, const void *const *box_logits, const int32_t *ldb, const void *const *cls_logits,
                       const int32_t *ldc, void *out, int batch, const int32_t *h, const int32_t *w, int nc, int anchors_total,
                       const float *strides, rva_stream_t stream_)
{
    if (!ctx || !box_logits || !cls_logits || !ldb || !ldc || !h || !w || !strides || !out || nc % 8)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_yolo_head3_f16: bad argument");
    Head3Args a{};
    int a0 = 0, total = 0;
    for (int l = 0; l < 3; ++l) {
        if (!box_logits[l] || !cls_logits[l] || ldb[l] % 8 || ldc[l] % 8 || h[l] <= 0 || w[l] <= 0)
            return rva_fail(ctx, RVA_ERR_ARG, "rva_yolo_head3_f16: bad level %d", l);
        a.lv[l] = HeadArgs{(const __half *)box_logits[l], ldb[l], (const __half *)cls_logits[l], ldc[l], (__half *)out, batch,
                           h[l], w[l], nc, anchors_total, a0, strides[l]};
        a.blocks[l] = rva_ceil_div(h[l] * w[l], 256);
        total += a.blocks[l];
        a0 += h[l] * w[l];
    }
    if (a0 != anchors_total) return rva_fail(ctx, RVA_ERR_ARG, "rva_yolo_head3_f16: anchors_total != sum of h*w");
    k_head3<<<dim3(total, batch), 256, 0, (hipStream_t)stream_>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_yolo_head_f16(rva_ctx *ctx, const void *box_logits, int ldb, const void *cls_logits, int ldc, void *out, int batch,
                      int h, int w, int nc, int anchors_total, int anchor_offset, float stride, rva_stream_t stream_)
{
    if (!ctx || !box_logits || !cls_logits || !out || nc % 8 || ldb % 8 || ldc % 8)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_yolo_head_f16: bad argument");
    HeadArgs a{(const __half *)box_logits, ldb, (const __half *)cls_logits, ldc, (__half *)out, batch, h, w, nc,
               anchors_total, anchor_offset, stride};
    dim3 g(rva_ceil_div(h * w, 256), batch);
    k_head<<<g, 256, 0, (hipStream_t)stream_>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

}  // extern "C"

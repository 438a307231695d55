// K7: baseline JPEG encoder on the device -- the "encode" half of SURVEY.md 8f-3.
//
// Replaces cv2.imencode('.jpg', frame, [IMWRITE_JPEG_QUALITY, q, ...]) of KafkaSink._render_frame
// (sinks/kafka_sink.py:260-284) and of StreamWorker._maybe_save_snapshot (pipeline.py:264-290) for the preview K6 renders:
// until round 3 the 6.2 MB BGR preview crossed PCIe to a host encoder; now the uint8 BGR image in HBM goes in and a JFIF byte
// stream (typically 100-300 KB) comes out.  The arithmetic is libjpeg's, step by step (oracle/jpeg_oracle.py is the CPU
// restatement, pinned by Pillow's libjpeg-turbo: its decode of this stream == its decode of its own encoding of the image):
//   k7_transform  one thread per 8x8 block, blocks in MCU order (4:2:0: Y00 Y01 Y10 Y11 Cb Cr per 16x16 MCU): RGB -> YCbCr in
//                 16-bit fixed point (jccolor.c), edge replication (last column before the 2x2 chroma downsampling with its
//                 alternating 1,2 bias, last DOWNSAMPLED row after it: jcprepct.c / jcsample.c), level shift, forward DCT
//                 "islow" (jfdctint.c), quantisation with the quality-scaled Annex-K tables (jcparam.c / jcdctmgr.c), zigzag;
//                 luma blocks wholly outside the image are dummy blocks (AC 0, DC copied: jccoefct.c).
//   k7_entropy    one wave per restart interval (= one MCU row; DRI in the header): 64 blocks per pass, a lane Huffman-codes its
//                 block (Annex-K tables, DC prediction inside the interval) into a private LDS bit buffer, a wave scan of the
//                 bit counts places the blocks, lanes OR their bits into the pass's stream (LDS atomics), the complete bytes
//                 are byte-stuffed (0xFF -> 0xFF 0x00; positions by ballot + popcount) into the interval's staging area, the
//                 odd bits carry into the next pass; the interval ends padded with ones + RSTn (EOI after the last).
//   k7_gather     header + intervals -> one contiguous stream, its length to `out_size`.
// Every stage is byte / integer work: bit-exact against the oracle (tests/test_gpu_api.py).  Not a hot-path kernel: previews are
// rate-limited to 10 per second per stream (kafka_sink.py:49) -- 1080p encodes in well under a millisecond, latency-bound.
#include <cstring>

#include "rva_internal.h"

namespace {

constexpr uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                                    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                    47, 55, 62, 63};
const uint8_t kStdLumaQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51,
                               87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101,
                               72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kStdChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99,
                                 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                 99, 99, 99, 99, 99, 99, 99, 99};
// Annex K.3: codes per length 1..16, symbols
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffTables {          // code << 8 | size per symbol: [0] DC luma, [1] DC chroma (12 entries used), [2] AC luma, [3] AC chroma
    uint32_t t[4][256];
};

struct K7Args {
    const uint8_t *bgr; int pitch, w, h;
    int mw, mh;                         // MCUs per row / rows
    uint16_t qdiv[2][64];               // quantisation divisors (table << 3), natural order: [0] luma, [1] chroma
    int16_t *coef;                      // [mh * mw * 6][64] zigzag order
    const HuffTables *huff;
    uint8_t *stage; int stage_stride;   // per restart interval: stage_stride bytes
    int32_t *isize;                     // [mh] bytes of each interval (marker included)
    int32_t *flags;                     // bit 0: an interval did not fit its staging area / the stream did not fit `out`
    uint8_t *out; int out_cap; int32_t *out_size;
    uint8_t header[640]; int header_len;
};

__device__ __forceinline__ void ycc(const uint8_t *p, int &y, int &cb, int &cr)
{
    const int b = p[0], g = p[1], r = p[2];                        // jccolor.c, SCALEBITS 16
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 1-D pass over eight values (FIRST: rows, output up-scaled by 2^PASS1_BITS; second: columns)
template <bool FIRST>
__device__ __forceinline__ void fdct8(int *d0, int stride)
{
    constexpr int C = 13, P = 2;
    int *d = d0;
    const int v0 = d[0], v1 = d[stride], v2 = d[2 * stride], v3 = d[3 * stride], v4 = d[4 * stride], v5 = d[5 * stride], v6 = d[6 * stride], v7 = d[7 * stride];
    const int t0 = v0 + v7, t7 = v0 - v7, t1 = v1 + v6, t6 = v1 - v6, t2 = v2 + v5, t5 = v2 - v5, t3 = v3 + v4, t4 = v3 - v4;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int SH = FIRST ? C - P : C + P;
    d[0] = FIRST ? (t10 + t11) << P : descale(t10 + t11, P);
    d[4 * stride] = FIRST ? (t10 - t11) << P : descale(t10 - t11, P);
    int z1 = (t12 + t13) * 4433;
    d[2 * stride] = descale(z1 + t13 * 6270, SH);
    d[6 * stride] = descale(z1 - t12 * 15137, SH);
    z1 = t4 + t7; int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
    z1 = -z1 * 7373; z2 = -z2 * 20995; z3 = -z3 * 16069 + z5; z4 = -z4 * 3196 + z5;
    d[7 * stride] = descale(a4 + z1 + z3, SH);
    d[5 * stride] = descale(a5 + z2 + z4, SH);
    d[3 * stride] = descale(a6 + z2 + z3, SH);
    d[stride] = descale(a7 + z1 + z4, SH);
}

__device__ __forceinline__ int quant(int x, int q)                 // jcdctmgr.c: symmetric round-half-up of |x| / q
{
    const int a = x < 0 ? -x : x;
    const int v = (a + (q >> 1)) / q;
    return x < 0 ? -v : v;
}

// sample of component `comp` (0 Y, 1 Cb, 2 Cr) at its own resolution, with libjpeg's edge rules
__device__ __forceinline__ int sample(const K7Args &a, int comp, int yy, int xx)
{
    int y, cb, cr;
    if (comp == 0) {
        ycc(a.bgr + (size_t)min(yy, a.h - 1) * a.pitch + 3 * min(xx, a.w - 1), y, cb, cr);
        return y;
    }
    const int he = a.h + (a.h & 1);
    const int cy = min(yy, he / 2 - 1);                            // rows past the image: the last DOWNSAMPLED row again
    int s = (xx & 1) ? 2 : 1;                                      // h2v2_downsample: bias 1, 2, 1, 2, ...
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            ycc(a.bgr + (size_t)min(2 * cy + dy, a.h - 1) * a.pitch + 3 * min(2 * xx + dx, a.w - 1), y, cb, cr);
            s += comp == 1 ? cb : cr;
        }
    return s >> 2;
}

__global__ void __launch_bounds__(64) k7_transform(K7Args a)
{
    const int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.mw * a.mh * 6) return;
    const int mcu = blk / 6, k = blk - mcu * 6, my = mcu / a.mw, mx = mcu - my * a.mw;
    const int comp = k < 4 ? 0 : k - 3;
    int by = comp == 0 ? 2 * my + (k >> 1) : my, bx = comp == 0 ? 2 * mx + (k & 1) : mx;
    int16_t *o = a.coef + (size_t)blk * 64;
    const uint16_t *qd = a.qdiv[comp ? 1 : 0];
    if (comp == 0) {
        const int nby = (a.h + 7) >> 3, nbx = (a.w + 7) >> 3;
        const bool dr = bx >= nbx, db = by >= nby;                 // dummy block at the right edge / in a dummy row at the bottom
        if (dr || db) {
            // jccoefct.c: AC = 0, DC = the DC of the block before it in the MCU buffer -- the block to the left (right edge), the
            // LAST block of the row above for both blocks of a dummy row (itself a right-edge dummy when the image ends there too).
            // The DC of a real block is the sum of its 64 samples - 128 (both passes of the DCT are exact for the DC term).
            if (db) { by = 2 * my; bx = 2 * mx + 1; }
            if (bx >= nbx) bx = 2 * mx;
            int sum = 0;
            for (int i = 0; i < 64; ++i) sum += sample(a, 0, by * 8 + (i >> 3), bx * 8 + (i & 7)) - 128;
            o[0] = (int16_t)quant(sum, qd[0]);
            for (int i = 1; i < 64; ++i) o[i] = 0;
            return;
        }
    }
    int d[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = sample(a, comp, by * 8 + (i >> 3), bx * 8 + (i & 7)) - 128;
#pragma unroll
    for (int r = 0; r < 8; ++r) fdct8<true>(d + 8 * r, 1);
#pragma unroll
    for (int c = 0; c < 8; ++c) fdct8<false>(d + c, 8);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int n = kZigzag[i];
        o[i] = (int16_t)quant(d[n], qd[n]);
    }
}

// ---- entropy coding ---------------------------------------------------------------------------------
constexpr int K7_PRIV = 56;             // dwords of a lane's private bit buffer: 64 coefficients x at most 27 bits = 1728 bits = 54 dwords
constexpr int K7_GROUP = 64 * 54 + 4;   // dwords of a pass's stream (carry + 64 blocks)

struct BitW { unsigned long long acc; int n, w; uint32_t *out; };
__device__ __forceinline__ void put(BitW &b, uint32_t code, int size)
{
    b.acc = (b.acc << size) | (code & ((1u << size) - 1u));
    b.n += size;
    if (b.n >= 32) { b.out[b.w++] = (uint32_t)(b.acc >> (b.n - 32)); b.n -= 32; }
}
__device__ __forceinline__ void put_sym(BitW &b, uint32_t e) { put(b, e >> 8, (int)(e & 0xff)); }
__device__ __forceinline__ int nbits(int v) { const int t = v < 0 ? -v : v; return t ? 32 - __clz(t) : 0; }

__global__ void __launch_bounds__(64) k7_entropy(K7Args a)
{
    __shared__ uint32_t priv[64 * K7_PRIV];
    __shared__ uint32_t grp[K7_GROUP];
    const int my = blockIdx.x, lane = threadIdx.x;
    const int nblk = a.mw * 6;
    uint8_t *dst = a.stage + (size_t)my * a.stage_stride;
    const int cap = a.stage_stride - 8;
    int out_pos = 0, carry_bits = 0;
    bool over = false;
    for (int i = lane; i < K7_GROUP; i += 64) grp[i] = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 64) {
        const int bi = base + lane;
        const bool live = bi < nblk;
        int tb = 0;                                                // bits of this lane's block
        uint32_t *mine = priv + lane * K7_PRIV;
        if (live) {
            const int mx = bi / 6, k = bi - mx * 6;
            const int16_t *zz = a.coef + ((size_t)my * nblk + bi) * 64;
            int pred = 0;                                          // DC prediction restarts with the interval
            if (k >= 1 && k <= 3) pred = zz[-64];
            else if (mx > 0) pred = k == 0 ? zz[-3 * 64] : zz[-6 * 64];
            const uint32_t *dc = a.huff->t[k < 4 ? 0 : 1], *ac = a.huff->t[k < 4 ? 2 : 3];
            BitW bw{0ull, 0, 0, mine};
            const int diff = (int)zz[0] - pred;
            int nb = nbits(diff);
            put_sym(bw, dc[nb]);
            if (nb) put(bw, (uint32_t)(diff >= 0 ? diff : diff - 1), nb);
            int run = 0;
            for (int i = 1; i < 64; ++i) {
                const int v = zz[i];
                if (v == 0) { ++run; continue; }
                while (run > 15) { put_sym(bw, ac[0xF0]); run -= 16; }
                nb = nbits(v);
                put_sym(bw, ac[(run << 4) | nb]);
                put(bw, (uint32_t)(v >= 0 ? v : v - 1), nb);
                run = 0;
            }
            if (run) put_sym(bw, ac[0]);
            tb = bw.w * 32 + bw.n;
            if (bw.n) mine[bw.w] = (uint32_t)(bw.acc << (32 - bw.n));      // left-aligned tail
        }
        // exclusive scan of the bit counts over the wave
        int incl = tb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int total = __shfl(incl, 63);
        const int off = carry_bits + incl - tb;
        if (live) {
            const int nw = (tb + 31) >> 5, s = off & 31, w0 = off >> 5;
            for (int j = 0; j < nw; ++j) {
                const uint32_t v = mine[j];
                atomicOr(&grp[w0 + j], v >> s);
                if (s) atomicOr(&grp[w0 + j + 1], v << (32 - s));
            }
        }
        __syncthreads();
        const int bits = carry_bits + total, nbytes = bits >> 3;
        for (int b0 = 0; b0 < nbytes; b0 += 64) {                  // byte stuffing, 64 bytes per step
            const int i = b0 + lane;
            const bool valid = i < nbytes;
            const uint32_t byte = valid ? (grp[i >> 2] >> (24 - 8 * (i & 3))) & 0xffu : 0u;
            const bool ff = valid && byte == 0xffu;
            const unsigned long long m = __ballot(ff);
            const int pos = out_pos + (i - b0) + __popcll(m & ((1ull << lane) - 1ull));
            const int step = min(64, nbytes - b0) + __popcll(m);
            if (out_pos + step > cap) over = true;
            if (valid && !over) {
                dst[pos] = (uint8_t)byte;
                if (ff) dst[pos + 1] = 0;
            }
            out_pos += step;
        }
        // the odd bits carry into the next pass
        carry_bits = bits & 7;
        const uint32_t carry = carry_bits ? ((grp[nbytes >> 2] >> (24 - 8 * (nbytes & 3))) & 0xffu) & (0xff00u >> carry_bits) : 0u;
        __syncthreads();
        for (int i = lane; i < K7_GROUP; i += 64) grp[i] = 0;
        __syncthreads();
        if (lane == 0) grp[0] = carry << 24;
        __syncthreads();
    }
    if (lane == 0) {
        if (carry_bits) {                                          // pad the last byte with ones
            const uint32_t byte = ((grp[0] >> 24) & 0xffu) | (0xffu >> carry_bits);
            if (out_pos + 2 <= cap && !over) { dst[out_pos] = (uint8_t)byte; if (byte == 0xffu) dst[out_pos + 1] = 0; }
            out_pos += byte == 0xffu ? 2 : 1;
        }
        if (out_pos + 2 <= cap + 8 && !over) {
            dst[out_pos] = 0xff;
            dst[out_pos + 1] = my + 1 < a.mh ? (uint8_t)(0xd0 + (my & 7)) : 0xd9;      // RSTn between intervals, EOI after the last
        }
        out_pos += 2;
        a.isize[my] = out_pos;
        if (over || out_pos > a.stage_stride) atomicOr(a.flags, 1);
    }
}

__global__ void __launch_bounds__(256) k7_gather(K7Args a)
{
    const int my = blockIdx.x, tid = threadIdx.x;
    long off = a.header_len;
    for (int j = 0; j < my; ++j) off += a.isize[j];
    const int n = a.isize[my];
    const bool fits = off + n <= a.out_cap && n <= a.stage_stride;
    if (my == 0)
        for (int i = tid; i < a.header_len && i < a.out_cap; i += 256) a.out[i] = a.header[i];
    if (fits) {
        const uint8_t *src = a.stage + (size_t)my * a.stage_stride;
        for (int i = tid; i < n; i += 256) a.out[off + i] = src[i];
    } else if (tid == 0) {
        atomicOr(a.flags, 1);
    }
    if (my == a.mh - 1 && tid == 0) *a.out_size = (int32_t)(off + n);
}

void make_codes(const uint8_t *bits, const uint8_t *vals, uint32_t *tbl)     // jchuff.c jpeg_make_c_derived_tbl
{
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len - 1]; ++i) tbl[vals[k++]] = (code++ << 8) | (uint32_t)len;
        code <<= 1;
    }
}

int put_seg(uint8_t *h, int p, uint8_t marker, const uint8_t *payload, int n)
{
    h[p++] = 0xff; h[p++] = marker; h[p++] = (uint8_t)((n + 2) >> 8); h[p++] = (uint8_t)((n + 2) & 0xff);
    memcpy(h + p, payload, (size_t)n);
    return p + n;
}

}  // namespace

struct rva_jpeg_state {      // per-context scratch of the encoder (grown on demand, freed with the context)
    HuffTables *huff = nullptr;
    int16_t *coef = nullptr; size_t coef_bytes = 0;
    uint8_t *stage = nullptr; size_t stage_bytes = 0;
    int32_t *isize = nullptr; int isize_n = 0;
    int32_t *flags = nullptr;
};

void rva_jpeg_free(rva_ctx *ctx)
{
    rva_jpeg_state *s = ctx->jpeg;
    if (!s) return;
    (void)hipFree(s->huff); (void)hipFree(s->coef); (void)hipFree(s->stage); (void)hipFree(s->isize); (void)hipFree(s->flags);
    delete s;
    ctx->jpeg = nullptr;
}

extern "C" {

int rva_jpeg_max_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    const long mw = (width + 15) / 16, mh = (height + 15) / 16;
    return (int)(640 + mh * (mw * 6 * 128 + 16));                 // header + 128 B per block: ample for photographic content at quality <= 95
}

int rva_jpeg_encode_bgr(rva_ctx *ctx, const void *bgr, int pitch, int width, int height, int quality, void *out, int out_capacity,
                        int32_t *out_size, rva_stream_t stream)
{
    if (!ctx || !bgr || !out || !out_size || width <= 0 || height <= 0 || width > 65500 || height > 65500 || pitch < 3 * width ||
        quality < 1 || quality > 100 || out_capacity < 1024)
        return rva_fail(ctx, RVA_ERR_ARG, "rva_jpeg_encode_bgr: bad argument (quality 1..100, pitch >= 3 width, capacity >= 1024)");
    RVA_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->jpeg) {
        ctx->jpeg = new rva_jpeg_state();
        HuffTables h{};
        make_codes(kDcLumaBits, kDcVals, h.t[0]);
        make_codes(kDcChromaBits, kDcVals, h.t[1]);
        make_codes(kAcLumaBits, kAcLumaVals, h.t[2]);
        make_codes(kAcChromaBits, kAcChromaVals, h.t[3]);
        RVA_HIP(ctx, hipMalloc(&ctx->jpeg->huff, sizeof(HuffTables)));
        RVA_HIP(ctx, hipMemcpy(ctx->jpeg->huff, &h, sizeof(HuffTables), hipMemcpyHostToDevice));
        RVA_HIP(ctx, hipMalloc(&ctx->jpeg->flags, 4));
        RVA_HIP(ctx, hipMemset(ctx->jpeg->flags, 0, 4));
    }
    rva_jpeg_state *st = ctx->jpeg;
    K7Args a{};
    a.bgr = (const uint8_t *)bgr; a.pitch = pitch; a.w = width; a.h = height;
    a.mw = (width + 15) / 16; a.mh = (height + 15) / 16;
    const size_t nblk = (size_t)a.mw * a.mh * 6;
    a.stage_stride = a.mw * 6 * 128 + 16;
    if (nblk * 128 > st->coef_bytes || (size_t)a.mh * a.stage_stride > st->stage_bytes || a.mh > st->isize_n) {
        RVA_HIP(ctx, hipDeviceSynchronize());                      // a larger picture than before: regrow the scratch (not on the steady path)
        (void)hipFree(st->coef); (void)hipFree(st->stage); (void)hipFree(st->isize);
        st->coef = nullptr; st->stage = nullptr; st->isize = nullptr; st->coef_bytes = st->stage_bytes = 0; st->isize_n = 0;
        RVA_HIP(ctx, hipMalloc(&st->coef, nblk * 128)); st->coef_bytes = nblk * 128;
        RVA_HIP(ctx, hipMalloc(&st->stage, (size_t)a.mh * a.stage_stride)); st->stage_bytes = (size_t)a.mh * a.stage_stride;
        RVA_HIP(ctx, hipMalloc(&st->isize, (size_t)a.mh * 4)); st->isize_n = a.mh;
    }
    a.coef = st->coef; a.huff = st->huff; a.stage = st->stage; a.isize = st->isize; a.flags = st->flags;
    a.out = (uint8_t *)out; a.out_cap = out_capacity; a.out_size = out_size;
    // quantisation tables: jcparam.c jpeg_quality_scaling + jpeg_add_quant_table (force_baseline)
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t ql[64], qc[64];
    for (int i = 0; i < 64; ++i) {
        long l = ((long)kStdLumaQ[i] * scale + 50) / 100, c = ((long)kStdChromaQ[i] * scale + 50) / 100;
        ql[i] = (uint8_t)(l < 1 ? 1 : (l > 255 ? 255 : l));
        qc[i] = (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
        a.qdiv[0][i] = (uint16_t)(ql[i] << 3);
        a.qdiv[1][i] = (uint16_t)(qc[i] << 3);
    }
    // header: SOI, JFIF APP0, two DQT, SOF0 (4:2:0), four DHT, DRI (one MCU row), SOS
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                   47, 55, 62, 63};
    uint8_t *h = a.header;
    int p = 0;
    h[p++] = 0xff; h[p++] = 0xd8;
    const uint8_t app0[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    p = put_seg(h, p, 0xe0, app0, 14);
    uint8_t dqt[65];
    dqt[0] = 0; for (int i = 0; i < 64; ++i) dqt[1 + i] = ql[zz[i]];
    p = put_seg(h, p, 0xdb, dqt, 65);
    dqt[0] = 1; for (int i = 0; i < 64; ++i) dqt[1 + i] = qc[zz[i]];
    p = put_seg(h, p, 0xdb, dqt, 65);
    const uint8_t sof[15] = {8, (uint8_t)(height >> 8), (uint8_t)(height & 0xff), (uint8_t)(width >> 8), (uint8_t)(width & 0xff), 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
    p = put_seg(h, p, 0xc0, sof, 15);
    uint8_t dht[1 + 16 + 162];
    auto dht_seg = [&](uint8_t id, const uint8_t *bits, const uint8_t *vals, int nv) {
        dht[0] = id; memcpy(dht + 1, bits, 16); memcpy(dht + 17, vals, (size_t)nv);
        p = put_seg(h, p, 0xc4, dht, 17 + nv);
    };
    dht_seg(0x00, kDcLumaBits, kDcVals, 12);
    dht_seg(0x10, kAcLumaBits, kAcLumaVals, 162);
    dht_seg(0x01, kDcChromaBits, kDcVals, 12);
    dht_seg(0x11, kAcChromaBits, kAcChromaVals, 162);
    const uint8_t dri[2] = {(uint8_t)(a.mw >> 8), (uint8_t)(a.mw & 0xff)};
    p = put_seg(h, p, 0xdd, dri, 2);
    const uint8_t sos[10] = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
    p = put_seg(h, p, 0xda, sos, 10);
    a.header_len = p;
    if (p > (int)sizeof a.header) return rva_fail(ctx, RVA_ERR_ARG, "rva_jpeg_encode_bgr: header overflow");
    k7_transform<<<(unsigned)((nblk + 63) / 64), 64, 0, s>>>(a);
    k7_entropy<<<a.mh, 64, 0, s>>>(a);
    k7_gather<<<a.mh, 256, 0, s>>>(a);
    RVA_HIP(ctx, hipGetLastError());
    return RVA_OK;
}

int rva_jpeg_status(rva_ctx *ctx, rva_stream_t stream, int *flags)
{
    if (!ctx || !flags) return RVA_ERR_ARG;
    *flags = 0;
    if (!ctx->jpeg) return RVA_OK;
    RVA_HIP(ctx, hipSetDevice(ctx->device));
    RVA_HIP(ctx, hipMemcpyAsync(flags, ctx->jpeg->flags, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    RVA_HIP(ctx, hipMemsetAsync(ctx->jpeg->flags, 0, 4, (hipStream_t)stream));
    RVA_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return RVA_OK;
}

}  // extern "C"

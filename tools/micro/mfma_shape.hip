// Micro-benchmark (round 4): does v_mfma_f32_32x32x16_f16 relieve the LDS fragment reads of the 3x3 kernels' MFMA phase?
//
// The MFMA phase of k_conv3_run<256,128> as it stands: eight waves, wave tile 64 px x 64 ch, per K-step of 96 (three
// horizontal taps x 32 channels) every wave reads its A rows (64 weight rows) and B rows (64 pixel rows) of the step from LDS
// with ds_read_b128 and issues the MFMAs; no DMA, no barrier here -- only the phase the verdict calls "at the LDS wall".
// Both shapes are built on the SAME wave tile, the same 64-byte-row swizzled LDS image and the same K per step:
//   16x16x32: per 32 of K  4 A + 4 B fragments (1 KiB each) -> 16 MFMAs of 16 cycles;
//   32x32x16: per 16 of K  2 A + 2 B fragments (1 KiB each) ->  4 MFMAs of 32 cycles, i.e. per 32 of K 4 A + 4 B -> 8 MFMAs.
// LDS bytes per MAC are a property of the WAVE TILE ((TM + TN) K 2 bytes for TM TN K MACs), not of the instruction shape: both
// read 8 KiB per 131 072 MACs.  What differs is the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back item 7) and the
// accumulator layout.  Random operands (rule 25).  Prints cycles-equivalent time, TFLOP/s and LDS bytes per MFMA-cycle.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int swz32(int row, int chunk) { return row * 32 + (((chunk + 2 * (row >> 2)) & 3) << 3); }

// LDS image: A = [3 taps][128 rows][32 halfs], B = [256 + 2 rows][32 halfs]  (one K-step of the run kernel), filled once
constexpr int AROWS = 3 * 128, BROWS = 272;

template <int SHAPE, int WTM, int WTN>   // SHAPE 16 or 32; wave tile WTM px x WTN ch
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) k(const __half *src, int steps, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half *A = (__half *)smem, *B = A + AROWS * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 100 * 1024 / 16; i += 512)       // the whole allocation holds random halfs (the 128-row wave tiles read past BROWS)
        reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(src)[(i + blockIdx.x * 977) % ((AROWS + BROWS) * 4)];
    __syncthreads();
    constexpr int WGN = 128 / WTN, WGM = 8 / WGN;
    const int wm = wv % WGM, wn = wv / WGM;
    if constexpr (SHAPE == 16) {
        constexpr int FM = WTM / 16, FN = WTN / 16;
        f4 acc[FN][FM];
        for (int i = 0; i < FN; ++i) for (int j = 0; j < FM; ++j) acc[i][j] = f4{0, 0, 0, 0};
        const int ch = lane >> 4;
        const int wl = swz32(wn * WTN + (lane & 15), ch);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ab = swz32(wm * WTM + (lane & 15) + dx, ch);
                h8 bf[FM], af[FN];
#pragma unroll
                for (int j = 0; j < FM; ++j) bf[j] = *reinterpret_cast<const h8 *>(B + ab + j * 16 * 32);
#pragma unroll
                for (int i = 0; i < FN; ++i) af[i] = *reinterpret_cast<const h8 *>(A + wl + (dx * 128 + i * 16) * 32);
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        float t = 0;
        for (int i = 0; i < FN; ++i) for (int j = 0; j < FM; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sink) sink[blockIdx.x * 512 + tid] = t;
    } else {
        constexpr int FM = WTM / 32, FN = WTN / 32;
        f16v acc[FN][FM];
        for (int i = 0; i < FN; ++i) for (int j = 0; j < FM; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
        // operand of v_mfma_f32_32x32x16_f16: lane l holds row l % 32, k = 8 (l / 32) .. + 8 of the 16-deep slice
        const int r32 = lane & 31, kh = lane >> 5;
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {                 // two 16-deep slices of the 32-channel chunk
                    h8 bf[FM], af[FN];
#pragma unroll
                    for (int j = 0; j < FM; ++j) bf[j] = *reinterpret_cast<const h8 *>(B + swz32(wm * WTM + j * 32 + r32 + dx, ks * 2 + kh));
#pragma unroll
                    for (int i = 0; i < FN; ++i) af[i] = *reinterpret_cast<const h8 *>(A + swz32(dx * 128 + wn * WTN + i * 32 + r32, ks * 2 + kh));
#pragma unroll
                    for (int i = 0; i < FN; ++i)
#pragma unroll
                        for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        float t = 0;
        for (int i = 0; i < FN; ++i) for (int j = 0; j < FM; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (sink) sink[blockIdx.x * 512 + tid] = t;
    }
}

int main()
{
    hipSetDevice(0);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t n = (size_t)(AROWS + BROWS) * 32;
    std::vector<__half> h(n);
    srand(7);
    for (auto &v : h) v = __float2half((float)rand() / RAND_MAX * 2.f - 1.f);
    __half *src; hipMalloc(&src, n * 2); hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
    float *sink; hipMalloc(&sink, (size_t)cus * 512 * 4);
    const size_t smem = 100 * 1024;                      // one workgroup per CU, two waves per SIMD: the run kernel's residency
    const int steps = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d; 512-thread workgroup per CU, %d K-steps of 96 per wave, operands uniform [-1,1); LDS-fed, no DMA, no barrier\n", cus, steps);
    printf("shape      wave tile   ms        TFLOP/s   MFMA-cycles/step/SIMD   LDS bytes/step/CU   LDS B per MFMA-cycle per CU\n");
    auto run = [&](const char *name, auto kern, int wtm, int wtn) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(kern, dim3(cus), dim3(512), smem, 0, src, steps / 4, sink);      // warm the clocks
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(cus), dim3(512), smem, 0, src, steps, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            const double macs = (double)cus * 8 * wtm * wtn * 96.0 * steps;
            const double mfma_cyc = 2.0 * wtm * wtn * 96.0 / 512.0;                        // two waves per SIMD, 512 MAC/clk/SIMD
            const double lds = 8.0 * (wtm + wtn) * 96.0 * 2.0;
            printf("%-10s %3dx%-3d     %7.3f   %7.1f   %8.0f                %8.0f            %6.1f\n", name, wtm, wtn, ms, 2 * macs / (ms * 1e-3) / 1e12,
                   mfma_cyc, lds, lds / mfma_cyc);
        }
    };
    run("16x16x32", k<16, 64, 64>, 64, 64);
    run("32x32x16", k<32, 64, 64>, 64, 64);
    run("16x16x32", k<16, 64, 32>, 64, 32);          // the BN = 64 run kernels' wave tile (no 32x32 form of it worth building: FM x FN = 2 x 1)
    run("32x32x16", k<32, 64, 32>, 64, 32);
    return 0;
}

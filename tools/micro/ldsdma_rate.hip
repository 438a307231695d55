// Micro-benchmark: per-CU LDS-DMA (global_load_lds_dwordx4) rate from an L2-resident table as a function of the row
// segment a 1 KiB piece is made of: 16 rows x 64 B (what the 32-channel K-steps of k_conv3_big/run stage), 8 rows x 128 B
// (64-channel steps: whole 128-byte lines), 4 x 256 B, and 1 x 1 KiB contiguous.  One 512-thread workgroup per CU, every
// wave issues NP pieces per round into its own LDS region, waits vmcnt(0), repeats.  Source rows are spaced `pitch` bytes
// apart (a pixel row of an NHWC tensor), the table is `tbl_bytes` large and shared by all workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_vptr;
__device__ __forceinline__ void lds_dma16(unsigned voff, const void *sbase, unsigned m0v)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory");
}

template <int SEG>   // bytes of one contiguous source segment: 64, 128, 256, 1024
__global__ void __launch_bounds__(512) k(const char *tbl, unsigned tbl_bytes, unsigned pitch, int rounds, int np, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lds_vptr)smem;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = SEG / 16;            // lanes per segment
    const int seg = lane / LPR, part = lane % LPR;
    unsigned base = (blockIdx.x * 7919u * 1024u) % tbl_bytes;
    for (int r = 0; r < rounds; ++r) {
        for (int p = 0; p < np; ++p) {
            // piece p of this wave: 64/LPR segments, consecutive rows `pitch` apart
            unsigned row0 = (base + (unsigned)((wv * np + p) * (64 / LPR)) * pitch) % (tbl_bytes - 64u * pitch);
            unsigned voff = row0 + (unsigned)seg * pitch + (unsigned)part * 16u;
            lds_dma16(voff, tbl, lds0 + (unsigned)(wv * np + p) * 1024u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        base = (base + 8u * np * 16u * pitch) % tbl_bytes;
    }
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = ((float *)smem)[0];
}

int main(int argc, char **argv)
{
    int dev = 0; hipSetDevice(dev);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    const unsigned tbl_bytes = 2u << 20;        // 2 MiB: stays in every XCD's 4 MiB L2
    char *tbl; hipMalloc(&tbl, tbl_bytes + (1 << 20)); hipMemset(tbl, 1, tbl_bytes + (1 << 20));
    float *sink; hipMalloc(&sink, cus * 4);
    const int rounds = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d, table %u KiB (L2-resident), 512-thread workgroup per CU, rounds %d\n", cus, tbl_bytes >> 10, rounds);
    printf("segment  pitch  pieces/wave/round   us      GB/s per CU   B/clk/CU@2.1GHz   chip TB/s\n");
    for (int np : {2, 5}) {
        for (int seg : {64, 128, 256, 1024}) {
            for (unsigned pitch : {256u, 512u}) {
                if ((unsigned)seg > pitch && seg != 1024) continue;
                unsigned pt = seg == 1024 ? 1024u : pitch;
                auto launch = [&](int rr) {
                    const size_t sm = (size_t)8 * np * 1024;
                    switch (seg) {
                    case 64: hipLaunchKernelGGL(k<64>, dim3(cus), dim3(512), sm, 0, tbl, tbl_bytes, pt, rr, np, sink); break;
                    case 128: hipLaunchKernelGGL(k<128>, dim3(cus), dim3(512), sm, 0, tbl, tbl_bytes, pt, rr, np, sink); break;
                    case 256: hipLaunchKernelGGL(k<256>, dim3(cus), dim3(512), sm, 0, tbl, tbl_bytes, pt, rr, np, sink); break;
                    default: hipLaunchKernelGGL(k<1024>, dim3(cus), dim3(512), sm, 0, tbl, tbl_bytes, pt, rr, np, sink); break;
                    }
                };
                launch(20);
                hipDeviceSynchronize();
                hipEventRecord(e0); launch(rounds); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes_cu = (double)rounds * 8 * np * 1024;
                const double gbs = bytes_cu / (ms * 1e-3) / 1e9;
                printf("%5d B  %4u   %2d                %8.1f  %8.1f      %8.1f        %6.2f\n", seg, pt, np, ms * 1e3, gbs, gbs / 2.1, gbs * cus / 1e3);
                if (seg == 1024) break;
            }
        }
    }
    return 0;
}

// L2 -> LDS fill-rate microbenchmark (round 6): what the chip delivers into LDS through global_load_lds_dwordx4 when EVERY CU
// streams tile-shaped pieces (8 rows x 128 B, row stride 2560 B = a K = 1280 f16 operand) with a counted number of 1-KiB pieces in
// flight per wave - the access pattern of every LDS-DMA GEMM / conv kernel of this repository, without any MFMA or fragment read.
// Swept: waves per CU (4 .. 16), KiB in flight per CU, and where the data lives (a 2-MiB working set per XCD = L2 hits, 32 MiB per
// XCD = Infinity Cache, 512 MiB per XCD = HBM).  Prints aggregate TB/s and GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_fill_rate tools/ubench/lds_fill_rate.hip && /tmp/lds_fill_rate
// Why: DESIGN.md section 8 prices the "two 256 x 128 blocks per CU" GEMM (1.5 x the fill bytes per flop of the 256 x 256 tile)
// against this ceiling, and gemm_t160_kernel (one block per CU at UNet batch 2) sits on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void lds_dma16(const void* base, unsigned byte_off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(byte_off), "s"(base), "s"(lds_addr) : "memory", "m0");
}

template <int DEPTH>   // 1-KiB pieces in flight per wave
__global__ void fill_kernel(const char* base, unsigned long long per_xcd_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + wave * DEPTH * 1024;
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    // a piece = 8 rows x 128 B of a row-major [rows][2560 B] matrix; the XCD's region holds per_xcd_bytes / 2560 rows
    const unsigned long long rows = per_xcd_bytes / 2560ull;
    const char* region = base + (unsigned long long)xcd * per_xcd_bytes;
    const unsigned lane_off = (unsigned)((lane >> 3) * 2560 + (lane & 7) * 16);
    // every wave walks the region in its own order; the CUs of an XCD overlap (that is what makes an L2 hit)
    unsigned long long piece = ((unsigned long long)cu * 7919ull + (unsigned long long)wave * 104729ull) % (rows / 8 * 20);
    const unsigned long long npieces = rows / 8 * 20;   // 20 column chunks of 128 B per 8-row band
    auto src = [&](unsigned long long q) -> const char* {
        const unsigned long long band = q / 20, col = q % 20;
        return region + band * 8ull * 2560ull + col * 128ull;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        lds_dma16(src(piece), lane_off, lds0 + d * 1024);
        piece = piece + nwaves * 3 + 1 < npieces ? piece + nwaves * 3 + 1 : (piece + nwaves * 3 + 1) % npieces;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");   // the oldest piece has landed: its slot is free
            lds_dma16(src(piece), lane_off, lds0 + d * 1024);
            piece = piece + nwaves * 3 + 1 < npieces ? piece + nwaves * 3 + 1 : (piece + nwaves * 3 + 1) % npieces;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);
}

template <int DEPTH>
static double run(const char* buf, unsigned long long per_xcd, int waves, int blocks_per_cu, int iters, unsigned* sink) {
    const int grid = 256 * blocks_per_cu;
    const size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(grid), dim3(waves * 64), lds, 0, buf, per_xcd, iters / 4, sink);   // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(grid), dim3(waves * 64), lds, 0, buf, per_xcd, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * waves * (double)(iters + 1) * DEPTH * 1024.0;
    return bytes / (ms * 1e-3) / 1e12;
}

int main() {
    const unsigned long long total = 8ull * 512ull * 1024 * 1024;   // 4 GiB: 512 MiB per XCD at most
    char* buf = nullptr;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, total);
    unsigned* sink = nullptr;
    hipMalloc(&sink, 4096 * 4);
    struct Where { const char* name; unsigned long long per_xcd; } where[3] = {
        {"L2 (2 MiB per XCD)", 2ull << 20}, {"Infinity Cache (24 MiB per XCD)", 24ull << 20}, {"HBM (512 MiB per XCD)", 512ull << 20}};
    printf("# L2 -> LDS fill rate, 256 CUs, 1-KiB LDS-DMA pieces (8 rows x 128 B, row stride 2560 B); TB/s aggregate (GB/s per CU)\n");
    for (auto& w : where) {
        printf("## %s\n", w.name);
        printf("%-34s %12s %12s %12s %12s\n", "waves per CU x blocks per CU", "4 KiB/wave", "8 KiB/wave", "16 KiB/wave", "24 KiB/wave");
        const int cfgs[6][2] = {{4, 1}, {5, 1}, {8, 1}, {4, 2}, {4, 3}, {16, 1}};
        for (auto& c : cfgs) {
            const int waves = c[0], bpc = c[1];
            char label[64];
            snprintf(label, sizeof label, "%2d waves x %d block%s (%2d waves/CU)", waves, bpc, bpc > 1 ? "s" : " ", waves * bpc);
            printf("%-34s", label);
            const int iters = 400;
            double r[4] = {0, 0, 0, 0};
            r[0] = run<4>(buf, w.per_xcd, waves, bpc, iters, sink);
            r[1] = run<8>(buf, w.per_xcd, waves, bpc, iters / 2, sink);
            if ((size_t)waves * bpc * 16 * 1024 <= 160 * 1024) r[2] = run<16>(buf, w.per_xcd, waves, bpc, iters / 4, sink);
            if ((size_t)waves * bpc * 24 * 1024 <= 160 * 1024) r[3] = run<24>(buf, w.per_xcd, waves, bpc, iters / 6, sink);
            for (int k = 0; k < 4; ++k) {
                if (r[k] > 0) printf(" %6.2f (%4.0f)", r[k], r[k] * 1e3 / 256);
                else printf(" %13s", "-");
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}

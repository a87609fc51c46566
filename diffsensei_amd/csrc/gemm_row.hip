// fp16 MFMA GEMM for the 640-channel level of the UNet: ONE block computes 128 rows x ALL 640 columns.
//
// Why.  The K = N = 640 projections of the 64 x 64-token transformers (attn out-projections + residual, attn2.to_q, proj_in /
// proj_out; reference: the nn.Linear layers of diffusers' BasicTransformerBlock / Transformer2DModel [3P] reached from
// src/models/unet.py:244-338) ran on the 128 x 128 kernels of gemm.hip at 0.26 of the MFMA peak and 0.27 of HBM (round-3
// verdict): arithmetic intensity 213 flop/B sits on the ridge, and five column tiles re-read every A row panel through L2.
// The 256 x 256 kernel wastes a sixth of its tiles on N = 640 (2.5 tile columns).  Here A is streamed exactly once (a block
// owns 128 complete rows), W (0.8 MB at K = 640, 3.3 MB at K = 2560) is the only operand re-read - from L2 - and the output
// leaves as whole 1280-byte rows.
//
//   * 512 threads = 8 waves = 2 (M) x 4 (N); wave tile 64 rows x 160 columns = 2 x 5 accumulator blocks of 32 x 32 (160 VGPRs);
//     operands swapped like the other kernels (first MFMA operand = W rows), so a lane ends with a tile ROW.
//   * k-tile = 32: A 128 x 64 B = 8 KiB + W 640 x 64 B = 40 KiB per stage, ring of THREE stages (144 KiB) filled by LDS-DMA
//     (six 1-KiB pieces per wave and stage); rows of 64 bytes, 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3): the four rows
//     that share a 256-byte bank row use four different slots - conflict-free ds_read_b128; the swizzle is applied to the
//     lane's SOURCE chunk, the DMA destination is lane-linear.
//   * one barrier per k-tile: wait (counted, vmcnt(6): everything but the newest stage) -> barrier -> refill the slot consumed
//     one iteration ago (stage t + 2) -> 14 fragment reads + 20 MFMAs.  Past the end of K the refill re-fetches the last
//     k-tile into slots nobody reads, so the loop has no branch and the count never changes.
//   * epilogue in two passes of 32 rows per wave through a wave-private LDS tile (the ring is free by then): + bias, round to
//     f16, transpose, + residual, 16-byte stores over whole rows.  Same arithmetic order as gemm_glds_kernel / gemm_pp_kernel:
//     bit-identical to them (tests/test_gpu_ops.py::test_gemm_row640_kernel).
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr int RN = 640, RM = 128, RBK = 32;
constexpr int A_ST = RM * RBK * 2;   // 8 KiB
constexpr int W_ST = RN * RBK * 2;   // 40 KiB
constexpr int ST = A_ST + W_ST;      // one stage
constexpr int NST = 3;
constexpr int EP_STRIDE = 336;       // bytes per row of the epilogue tile: 160 f16 + 16 B pad
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ __launch_bounds__(512, 2) void gemm_row640_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * RM;
    const int nk = p.K / RBK;

    // ---- staging: piece q of a stage = LDS rows 16 q .. 16 q + 15 (1 KiB); lane -> row (lane >> 2), slot (lane & 3)
    const int srow = lane >> 2, sslot = lane & 3;
    const int schunk = sslot ^ ((srow >> 2) & 3);   // (16 q + srow) >> 2 & 3 == (srow >> 2) & 3
    const half_t* const a_src = p.A + (long)(m0 + wave * 16 + srow) * p.lda + schunk * 8;
    const half_t* w_src[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) w_src[j] = p.W + (long)((wave + 8 * j) * 16 + srow) * p.ldw + schunk * 8;
    auto issue = [&](int kt, int slot) {
        char* d = smem + slot * ST;
        __builtin_amdgcn_global_load_lds((glb_void*)(a_src + kt * RBK), (lds_void*)(d + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(w_src[j] + kt * RBK), (lds_void*)(d + A_ST + (wave + 8 * j) * 1024), 16, 0, 0);
    };

    // ---- fragment addresses (bytes inside a stage): row r at r * 64, chunk c at slot c ^ ((r >> 2) & 3)
    unsigned fa[2][2], fw[5][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int r = wr * 64 + mi * 32 + l31;
            fa[mi][kk] = r * 64 + (((kk * 2 + lhi) ^ ((r >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) {
            const int r = wc * 160 + ni * 32 + l31;
            fw[ni][kk] = A_ST + r * 64 + (((kk * 2 + lhi) ^ ((r >> 2) & 3)) << 4);
        }
    }

    f32x16 acc[2][5];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 5; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue(0, 0);
    issue(nk > 1 ? 1 : 0, 1);
    int slot = 0, refill = 2;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // stage kt has landed (this wave's pieces); stage kt + 1 may be in flight
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // ... everybody's; and everybody is done reading stage kt - 1
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue(min(kt + 2, nk - 1), refill);
        const char* s = smem + slot * ST;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 af[2], wf[5];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const h8*>(s + fa[mi][kk]);
#pragma unroll
            for (int ni = 0; ni < 5; ++ni) wf[ni] = *reinterpret_cast<const h8*>(s + fw[ni][kk]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 5; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        slot = slot == NST - 1 ? 0 : slot + 1;
        refill = refill == NST - 1 ? 0 : refill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the over-fetched stages have landed: the ring can be reused
    __syncthreads();

    // ---- epilogue.  D layout (operands swapped): lane holds tile row ... + l31; register r of block ni is column
    // 32 ni + (r & 3) + 8 (r >> 2) + 4 lhi of the wave's 160.
    char* const ep = smem + wave * (32 * EP_STRIDE);
    const half_t* const Rg = p.residual;
    const int ncol0 = wc * 160;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int mrow0 = m0 + wr * 64 + mi * 32;
#pragma unroll
        for (int ni = 0; ni < 5; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ni * 32 + 8 * g + 4 * lhi;
                h4 bv = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                if (p.bias) bv = *reinterpret_cast<const h4*>(p.bias + ncol0 + c);
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)(acc[mi][ni][4 * g + e] + (float)bv[e]);
                *reinterpret_cast<h4*>(ep + l31 * EP_STRIDE + c * 2) = o;
            }
#pragma unroll
        for (int j = 0; j < 10; ++j) {   // 32 rows x 20 sixteen-byte chunks = 640 = 10 per lane
            const int idx = lane + 64 * j;
            const int row = idx / 20, ch = idx - row * 20;
            h8 v = *reinterpret_cast<const h8*>(ep + row * EP_STRIDE + ch * 16);
            const long off = (long)(mrow0 + row);
            if (Rg) {
                const h8 rv = *reinterpret_cast<const h8*>(Rg + off * p.ldr + ncol0 + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<h8*>(p.C + off * p.ldc + ncol0 + ch * 8) = v;
        }
    }
}

int g_row_variant = 0;  // 0 auto, 1 never (A/B), 2 always where applicable

}  // namespace

void ds_gemm_row_set_variant(int v) { g_row_variant = v; }

// Shapes the kernel takes: N = 640 exactly, plain epilogue (bias / residual), M a multiple of 128, K a multiple of 32 with at
// least two k-tiles, a single A source, no batch.  ds_launch_gemm decides when it is also the faster choice.
bool ds_gemm_row_applicable(const GemmParams& p, int batch) {
    return p.conv == 0 && p.A2 == nullptr && batch == 1 && p.N == RN && p.M % RM == 0 && p.K % RBK == 0 && p.K >= 2 * RBK &&
           p.epi == EPI_NONE && p.rowbias == nullptr && p.dtype == DS_DTYPE_F16 && p.ln_stats == nullptr && p.stats_out == nullptr &&
           p.lda % 8 == 0 && p.ldw % 8 == 0 && p.ldc % 8 == 0 && (p.residual == nullptr || p.ldr % 8 == 0);
}
bool ds_gemm_row_preferred(const GemmParams& p, int batch) {
    if (g_row_variant == 1 || !ds_gemm_row_applicable(p, batch)) return false;
    return g_row_variant == 2 || p.M / RM >= 512;   // >= two rounds of one block per CU
}

int ds_launch_gemm_row(const GemmParams& p, hipStream_t stream) {
    DS_REQUIRE(ds_gemm_row_applicable(p, 1), "gemm_row640: shape M=%d N=%d K=%d not supported", p.M, p.N, p.K);
    const size_t lds = (size_t)NST * ST;
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs))
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_row640_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gemm_row640_kernel, dim3(p.M / RM), dim3(512), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

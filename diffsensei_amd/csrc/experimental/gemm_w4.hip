// EXPERIMENTAL - not part of the production library, never dispatched by ds_launch_gemm.  Written at the end of round 2
// (no GPU minutes left): it compiles for gfx950 and its register / instruction budget has been read off the assembly, but
// it has NOT run yet (its address arithmetic has: tools/w4_index_model.py replays staging, fragment reads, the MFMA
// layout and the epilogue per thread in numpy).  `python -m diffsensei_amd.build --experimental` adds it behind
// gemm_variant 12; tests/test_gpu_ops.py::test_gemm_w4_experimental and tools/w4_check.py are the first contact.
//
// fp16 MFMA GEMM, 256 x 256 x 64 block tile, FOUR waves (2 x 2), wave tile 128 x 128, one wave per SIMD, persistent.
// Same GemmParams / epilogue semantics as gemm.hip and gemm_pp.hip for the shapes it takes (M, N multiples of 256,
// K of 64, bias / residual / GEGLU).
//
// Why.  gemm_pp_kernel (8 waves, 128 x 64 wave tiles, two waves per SIMD taking turns) sits at the level of hipBLASLt on
// the UNet's shapes (1.10-1.15 PFLOP/s) with the matrix pipe busy 0.61 of the time; its k-tile is eight barrier
// intervals, every fragment byte is read by two waves, and its operands arrive by LDS-DMA, whose issue cost (~60 cycles
// a piece next to MFMAs) only fits because the partner wave of the SIMD computes meanwhile.  This kernel is the other
// classic point of the design space - what hipBLASLt's MT256x256x64 / 256-thread kernels and CK's 128x128-per-wave
// pipelines do: 256 accumulator registers per lane (the unified 512-entry file at one wave per SIMD), 0.5 KiB of
// fragment reads per MFMA instead of 0.75, ONE barrier per k-tile (64 MFMAs), operands staged global -> registers ->
// LDS (a 13-cycle ds_write_b128 fits an MFMA gap; an LDS-DMA piece does not), and every wait left to the compiler:
// plain loads and ds operations are on its scoreboard, so there is no hand-counted s_waitcnt in this file.
//
//   LDS   two stages x (A 256 rows x 128 B | B 256 rows x 128 B) = 128 KiB, rows XOR-swizzled like every other tile
//         (chunk c of row r at slot c ^ ((r>>1)&7)), + 4 x 8 KiB wave-private epilogue staging = 160 KiB.
//   k-tile kt (stage b = kt & 1), four k-steps of 16 MFMAs; the fragments of step s+1 are read under the MFMAs of s:
//         steps 0..2  write k-tile kt+1 from the staging registers into stage b^1 (16 ds_write_b128 per lane) and, as
//                     each staging register frees up, request k-tile kt+2 into it (16 global_load_dwordx4): one whole
//                     k-tile (~2000 cycles) of latency cover with 64 staging registers;
//         barrier     after step 2: every read of stage b has returned (step 3's fragments were fetched under step 2),
//                     every wave's writes to b^1 have landed;
//         step 3      MFMAs from registers while the step-0 fragments of k-tile kt+1 are read from stage b^1.
//   The k-tile stream runs ACROSS output tiles: the loads behind a tile's last k-tiles are the next tile's first ones, so
//   when the epilogue starts the next tile's k-tile 0 is in LDS (its first fragments in registers) and k-tile 1 in the
//   staging registers - the matrix pipe only idles for the epilogue itself.
#include "../ds_common.h"
#include "../ds_kernels.h"

namespace {

template <int V>
struct IC {
    static constexpr int value = V;
};

constexpr int W4_STAGE = 65536;          // bytes per stage: A tile | B tile
constexpr int W4_B = 32768;              // offset of the B tile inside a stage
constexpr int W4_EP = 2 * W4_STAGE;      // wave-private epilogue staging: 4 x 8 KiB

template <bool GEGLU>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const long bz = blockIdx.z;
    const int nk = p.K / 64;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int tile_local = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;

    // ---- staging: thread t moves 16 bytes (8 k values) of row 32 j + (t >> 3), j < 8, of both operand tiles
    const int lrow = tid >> 3, chunk = tid & 7;
    const unsigned w0 = lrow * 128 + ((chunk ^ ((lrow >> 1) & 7)) << 4);  // (32 j + lrow) >> 1 & 7 == (lrow >> 1) & 7
    // ---- fragments: row r of a tile at r * 128, 16-byte chunk c at slot c ^ ((r >> 1) & 7)
    unsigned fa[4], fb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned o = l31 * 128 + (((kk * 2 + lhi) ^ ((l31 >> 1) & 7)) << 4);
        fa[kk] = wr * 16384 + o;
        fb[kk] = W4_B + wc * 16384 + o;
    }

    // ---- the k-tile stream: (ld_id, ld_kt) is the next k-tile to request
    int ld_id = tile_local, ld_kt = 0;
    if (ld_id >= ntiles) return;
    const half_t *ld_a, *ld_b;
    auto set_ptrs = [&](int id) {
        int tm, tn;
        tile_coords(id, p.tiles_m, p.tiles_n, tm, tn);
        ld_a = p.A + bz * p.sA + (long)(tm * 256 + lrow) * p.lda + chunk * 8;
        ld_b = p.W + bz * p.sW + (long)(tn * 256 + lrow) * p.ldw + chunk * 8;
    };
    set_ptrs(ld_id);
    h8 sa[8], sb[8];  // staging registers: one k-tile of both operands
    auto gload = [&](int j) {
        sa[j] = *reinterpret_cast<const h8*>(ld_a + (long)j * 32 * p.lda + ld_kt * 64);
        sb[j] = *reinterpret_cast<const h8*>(ld_b + (long)j * 32 * p.ldw + ld_kt * 64);
    };
    auto advance = [&]() {  // past the last tile the stream keeps re-reading its final k-tile (never consumed)
        if (++ld_kt == nk) {
            if (ld_id + G < ntiles) {
                ld_id += G;
                ld_kt = 0;
                set_ptrs(ld_id);
            } else {
                ld_kt = nk - 1;
            }
        }
    };
    auto lwrite = [&](int stage, int j) {
        *reinterpret_cast<h8*>(smem + stage * W4_STAGE + j * 4096 + w0) = sa[j];
        *reinterpret_cast<h8*>(smem + stage * W4_STAGE + W4_B + j * 4096 + w0) = sb[j];
    };
    auto fread = [&](int stage, int kk, h8 (&a)[4], h8 (&b)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const h8*>(smem + stage * W4_STAGE + fa[kk] + i * 4096);
            b[i] = *reinterpret_cast<const h8*>(smem + stage * W4_STAGE + fb[kk] + i * 4096);
        }
    };
    // One k-step = 16 MFMAs + 8 fragment reads (+ NW LDS writes and NW global loads): pin them as 8 x {2 MFMA, 1 read,
    // [1 write, 1 load]} so that the matrix pipe never waits for the wave to get through a block of memory instructions
    // (one wave per SIMD: nobody else issues MFMAs meanwhile; <= 5 single-issue instructions fit an MFMA's 32 cycles).
    auto interleave = [&](auto nwc) {
        constexpr int NW = decltype(nwc)::value;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                 // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 // DS read
            if (g < NW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
            if (g < NW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc[4][4];
    auto mma = [&](h8 (&a)[4], h8 (&b)[4]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
    };

    // one k-step: the 16 MFMAs of (ac, bc) with, in issue order, the 8 fragment reads of k-step `kk` of stage `rs` into
    // (an, bn) and NW / 2 staging pieces (first J0): written to stage `ws`, then re-requested from the stream
    auto step = [&](int rs, int kk, h8 (&an)[4], h8 (&bn)[4], h8 (&ac)[4], h8 (&bc)[4], int ws, auto j0c, auto nwc) {
        constexpr int J0 = decltype(j0c)::value, NW = decltype(nwc)::value;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < 4) an[g] = *reinterpret_cast<const h8*>(smem + rs * W4_STAGE + fa[kk] + g * 4096);
            else bn[g - 4] = *reinterpret_cast<const h8*>(smem + rs * W4_STAGE + fb[kk] + (g - 4) * 4096);
            if (g < NW) {
                const int j = J0 + g / 2;
                if ((g & 1) == 0) {
                    *reinterpret_cast<h8*>(smem + ws * W4_STAGE + j * 4096 + w0) = sa[j];
                    sa[j] = *reinterpret_cast<const h8*>(ld_a + (long)j * 32 * p.lda + ld_kt * 64);
                } else {
                    *reinterpret_cast<h8*>(smem + ws * W4_STAGE + W4_B + j * 4096 + w0) = sb[j];
                    sb[j] = *reinterpret_cast<const h8*>(ld_b + (long)j * 32 * p.ldw + ld_kt * 64);
                }
            }
        }
        mma(ac, bc);
        interleave(nwc);
    };

    // ---- stream prologue: k-tile 0 into stage 0, k-tile 1 into the staging registers
#pragma unroll
    for (int j = 0; j < 8; ++j) gload(j);
    advance();
#pragma unroll
    for (int j = 0; j < 8; ++j) lwrite(0, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) gload(j);
    advance();
    __syncthreads();
    h8 a0[4], b0[4], a1[4], b1[4];
    fread(0, 0, a0, b0);
    int stage = 0;

    half_t* const Cg = p.C + bz * p.sC;
    const half_t* const Rg = p.residual ? p.residual + bz * p.sR : nullptr;
    char* const ep = smem + W4_EP + wave * 8192;

    for (int id = tile_local; id < ntiles; id += G) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            const int other = stage ^ 1;
            // (reads, writes and loads are written in the order they should issue: the compiler cannot tell the two stages
            // apart, so it keeps LDS reads and writes in program order - eight reads followed by six writes would reach
            // the matrix pipe as one block of fourteen memory instructions)
            step(stage, 1, a1, b1, a0, b0, other, IC<0>{}, IC<6>{});
            step(stage, 2, a0, b0, a1, b1, other, IC<3>{}, IC<6>{});
            step(stage, 3, a1, b1, a0, b0, other, IC<6>{}, IC<4>{});
            __syncthreads();  // stage `other` complete everywhere; nobody reads `stage` any more
            step(other, 0, a0, b0, a1, b1, other, IC<0>{}, IC<0>{});
            // pin the fragments just read in front of the branch below (the compiler otherwise sinks the eight reads past
            // it, behind this step's MFMAs: the next k-tile then starts by waiting for them)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a0[i]), "+v"(b0[i]));
            advance();        // (a branch: kept out of the scheduled regions) the next k-tile of the stream
            stage = other;
        }

        // ---- epilogue (the stream stands: next k-tile 0 in LDS, its first fragments in a0 / b0, k-tile 1 in sa / sb)
        // lane-derived addresses are rebuilt per tile from an opaque v_mbcnt: hoisted out of the tile loop they are spilled
        unsigned zz = 0;
        asm volatile("" : "+v"(zz));
        const int le = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zz));
        const int l31 = le & 31, lhi = le >> 5;
        int tm, tn;
        tile_coords(id, p.tiles_m, p.tiles_n, tm, tn);
        const int m0 = tm * 256 + wr * 128, n0 = tn * 256 + wc * 128;
        // D layout (operands swapped): lane holds row ..+l31; register r of a 32 x 32 block is column
        // (r & 3) + 8 (r >> 2) + 4 lhi.  A 32-row piece goes through the wave's 8 KiB of LDS so that stores and residual
        // loads are whole 16-byte pieces of rows.
        constexpr int NO = GEGLU ? 2 : 4;            // 32-column blocks of output per wave
        constexpr int ROWB = NO * 64;                // bytes per staged row
        const int no = GEGLU ? (n0 >> 1) : n0;       // first output column of the wave
        h4 bias[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bias[ni][g] = p.bias ? *reinterpret_cast<const h4*>(p.bias + n0 + ni * 32 + 8 * g + 4 * lhi) : h4{0, 0, 0, 0};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NO; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (GEGLU) {  // packed weights: the wave's 128 columns = 64 hidden | their 64 gates
                            const float hq = (float)(half_t)(acc[mi][ni][4 * g + e] + (float)bias[ni][g][e]);
                            const float gq = (float)(half_t)(acc[mi][ni + 2][4 * g + e] + (float)bias[ni + 2][g][e]);
                            o[e] = (half_t)(hq * (float)(half_t)ds_gelu_erf(gq));
                        } else {
                            o[e] = (half_t)(acc[mi][ni][4 * g + e] + (float)bias[ni][g][e]);
                        }
                    }
                    const int c16 = ni * 4 + g;  // 16-byte chunk of the staged row; the lane's 8 bytes are its half lhi
                    *reinterpret_cast<h4*>(ep + l31 * ROWB + ((c16 ^ (l31 & (NO * 4 - 1))) << 4) + lhi * 8) = o;
                }
            constexpr int CPR = NO * 4;              // 16-byte chunks per staged row
            constexpr int RPI = 64 / CPR;            // rows per pass of the wave
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = i * RPI + le / CPR, ch = le % CPR;
                h8 v = *reinterpret_cast<const h8*>(ep + row * ROWB + ((ch ^ (row & (CPR - 1))) << 4));
                const long m = m0 + mi * 32 + row;
                if (!GEGLU && Rg) {
                    const h8 rv = *reinterpret_cast<const h8*>(Rg + m * p.ldr + no + ch * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
                }
                *reinterpret_cast<h8*>(Cg + m * p.ldc + no + ch * 8) = v;
            }
        }
    }
}

}  // namespace

bool ds_gemm_w4_applicable(const GemmParams& p) {
    return p.conv == 0 && p.A2 == nullptr && p.rowbias == nullptr && p.dtype == DS_DTYPE_F16 && p.M % 256 == 0 &&
           p.N % 256 == 0 && p.K % 64 == 0 && p.K >= 128 && (p.epi == EPI_NONE || p.epi == EPI_GEGLU);
}

int ds_launch_gemm_w4(const GemmParams& p0, int batch, hipStream_t stream) {
    GemmParams p = p0;
    DS_REQUIRE(ds_gemm_w4_applicable(p), "gemm_w4: shape M=%d N=%d K=%d / epilogue not supported", p.M, p.N, p.K);
    p.tiles_m = p.M / 256;
    p.tiles_n = p.N / 256;
    const size_t lds = W4_EP + 4 * 8192;
    static int cus = 0;
    if (!cus) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        DS_HIP(hipGetDevice(&dev));
        DS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus <= 0) cus = 256;
    }
    const int tiles = p.tiles_m * p.tiles_n;
    int nblk = tiles < cus ? tiles : cus;
    if (tiles > cus) {  // same number of rounds, every round full (see gemm_pp.hip)
        const int rounds = (tiles + cus - 1) / cus;
        nblk = ((tiles + rounds - 1) / rounds + 7) / 8 * 8;
        if (nblk > cus) nblk = cus;
    }
    if (p.epi == EPI_GEGLU) hipLaunchKernelGGL(gemm_w4_kernel<true>, dim3(nblk, 1, batch), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(gemm_w4_kernel<false>, dim3(nblk, 1, batch), dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

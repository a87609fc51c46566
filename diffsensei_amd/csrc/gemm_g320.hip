// fp16 MFMA GEMM with the GEGLU epilogue for the feed-forward projection of a SMALL-BATCH request:
//     C[M, N/2] = h * gelu(g),   [h | g] = LN?(A)[M,K] W[N,K]^T + b,   block tile 256 x 320 x 64, eight waves, ONE block per CU.
// Reference: diffusers' GEGLU (ff.net.0.proj + gating) [3P] inside BasicTransformerBlock, reached from
// src/models/unet.py:244-338; same GemmParams and epilogue arithmetic as gemm.hip's EPI_GEGLU (+ its fused-LayerNorm consumer).
//
// Why a fourth tile shape (round 6; VERDICT r5 item 2: the reference's own call shape - one request, num_samples 1,
// scripts/demo/gradio_wo_mllm.py:45-62).  At UNet batch 2 and 1024 x 1024 the GEGLU projection of the 1280-channel level is
// M = 2048, N = 10240, K = 1280 - the largest GEMM of a batch-1 forward (60 launches, 16 % of it).  Its 256 x 256 grid is 320
// tiles = 1.25 rounds of the 256 CUs (two tile times: 73 us); its 128 x 128 grid (1280 blocks, gemm_glds_kernel<128,false,1>)
// moves 839 MB through L2 -> LDS and sits on that path's ceiling (72.7 us, 738 TFLOP/s).  2048 x 10240 / 256 CUs = 256 x 320
// outputs per CU: with THAT tile every CU gets exactly one block, the fill is (256 + 320) x 1280 x 2 B = 1.47 MB per CU instead
// of 3.3 (31 us at the ~47 GB/s a CU's fill path gives one block, profiles/r06_lds_fill_rate.txt) and the MFMA work of a CU
// (210 MFLOP) is 41 us at the 1.3 PFLOP/s the ping-pong kernel sustains - both run concurrently.
//
// Packing (engine.pack_geglu320): packed W rows [320 t, 320 t + 160) are the hidden rows 160 t .. 160 t + 159, rows
// [320 t + 160, 320 t + 320) their gate rows - a tile holds 160 output columns with both halves.  A 160-column half is five
// 32 x 32 MFMA blocks, so hidden and gate of one output cannot sit in one lane; the pairing goes through the LDS staging tile of
// the epilogue exactly as in gemm.hip: stage 1 rounds h and g to f16 (what nn.Linear hands to GEGLU in the reference), stage 2
// computes f16(h * f16(gelu(g))) on 16-byte row pieces.
//
// Structure.  Eight waves = 4 (M) x 2 (N); a wave owns 64 rows x 160 columns (2 x 5 accumulator blocks of
// v_mfma_f32_32x32x16_f16 = 160 VGPRs, operands swapped like every GEMM here: a lane ends with tile row l & 31 and 4-column
// groups); 7 fragment reads per 10 MFMAs (0.7 KiB of LDS reads per MFMA; the ping-pong kernel: 0.75).  A k-tile is 72 one-KiB
// LDS-DMA pieces (32 of A, 40 of W); wave w moves pieces w + 8 j (four of A, five of W).  Two stages of 72 KiB: the pieces of
// k-tile kt + 1 are issued three at a time between the MFMA groups of k-tile kt (its stage was read during kt - 1, and every
// wave has passed the barrier at the top of kt since), the fragments of k-step kk + 1 are requested ahead of the MFMAs of k-step
// kk (two register sets).  One barrier per k-tile.
// (Measured alternative, same round: 32-wide k-tiles in 64-byte LDS rows - chunk c of row r at slot c ^ ((r >> 2) & 3), conflict-
// free for ds_read_b128 - in a ring of FOUR 36-KiB stages with counted vmcnt waits, two k-tiles in flight beside the two being
// read and the fragment read-ahead carried across the k-tile boundary: bit-identical and SLOWER, 63.4 vs 57.5 us back to back
// (profiles/r06_g320_v2_ring4_k32_microbench.txt against r06_g320_v1_microbench.txt).  The two-stage loop is not waiting for
// its DMA: 20 k-tiles in ~48 us = 2.4 us per 10.5 MFLOP = the matrix pipe ~0.6 busy at the clock this load sustains, the place
// every matrix kernel of this library sits; twice the barriers and twice the DMA pieces per byte cost
// more than the deeper ring hides.  Four schedule variants of THIS loop - s_setprio 1 around the MFMA groups, waves 4..7 issuing
// their pieces one k-step later, the pieces between the halves of an MFMA group, static priority for waves 4..7 - all measured
// 57.1-58.6 us against 57.2-57.5: profiles/r06_g320_schedule_variants.txt.)
// The tile's bias and (-c hi, -c lo) slices are LDS-DMA pieces of the prologue; the consumer's row statistics (the producer's
// partial sums, the format of gemm.hip / gemm_t160.hip) are summed in the prologue under the first k-tile's round trip - four
// floats per lane live across the k-loop.
#include "ds_common.h"
#include "ds_kernels.h"

static thread_local int g_g320 = 0;  // 0 auto, 1 never (A/B: ds_set_option "gemm_g320")
void ds_gemm_set_g320(int v) { g_g320 = v; }

namespace {

constexpr int GM = 256, GN = 320;
constexpr int A_B = GM * 128;              // 32 KiB: A rows 0..255 (32 pieces)
constexpr int STAGE_B = (GM + GN) * 128;   // 72 KiB: then W rows 0..319 (40 pieces)
constexpr int BIAS_OFF = 2 * STAGE_B;      // one 1-KiB piece: f16 bias of the tile's 320 packed columns (640 B used)
constexpr int C_OFF = BIAS_OFF + 1024;     // two pieces: (-c hi, -c lo) f16 pairs of the 320 columns (1280 B used)
constexpr int LDS_B = C_OFF + 2048;        // 147 KiB
constexpr int CS = 656;                    // bytes per row of the epilogue staging tile (320 f16 + 8 pad)
static_assert(128 * CS <= 2 * STAGE_B, "gemm_g320: the 128-row staging tile lies over the two stages");
typedef __attribute__((address_space(3))) void lds_void;

constexpr int NWAVES = 8;
template <int V>
struct IC3 {
    static constexpr int value = V;
};

// GEGLU = false (added at the end of round 6): the same main loop with a PLAIN epilogue (bias or the fused-LayerNorm consumer form,
// 320 output columns per tile, natural weight order) for the one plain projection whose 256 x 320 grid is exactly one block per CU:
// q|k at M = 8192, N = 2560 (UNet batch 8 at 1024 x 1024 = BASELINE configs[2]; batch 2 at 2048 x 2048 = configs[4]), 320 tiles of
// 256 x 256 = 1.25 rounds otherwise (78 us on the 128 x 128 kernel).
// (Measured and NOT kept: a 128 x 320 instantiation with +residual / LayerNorm-producer epilogues for the N = 1280 projections at
// M = 8192 - 64 x 4 = 256 blocks where gemm_pp_kernel runs 160 tiles in one round on 62 % of the CUs.  Bit-identical tiles, and
// slower: K = 5120 121.6 -> 128.8 us, K = 1280 43.1 -> 44.5 us, the consumer form 40.7 -> 43.4 us
// (profiles/r06_g320_128row_tiles_forward_ab_b8.txt): 56 KiB of fill per 5.2 MFLOP k-tile is 1.6x this tile's bytes per flop - the
// block is back on its CU's fill path - while 160 busy CUs clock higher than 256.)
template <bool GEGLU>
__global__ __launch_bounds__(NWAVES * 64, 1) void gemm_g320_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * GM, n0 = tn * GN;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;

    // ---- staging: piece q = wave + 8 j covers LDS rows 8 q .. 8 q + 7 of the stage (A rows first); the DMA destination is
    // lane-linear, so the XOR swizzle goes on the lane's SOURCE chunk.  Rows past M re-read the last row (never stored).
    unsigned off[9];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int q = wave + NWAVES * j;
            if (j < 4) {
                const int row = q * 8 + lrow;
                const int chunk = slot ^ ((row >> 1) & 7);
                const int mr = min(m0 + row, p.M - 1) - m0;
                off[j] = (unsigned)(mr * (int)p.lda + chunk * 8) * 2u;
            } else {
                const int row = (q - 32) * 8 + lrow;
                const int chunk = slot ^ ((row >> 1) & 7);
                off[j] = (unsigned)(row * (int)p.ldw + chunk * 8) * 2u;
            }
        }
    }
    const half_t* const a_tile = p.A + (long)m0 * p.lda;
    const half_t* const w_tile = p.W + (long)n0 * p.ldw;
    auto issue3 = [&](int kt, int buf, auto jc) {   // pieces 3 jc .. 3 jc + 2 of the wave's nine
        constexpr int J0 = decltype(jc)::value * 3;
        const half_t* const a = a_tile + kt * 64;
        const half_t* const w = w_tile + kt * 64;
        const unsigned dst = lds0 + (unsigned)buf * STAGE_B;
#pragma unroll
        for (int j = J0; j < J0 + 3; ++j) {
            const int q = wave + NWAVES * j;
            lds_dma16(j < 4 ? (const void*)a : (const void*)w, off[j], dst + (unsigned)q * 1024u);
        }
    };

    // ---- fragment addresses (bytes into a stage): fragment (block, k-step kk) of a lane sits at (base + block * 4096) ^ (kk << 5)
    const unsigned fsw = (unsigned)((lhi ^ ((l31 >> 1) & 7)) << 4);
    const unsigned fa0 = (unsigned)((wm * 64 + l31) * 128) + fsw;
    const unsigned fb0 = (unsigned)(A_B + (wn * 160 + l31) * 128) + fsw;   // (160 is a multiple of 16: the swizzle phase is l31's)

    f32x16 acc[2][5];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 5; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // ---- prologue: k-tile 0, the tile's bias / c slices (waves 0..2, one piece each; lanes past the slice re-read its last chunk),
    // then the consumer's row statistics - their round trip runs beside the first k-tile's
    typedef IC3<0> I0;
    typedef IC3<1> I1;
    typedef IC3<2> I2;
    issue3(0, 0, I0{});
    issue3(0, 0, I1{});
    issue3(0, 0, I2{});
    const bool ln_in = p.ln_stats != nullptr;
    if (wave == 0 && p.bias) lds_dma16_v(p.bias + n0 + min(lane, 39) * 8, lds0 + BIAS_OFF);
    if (ln_in && (wave == 1 || wave == 2)) lds_dma16_v(p.ln_c + 2 * n0 + min((wave - 1) * 64 + lane, 79) * 8, lds0 + C_OFF + (wave - 1) * 1024);
    float ln_mean[2] = {0.f, 0.f}, ln_rstd[2] = {1.f, 1.f};
    if (ln_in) {
        // (mean, rstd) of the lane's two rows from the producer's partial sums, every load in flight at once; the summation
        // order is the one of gemm.hip's consumer / ln_finalize_kernel (four interleaved chains, (0 + 1) + (2 + 3)): the two
        // half-waves hold the same rows and take two chains each
        const int strips = p.ln_nstrips > 0 ? p.ln_nstrips : (p.K >> 6);
        constexpr int JJ = 6;
        float sa[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, qa[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int base = 0; base < strips; base += 4 * JJ) {
            f32x2 t[2][2][JJ];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = min(m0 + wm * 64 + mi * 32 + l31, p.M - 1);
                const f32x2* part = reinterpret_cast<const f32x2*>(p.ln_stats) + m;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) t[mi][cc][jj] = part[(long)min(base + 2 * lhi + cc + 4 * jj, strips - 1) * p.M];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) {
                        const bool in = base + 2 * lhi + cc + 4 * jj < strips;
                        sa[mi][cc] += in ? t[mi][cc][jj][0] : 0.f;
                        qa[mi][cc] += in ? t[mi][cc][jj][1] : 0.f;
                    }
        }
        const float inv_c = 1.0f / (float)p.K;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const float s2 = sa[mi][0] + sa[mi][1], q2 = qa[mi][0] + qa[mi][1];
            const float s = s2 + __shfl_xor(s2, 32, 64), q = q2 + __shfl_xor(q2, 32, 64);
            ln_mean[mi] = s * inv_c;
            ln_rstd[mi] = rsqrtf(fmaxf(fmaf(-ln_mean[mi], ln_mean[mi], q * inv_c), 0.f) + p.ln_eps);
        }
    }

    const int nk = p.K / 64;
    h8 af[2][2], bf[2][5];
    auto load = [&](int set, const char* st, int kk) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[set][mi] = *reinterpret_cast<const h8*>(st + ((fa0 + mi * 4096) ^ (unsigned)(kk << 5)));
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) bf[set][ni] = *reinterpret_cast<const h8*>(st + ((fb0 + ni * 4096) ^ (unsigned)(kk << 5)));
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int ni = 0; ni < 5; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
    };
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of k-tile kt (issued a k-tile ago) have landed ...
        __builtin_amdgcn_s_barrier();                       // ... and everybody's; every wave has retired its reads of k-tile kt - 1
        asm volatile("" ::: "memory");
        const char* const st = smem + (kt & 1) * STAGE_B;
        const bool more = kt + 1 < nk;
        load(0, st, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) load((kk + 1) & 1, st, kk + 1);
            if (more) {
                if (kk == 0) issue3(kt + 1, (kt + 1) & 1, I0{});
                if (kk == 1) issue3(kt + 1, (kt + 1) & 1, I1{});
                if (kk == 2) issue3(kt + 1, (kt + 1) & 1, I2{});
            }
            __builtin_amdgcn_sched_barrier(0);   // the reads and the DMA pieces go out FIRST
            mfmas(kk & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // the last stage has been read by everyone: the staging tile of the epilogue may overwrite it

    // ---- epilogue in two 128-row passes (the waves with wm >> 1 == pass own its rows).  Stage 1: bias (or the fused-LayerNorm
    // consumer form), round to f16, park [128][320] in LDS.  D layout (operands swapped): register r of a block is tile column
    // (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block's 32.  Stage 2: GEGLU pairing on 16-byte pieces, 320-byte row segments out.
    char* const sC = smem;
    half_t* const Cg = p.C;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        if ((wm >> 1) == pass) {
#pragma unroll
            for (int ni = 0; ni < 5; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * 160 + ni * 32 + 8 * g + 4 * lhi;
                    h4 bq = h4{0, 0, 0, 0};
                    if (p.bias) bq = *reinterpret_cast<const h4*>(smem + BIAS_OFF + nl * 2);
                    h8 cq = h8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (ln_in) cq = *reinterpret_cast<const h8*>(smem + C_OFF + nl * 4);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const int ms = (wm & 1) * 64 + mi * 32 + l31;
                        float v[4];
                        if (ln_in) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float nc = (float)cq[2 * e] + (float)cq[2 * e + 1];   // -c_n
                                v[e] = fmaf(fmaf(ln_mean[mi], nc, acc[mi][ni][4 * g + e]), ln_rstd[mi], (float)bq[e]);
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e] + (float)bq[e];
                        }
                        h4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
                        *reinterpret_cast<h4*>(sC + ms * CS + nl * 2) = o;
                    }
                }
        }
        __syncthreads();
        if constexpr (!GEGLU) {
            // stage 2, plain: thread t takes 16-byte chunk id % 40 of row id / 40, id = t + 512 j: 640-byte row segments
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int id = tid + NWAVES * 64 * j;
                const int row = id / 40, c = id - row * 40;
                const int m = m0 + pass * 128 + row;
                const h8 v = *reinterpret_cast<const h8*>(sC + row * CS + c * 16);
                if (m < p.M) *reinterpret_cast<h8*>(Cg + (long)m * p.ldc + n0 + c * 8) = v;
            }
            continue;
        }
        // stage 2: thread t takes 16-byte output chunk id % 20 of row id / 20, id = t + 512 j
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int id = tid + NWAVES * 64 * j;
            const int row = id / 20, c = id - row * 20;
            const int m = m0 + pass * 128 + row;
            const h8 hv = *reinterpret_cast<const h8*>(sC + row * CS + c * 16);
            const h8 gv = *reinterpret_cast<const h8*>(sC + row * CS + 320 + c * 16);
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 ge = ds_gelu_erf2(f32x2{(float)gv[e], (float)gv[e + 1]});
                o[e] = (half_t)((float)hv[e] * (float)(half_t)ge[0]);
                o[e + 1] = (half_t)((float)hv[e + 1] * (float)(half_t)ge[1]);
            }
            if (m < p.M) *reinterpret_cast<h8*>(Cg + (long)m * p.ldc + (n0 >> 1) + c * 8) = o;
        }
    }
}

}  // namespace

// Shape rule (pure host logic; the launch planner asks it through ds_gemm_g320_fits BEFORE it packs the weights in 320-row
// groups): a GEGLU projection whose 256 x 320 grid is at most one block per CU on at least 5/8 of the CUs, while the
// 256 x 256 grid would need a second, mostly empty round.  256 CUs assumed on the host side (MI355X).
bool ds_gemm_g320_shape(int M, int N, int K, int batch) {
    if (g_g320 == 1 || batch != 1 || M <= 0 || N <= 0 || K < 256 || N % GN != 0 || K % 64 != 0) return false;
    const long b320 = (long)((M + GM - 1) / GM) * (N / GN);
    const long b256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    return b320 <= 256 && b320 >= 160 && b256 > 256;
}

// what the kernel can run at all (GEGLU: the packed layout is the caller's promise, epi == EPI_GEGLU320; plain: epi == EPI_NONE)
bool ds_gemm_g320_possible(const GemmParams& p, int batch) {
    if (p.conv || p.A2 || p.rowbias || p.residual || p.stats_out || (p.epi != EPI_GEGLU320 && p.epi != EPI_NONE) || p.dtype != DS_DTYPE_F16 ||
        p.ln_swapped || batch != 1)
        return false;
    if (p.M <= 0 || p.N % GN != 0 || p.K % 64 != 0 || p.K <= 0) return false;
    if (p.ln_stats && !(p.ln_partial && p.ln_c)) return false;   // finalised statistics are gemm_pp_kernel's consumer form
    if (p.lda * 255 + 64 >= (1L << 30) || p.ldw * 319 + 64 >= (1L << 30)) return false;   // 32-bit lane offsets
    return true;
}

// automatic dispatch of the PLAIN form (ds_launch_gemm's choice, gemm.hip): the shape rule above on a problem the kernel can run
bool ds_gemm_g320_plain_applicable(const GemmParams& p, int batch) {
    return p.epi == EPI_NONE && ds_gemm_g320_possible(p, batch) && ds_gemm_g320_shape(p.M, p.N, p.K, batch);
}

int ds_launch_gemm_g320(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    DS_REQUIRE(ds_gemm_g320_possible(p, 1), "gemm_g320: problem M=%d N=%d K=%d (epilogue %d) not supported", p.M, p.N, p.K, p.epi);
    p.tiles_m = (p.M + GM - 1) / GM;
    p.tiles_n = p.N / GN;
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_g320_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_B));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_g320_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_B));
    }
    if (p.epi == EPI_GEGLU320) hipLaunchKernelGGL(gemm_g320_kernel<true>, dim3(p.tiles_m * p.tiles_n), dim3(NWAVES * 64), (size_t)LDS_B, stream, p);
    else hipLaunchKernelGGL(gemm_g320_kernel<false>, dim3(p.tiles_m * p.tiles_n), dim3(NWAVES * 64), (size_t)LDS_B, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

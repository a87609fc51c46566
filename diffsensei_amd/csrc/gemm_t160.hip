// fp16 MFMA GEMM for the SMALL-BATCH projections of the 1280-channel level: C[M,N] = A[M,K] W[N,K]^T (+bias, +residual, fused-
// LayerNorm producer / consumer), block tile 64 x 160 x 64, five waves, a ring of five LDS stages filled by LDS-DMA.
// Same GemmParams and epilogue semantics as gemm.hip (the reference call sites are listed there: the nn.Linear projections of
// diffusers' BasicTransformerBlock reached from reference src/models/unet.py:244-338 / attention_processor.py:56-84,207-261).
//
// Why a third tile shape (round 6; VERDICT r5 item 2: the reference's own call shape - one request, num_samples 1,
// scripts/demo/gradio_wo_mllm.py:45-62 - ran its dominant kernel at 0.13 of the matrix peak for four rounds).  At UNet batch 2 and
// 1024 x 1024 the level-2 projections are M = 2048, N = 1280, K = 1280 | 5120.  The 64 x 128 ring kernel (gemm_glds_kernel<64,
// false,3>) cuts that into 32 x 10 = 320 blocks for 256 CUs: 64 CUs hold two blocks, 192 hold one, and the launch lasts as long
// as a doubly-loaded CU needs - whose L2 -> LDS fill runs at ~72 GB/s while the other three quarters of the chip idle at half
// that (profiles/r05_weight_prefetch_potential.txt: 20.0 / 52.6 us with hot weights for 0.56 / 2.2 MB of fill per block pair).
// Cross-CU split-K to even that out is priced out on this part (a seam costs 5-13 us, MI355X_MICROARCH.md "splitk-seam";
// measured in round 2: 35 -> 144 us).  A 64 x 160 tile gives 32 x 8 = EXACTLY 256 blocks - one per CU, no reduction, no
// workspace - with 6 % fewer fill bytes per flop; the single block then has to keep the CU's fill path busy by itself, hence
// five stages (four k-tiles = 112 KiB in flight per CU; 141 KiB of LDS).
//
// Structure.  Eight waves.  Waves 0..4 compute: wave w owns the 32-column strip w of the tile and all 64 rows (2 accumulator
// blocks of v_mfma_f32_32x32x16_f16, operands swapped like every GEMM here: a lane ends with tile row l & 31 and 4-column
// groups).  ALL eight waves stage: a k-tile is 28 one-KiB LDS-DMA pieces (8 of A, 20 of W), wave w moves pieces w, w + 8, w + 16
// (and w + 24 for w < 4) - waves 5..7 do nothing else: every SIMD issues the same share of the fill and nobody pads.
// What bounds it (measured, profiles/r06_lds_fill_rate.txt + r06_t160_forward_ab_b2*.txt): a k-tile takes 0.64 us = 28 KiB at
// ~47 GB/s per CU, which is what the L2 -> LDS path of a CU delivers to ONE block whatever is in flight (microbenchmark
// tools/ubench/lds_fill_rate.hip: 45 GB/s per CU from 4 issuing waves, 56 from 8, 61 from 16 on L2-resident operands, the same
// at 4 or 24 KiB in flight per wave; 35-47 out of the Infinity Cache) - the chip-wide ~12 TB/s that the 128 x 128 GEMMs
// (12.2 TB/s) and the 256 x 256 ping-pong kernel (10.3) also sit under.  The first version of this kernel - five waves, six
// pieces each, two of them dummies - measured the same 50.9 / 21.2 us at K = 5120 / 1280 as this one (51.2 / 21.2): the helper
// waves are kept because they cost nothing and remove the padding, not because they pay.  At UNet batch 2 a projection is
// therefore bound by its fill bytes (M N K 2 (1/64 + 1/160) = 147 MB at K = 1280: 12 us) plus ~9 us of launch, first round
// trip and epilogue; a smaller bytes-per-flop needs tiles that no longer give every CU a block.
// One barrier per k-tile, as in the ring kernel.
// Epilogue through LDS (row stride 336 B: conflict-free 8-byte writes), then 16-byte stores of whole 320-byte row segments.
// Fused LayerNorm: a 160-column tile does not hold whole 64-column strips, so the producer emits THREE partial pairs per row and
// tile - columns 0..63, 64..127, 128..159 - i.e. 3 N / 160 entries per row instead of N / 64 (24 instead of 20 at N = 1280; a
// consumer only ever adds the entries up, so their widths need not be equal).  The format is an explicit request of the launch
// planner (GemmParams::stats_strip = 160), and consumers are told the entry count (GemmParams::ln_nstrips); the consumer form
// finalises its own rows from the partial sums like the 128-wide kernels (gemm.hip).  (First version: five 32-column entries
// per tile = 40 per row - the consumers' epilogues then needed twice the load rounds: +0.3 ms per batch-2 forward.)
#include "ds_common.h"
#include "ds_kernels.h"

static thread_local int g_t160 = 0;  // 0 auto, 1 never, 2 never the 128-row tile, 3 128-row tiles wherever the kernel runs (A/B, tests: ds_set_option "gemm_t160")
void ds_gemm_set_t160(int v) { g_t160 = v; }

namespace {

constexpr int TN = 160;
constexpr int CS = 336;                    // bytes per row of the epilogue staging tile (160 f16 + 8 pad)
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swz(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

constexpr int NWAVES = 8, NCOMP = 5;   // waves per block / of them computing (one per 32-column strip)

// TMI = 32-row accumulator blocks per compute wave: 2 -> the 64 x 160 tile (five stages of 28 KiB), 4 -> a 128 x 160 tile (four
// stages of 36 KiB) for the projections whose 64 x 160 grid would be TWO blocks per CU (q|k of a batch-1 request: M = 2048,
// N = 2560 -> 16 x 16 = 256 blocks of 128 x 160; 33.5 us on the 64 x 128 one-buffer kernel, profiles/r06_g320_v1_forward_ab_b2.txt)
template <int STAGES, int TMI>
__global__ __launch_bounds__(NWAVES * 64, 1) void gemm_t160_kernel(const GemmParams p) {
    constexpr int TM = 32 * TMI;
    constexpr int STAGE_B = (TM + TN) * 128;   // A rows 0..TM-1 (TM / 8 pieces), then W rows 0..159 (20 pieces)
    constexpr int NPIECE = TM / 8 + 20;        // 28 | 36 pieces per k-tile
    constexpr int PW = (NPIECE + NWAVES - 1) / NWAVES;   // 4 | 5 per wave for waves 0..3, one fewer for waves 4..7
    static_assert(NPIECE % NWAVES == 4, "waves 0..3 move PW pieces per k-tile, waves 4..7 PW - 1");
    constexpr int RED_OFF = ((TM * CS + 1023) / 1024 + 2) * 1024;   // LayerNorm partials of the tile's five 32-column pieces, behind the staging tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    const bool compute = wave < NCOMP;   // wave-uniform

    // ---- staging: piece q = wave + 8 j covers LDS rows 8 q .. 8 q + 7 of the stage (A rows first); the DMA destination is
    // lane-linear, so the XOR swizzle goes on the lane's SOURCE chunk.  Rows past M re-read the last row (never stored).
    unsigned off[PW];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int q = wave + NWAVES * j;
            if (q < TM / 8) {
                const int row = q * 8 + lrow;
                const int chunk = slot ^ ((row >> 1) & 7);
                const int mr = min(m0 + row, p.M - 1) - m0;
                off[j] = (unsigned)(mr * (int)p.lda + chunk * 8) * 2u;
            } else {
                const int row = (min(q, NPIECE - 1) - TM / 8) * 8 + lrow;
                const int chunk = slot ^ ((row >> 1) & 7);
                off[j] = (unsigned)(row * (int)p.ldw + chunk * 8) * 2u;
            }
        }
    }
    const half_t* const a_tile = p.A + (long)m0 * p.lda;
    const half_t* const w_tile = p.W + (long)n0 * p.ldw;
    auto issue = [&](int kt, int buf) {
        const half_t* const a = a_tile + kt * 64;
        const half_t* const w = w_tile + kt * 64;
        const unsigned dst = lds0 + (unsigned)buf * STAGE_B;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int q = wave + NWAVES * j;
            if (q < NPIECE) lds_dma16(q < TM / 8 ? (const void*)a : (const void*)w, off[j], dst + (unsigned)q * 1024u);   // (wave-uniform)
        }
    };

    // ---- fragment addresses (bytes into a stage)
    unsigned fa[4], fb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        fa[kk] = l31 * 128 + swz(l31, kk * 2 + lhi);
        const int rb = wave * 32 + l31;
        fb[kk] = TM * 128 + rb * 128 + swz(rb, kk * 2 + lhi);
    }

    f32x16 acc[TMI];
#pragma unroll
    for (int mi = 0; mi < TMI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    // ---- the epilogue's operands are REQUESTED HERE, ahead of the first DMA piece (round 6): bias, the residual rows of the
    // thread's stage-2 pieces, and - consumer form - the first 24 partial-sum entries of the lane's rows and its c values.  With
    // one block per CU nothing else hides them: requested in the epilogue they were two memory round trips back to back (bias /
    // partial sums, then - behind a barrier - the residual) of a launch that lasts 21 us.  They are OLDER than every DMA piece, and
    // vmcnt retires in issue order, so the counted waits of the k-loop ("all but the newest pieces") are unchanged; C == residual
    // (the in-place h += ...) is safe: the block reads exactly the tile it writes later.
    const int nw = n0 + wave * 32;                       // stage 1: the wave's 32-column strip (compute waves)
    const int c20 = tid % 20, r16 = tid / 20;            // stage 2: thread t takes 16-byte chunk t % 20 of rows t / 20 + 16 j
    const int n2 = n0 + c20 * 8;
    const bool ln_in = p.ln_stats != nullptr;
    const int strips = p.ln_nstrips > 0 ? p.ln_nstrips : (p.K >> 6);
    constexpr int JJ = 6;   // 24 entries (a gemm_t160_kernel producer at N = 1280) in one round of loads
    constexpr bool PRE = TMI == 2;   // the 64-row tile keeps the first 24 partial-sum entries of its rows in registers across the k-loop;
                                     // the 128-row tile (twice the rows per lane) finalises its statistics in the prologue instead
    h4 bq[4];
    h8 rv[TM / 16], cq[4];
    f32x2 t0[PRE ? 2 : 1][2][JJ];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = h4{0, 0, 0, 0};
    if (compute) {
        if (p.bias) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const h4*>(p.bias + nw + 8 * g + 4 * lhi);
        }
        if (p.residual) {   // (tid < 320 == the compute waves)
#pragma unroll
            for (int j = 0; j < TM / 16; ++j) {
                const int m = min(m0 + r16 + 16 * j, p.M - 1);
                rv[j] = *reinterpret_cast<const h8*>(p.residual + (long)m * p.ldr + n2);
            }
        }
        if (ln_in) {
            if constexpr (PRE) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int m = min(m0 + mi * 32 + l31, p.M - 1);
                    const f32x2* part = reinterpret_cast<const f32x2*>(p.ln_stats) + m;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                        for (int jj = 0; jj < JJ; ++jj) t0[mi][cc][jj] = part[(long)min(2 * lhi + cc + 4 * jj, strips - 1) * p.M];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) cq[g] = *reinterpret_cast<const h8*>(p.ln_c + 2 * (nw + 8 * g + 4 * lhi));
        }
    }
    asm volatile("" ::: "memory");   // (the requests stay in front of the DMA prologue)

    const int nk = p.K / 64;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s, s);
    // consumer: (mean, rstd) of the lane's rows from the producer's partial sums, every load in flight at once; the summation
    // order is the one of gemm.hip's consumer / ln_finalize_kernel (four interleaved chains, (0 + 1) + (2 + 3)): the two
    // half-waves hold the same rows and take two chains each
    float ln_mean[TMI], ln_rstd[TMI];
#pragma unroll
    for (int mi = 0; mi < TMI; ++mi) ln_mean[mi] = 0.f, ln_rstd[mi] = 1.f;
    auto ln_rows = [&]() {
#pragma unroll
        for (int m2 = 0; m2 < TMI; m2 += 2) {   // two rows per lane and round of loads
            float sa[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, qa[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            for (int base = 0; base < strips; base += 4 * JJ) {
                f32x2 t[2][2][JJ];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int m = min(m0 + (m2 + mi) * 32 + l31, p.M - 1);
                    const f32x2* part = reinterpret_cast<const f32x2*>(p.ln_stats) + m;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                        for (int jj = 0; jj < JJ; ++jj) {
                            const int j = base + 2 * lhi + cc + 4 * jj;
                            if (PRE && base == 0) t[mi][cc][jj] = t0[PRE ? mi : 0][cc][jj];
                            else t[mi][cc][jj] = part[(long)min(j, strips - 1) * p.M];
                        }
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
                ln_mean[m2 + mi] = s * inv_c;
                ln_rstd[m2 + mi] = rsqrtf(fmaxf(fmaf(-ln_mean[m2 + mi], ln_mean[m2 + mi], q * inv_c), 0.f) + p.ln_eps);
            }
        }
    };
    if constexpr (!PRE) {
        if (compute && ln_in) ln_rows();   // (under the prologue's round trip; eight floats live across the k-loop)
    }
    int buf = 0, fill = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // k-tile kt has landed once all but the newer k-tiles' pieces (PW per k-tile for waves 0..3, PW - 1 for waves 4..7) are done
        const int newer = min(STAGES - 2, nk - 1 - kt);
        static_assert(STAGES >= 3 && STAGES <= 5, "the counted waits below cover up to three newer k-tiles");
        if (wave < 4) {
            if (newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PW) : "memory");
            else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
            else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (PW - 1)) : "memory");
            else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PW - 1)) : "memory");
            else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave has retired its reads of k-tile kt - 1
        asm volatile("" ::: "memory");
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, fill);
        const char* const st = smem + buf * STAGE_B;
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        fill = fill + 1 == STAGES ? 0 : fill + 1;
        if (!compute) continue;   // waves 5..7 only stage
        h8 af[4][TMI], bf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf[kk] = *reinterpret_cast<const h8*>(st + fb[kk]);
#pragma unroll
            for (int mi = 0; mi < TMI; ++mi) af[kk][mi] = *reinterpret_cast<const h8*>(st + fa[kk] + mi * 32 * 128);
        }
        // all twelve (twenty) fragment reads are in flight before the first MFMA (left alone the scheduler, saving registers this kernel
        // has plenty of, put every read right in front of its MFMA: eight exposed LDS round trips per k-tile); the compiler's
        // counted lgkmcnt waits then release the MFMAs in read order
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < TMI; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk], af[kk][mi], acc[mi], 0, 0, 0);
    }
    __syncthreads();   // the last stage has been read by everyone: the staging tile of the epilogue may overwrite it

    // ---- epilogue, stage 1: bias (or the fused-LayerNorm consumer form), round to f16, park the tile in LDS as [m][n].
    // D layout (operands swapped): register r of a block is tile column (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the wave's strip.
    char* const sC = smem;
    if (compute) {
    if constexpr (PRE) {
        if (ln_in) ln_rows();   // (the first 24 entries - all of them at K = 1280 - have been in registers since the top of the kernel)
    }
#pragma unroll
    for (int mi = 0; mi < TMI; ++mi) {
        const int ml = mi * 32 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
            if (ln_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float nc = (float)cq[g][2 * e] + (float)cq[g][2 * e + 1];   // -c_n
                    v[e] = fmaf(fmaf(ln_mean[mi], nc, acc[mi][4 * g + e]), ln_rstd[mi], (float)bq[g][e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mi][4 * g + e] + (float)bq[g][e];
            }
            h4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<h4*>(sC + ml * CS + (wave * 32 + 8 * g + 4 * lhi) * 2) = o;
        }
    }
    }   // compute waves
    __syncthreads();
    // ---- stage 2: thread t takes 16-byte chunk t % 20 of rows t / 20 + 16 j: 320-byte row segments per 20 lanes
    const int c = c20, n = n2;
    if (tid < NCOMP * 64) {
#pragma unroll
    for (int j = 0; j < TM / 16; ++j) {
        const int row = r16 + 16 * j, m = m0 + row;
        h8 v = *reinterpret_cast<const h8*>(sC + row * CS + c * 16);
        if (p.residual) v = v + rv[j];   // v_pk_add_f16: the same number as (f16)((float)a + (float)b) (tests/test_f16_add_equivalence.py)
        if (m < p.M) *reinterpret_cast<h8*>(p.C + (long)m * p.ldc + n) = v;
        if (p.stats_out) {
            // producer: (sum, sum of squares) of the 32-column piece c >> 2 of this row = the four lanes of a quad (20 lanes per
            // row: quads are the largest lane groups that never straddle a row or a wave) -> LDS behind the staging tile
            float s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s1 += f;
                q1 = fmaf(f, f, q1);
            }
            s1 += ds_dpp_f32<0xB1>(s1);
            q1 += ds_dpp_f32<0xB1>(q1);
            s1 += ds_dpp_f32<0x4E>(s1);
            q1 += ds_dpp_f32<0x4E>(q1);
            if ((c & 3) == 0) *reinterpret_cast<f32x2*>(smem + RED_OFF + (row * 5 + (c >> 2)) * 8) = f32x2{s1, q1};
        }
    }
    }   // threads 0..319
    if (p.stats_out) {
        // three entries per row and tile: pieces (0 + 1), (2 + 3), 4; threads e TM .. e TM + TM - 1 write entry e of the TM rows
        __syncthreads();
        if (tid < 3 * TM) {
            const int e = tid / TM, row = tid - e * TM, m = m0 + row;
            const f32x2* const r5 = reinterpret_cast<const f32x2*>(smem + RED_OFF) + row * 5;
            f32x2 o2 = r5[2 * e];
            if (e < 2) o2 = o2 + r5[2 * e + 1];
            if (m < p.M) *reinterpret_cast<f32x2*>(p.stats_out + 2 * ((long)(tn * 3 + e) * p.M + m)) = o2;
        }
    }
}

}  // namespace

// Shape rule (pure host logic; the launch planner asks it through ds_gemm_t160_shape before it requests 32-column statistics):
// a plain f16 GEMM whose 64 x 160 grid covers between 5/8 and all of the CUs with one block each, while the 64 x 128 grid
// would leave some CUs with two blocks and the rest with one.
// Rows per tile the rule picks: 64, or 128 where the 64-row grid would be more than one block per CU but the 128-row grid is
// again one per CU on at least 5/8 of them (q|k of a batch-1 request: M = 2048, N = 2560 -> 16 x 16 blocks); 0: not this kernel.
static int t160_rows(int M, int N, int K, int batch) {
    if (g_t160 == 1 || batch != 1 || M <= 0 || N <= 0 || K < 256 || N % TN != 0 || K % 64 != 0) return 0;
    const long b64 = (long)((M + 63) / 64) * (N / TN);
    const long b128 = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (b64 <= 256 && b64 >= 160 && b128 > 256) return 64;
    const long b2 = (long)((M + 127) / 128) * (N / TN);
    if (g_t160 != 2 && b64 > 256 && b2 <= 256 && b2 >= 160) return 128;
    return 0;
}
bool ds_gemm_t160_shape(int M, int N, int K, int batch) { return t160_rows(M, N, K, batch) != 0; }
int ds_gemm_t160_rows(int M, int N, int K, int batch) { return g_t160 == 3 ? 128 : t160_rows(M, N, K, batch); }

// what the kernel can run at all (ds_set_option "gemm_variant" 11 forces it on every such problem: parity tests, A/B)
bool ds_gemm_t160_possible(const GemmParams& p, int batch) {
    if (p.conv || p.A2 || p.rowbias || p.epi != EPI_NONE || p.dtype != DS_DTYPE_F16 || p.ln_swapped || batch != 1) return false;
    if (p.M <= 0 || p.N % TN != 0 || p.K % 64 != 0 || p.K <= 0) return false;
    if (p.ln_stats && !p.ln_partial) return false;            // finalised statistics are gemm_pp_kernel's consumer form
    if (p.stats_out && p.stats_strip != 160) return false;    // 64-column statistics need whole 64-column strips per tile
    if (p.lda * 127 + 64 >= (1L << 30) || p.ldw * 159 + 64 >= (1L << 30)) return false;   // 32-bit lane offsets
    return true;
}

bool ds_gemm_t160_applicable(const GemmParams& p, int batch) {
    return ds_gemm_t160_possible(p, batch) && ds_gemm_t160_shape(p.M, p.N, p.K, batch);
}

int ds_launch_gemm_t160(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    DS_REQUIRE(p.N % TN == 0 && p.K % 64 == 0 && !p.A2 && !p.rowbias && p.epi == EPI_NONE,
               "gemm_t160: shape M=%d N=%d K=%d not supported", p.M, p.N, p.K);
    DS_REQUIRE(!p.stats_out || p.stats_strip == 160, "gemm_t160: emits its own statistics format only (stats_strip = %d, expected 160)", p.stats_strip);
    DS_REQUIRE(!p.ln_stats || (p.ln_partial && p.ln_c && !p.ln_swapped), "gemm_t160: consumes partial LayerNorm sums in the row form only");
    // 128-row tiles where the rule says so (or gemm_t160 = 3: wherever the kernel runs - parity tests), else 64-row tiles
    const bool tall = g_t160 == 3 || t160_rows(p.M, p.N, p.K, 1) == 128;
    const int tm_rows = tall ? 128 : 64;
    p.tiles_m = (p.M + tm_rows - 1) / tm_rows;
    p.tiles_n = p.N / TN;
    const size_t lds = tall ? (size_t)4 * (128 + TN) * 128 : (size_t)5 * (64 + TN) * 128;   // 144 | 140 KiB
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_t160_kernel<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * (64 + TN) * 128));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_t160_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (128 + TN) * 128));
    }
    if (tall) hipLaunchKernelGGL((gemm_t160_kernel<4, 4>), dim3(p.tiles_m * p.tiles_n), dim3(NWAVES * 64), lds, stream, p);
    else hipLaunchKernelGGL((gemm_t160_kernel<5, 2>), dim3(p.tiles_m * p.tiles_n), dim3(NWAVES * 64), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

// fp16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+bias, +row-group bias, +residual, GEGLU/GELU) and the
// implicit-GEMM 3x3 convolution over NHWC activations that shares its main loop.
//
// Replaces, on the DiffSensei UNet hot path (reference src/models/unet.py:206,244-338 -> diffusers blocks):
//   * every nn.Linear (to_q/to_k/to_v/to_out, proj_in/proj_out, GEGLU ff.net.0.proj, ff.net.2)
//     reference src/models/attention_processor.py:56,63,64,84,207,225,226,245,246,261
//   * every 3x3 conv of ResnetBlock2D / Downsample2D / Upsample2D and the 1x1 conv_shortcut
//
// CDNA4 design: 256 threads = 4 waves (2x2), block tile BM x 128 x 64, v_mfma_f32_32x32x16_f16 with the
// operands swapped (D[n][m]) so each lane ends with 4 consecutive output columns of one row.  LDS tiles are
// [rows][64 k] with the eight 16-byte chunks of a row XOR-swizzled (chunk ^= (row>>1)&7) so every
// ds_read_b128 lane group hits 16 distinct slots.  Two staging pipelines:
//   * gemm_glds_kernel   (K % 64 == 0, the hot path): global_load_lds_dwordx4 straight into LDS — no VGPR round
//     trip, no ds_write pass; the swizzle is applied to each lane's SOURCE address (the DMA destination is
//     lane-linear); two LDS buffers, one barrier per k-tile, tile t+1 in flight under the MFMAs of tile t.
//     Convolution halo / ragged rows are served from a zero page / clamped rows.
//   * gemm_f16_kernel    (any K % 8 == 0): global -> VGPR -> LDS, predicated loads; generic fallback.
// The epilogue is staged through LDS so global stores are full 16 B/lane rows.  Convolution gathers its A tile
// straight from the NHWC tensor (zero halo, optional stride 2, optional fused nearest x2 upsample) — no im2col
// buffer ever exists in HBM.
#include "ds_common.h"
#include "ds_kernels.h"

static thread_local int g_gemm_variant = 0;  // 0 auto; A/B: 1 register staging, 2 two-buffer glds, 3 ping-pong 256x256, 7 two-buffer BM 64, 8/9 one-buffer glds (BM 128/64), 10 halo conv, 11 64x160 tiles (gemm_t160.hip)
void ds_gemm_set_variant(int v) { g_gemm_variant = v; }
static thread_local int g_gemm_ring = 0;  // 0 auto, 1 never use the ring-buffered small-grid kernel (A/B)
void ds_gemm_set_ring(int v) { g_gemm_ring = v; }
static thread_local int g_pp_narrow = 0;  // 0 auto, 1: the N, K <= 640 projections never run gemm_pp_kernel (the rule until round 6; A/B)
void ds_gemm_set_pp_narrow(int v) { g_pp_narrow = v; }
static thread_local int g_gemm_debug = 0;  // ablation switches, see GemmParams::debug
void ds_gemm_set_debug(int v) { g_gemm_debug = v; }

namespace {

constexpr int BN = 128;
constexpr int BK = 64;
constexpr int CS_STRIDE = 272;  // bytes per row of the epilogue staging tile (128 f16 + 8 pad)

__device__ __attribute__((aligned(256))) char g_zero_page[256];  // zero-initialised: conv halo source

__device__ __forceinline__ int swz(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------- MFMA over one k-tile
template <int WR, int WC = 64>  // rows / columns of the block tile owned by one wave (32 * MI, 32 * NI)
__device__ __forceinline__ void mma_tile(const char* cA, const char* cB, f32x16 (&acc)[WR / 32][WC / 32], int wm,
                                         int wn, int l31, int lhi) {
    constexpr int MI = WR / 32, NI = WC / 32;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        h8 af[MI], bf[NI];
        const int ch = kk * 2 + lhi;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int r = wm * WR + mi * 32 + l31;
            af[mi] = *reinterpret_cast<const h8*>(cA + r * 128 + swz(r, ch));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int r = wn * WC + ni * 32 + l31;
            bf[ni] = *reinterpret_cast<const h8*>(cB + r * 128 + swz(r, ch));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------- epilogue
// stage 1: bias / row-group bias, round to f16, park the tile in LDS as [m][n] (BM > 128 goes in 128-row halves);
// stage 2: coalesced 16-byte rows out of LDS (+ activation, + residual, or GEGLU pairing).
// D layout (operands swapped): lane holds row m_local = ..+(lane&31); regs r -> n = (r&3)+8*(r>>2)+4*(lane>>5)
//
// Fused LayerNorm (LNF; the plain GEMMs of the LDS-DMA kernels; gemm_pp.hip's header has the algebra): the small-batch and
// 640-channel counterpart of gemm_pp_kernel's producer / consumer pair, in the SAME statistics format, so producers and
// consumers of either family combine.
//   producer (p.stats_out): stage 2 holds the final f16 values (residual added) as 8 columns per thread, 16 threads per row:
//     (sum, sum of squares) per row and 64-column strip - a 3-step butterfly over the 8 lanes of a strip - go to
//     stats_out[(n / 64) * M + m] (float2).  Needs N % 128 == 0 (every strip complete).
//   consumer (p.ln_stats with p.ln_partial): the lane owns tile ROW m in stage 1, so it sums the row's K / 64 partials itself
//     (the order of ln_finalize_kernel: four interleaved chains, then (0 + 1) + (2 + 3) - the same bits as the finalize
//     launch the 256 x 256 consumers use) and applies y = rstd (acc - mean c_n) + b'_n in fp32 where the bias add was:
//     no finalize launch, no LayerNorm launch.  c_n comes as the negated f16 (hi, lo) pair the 256 x 256 kernel uses.
template <int BM, int WR, int NT, bool LNF = false>  // block rows, rows per wave, threads per block, fused-LayerNorm code compiled in
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x16 (&acc)[WR / 32][2], char* smem, int m0, int n0,
                                         long bz, int wm, int wn, int l31, int lhi, int tid) {
    constexpr int MI = WR / 32;
    constexpr int HALVES = BM > 128 ? BM / 128 : 1;
    constexpr int HR = BM / HALVES;  // rows staged per pass
    char* sC = smem;
    half_t* Cg = p.C + bz * p.sC;
    const half_t* Rg = p.residual ? p.residual + bz * p.sR : nullptr;
    // the lane's bias values, fetched once (clamped: masked columns are never stored) instead of behind a branch - and
    // the s_waitcnt vmcnt(0) that follows a conditional load - per 4 values
    h4 bq[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[ni][g] = h4{0, 0, 0, 0};
    if (p.bias) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bq[ni][g] = *reinterpret_cast<const h4*>(p.bias + min(n0 + wn * 64 + ni * 32 + 8 * g + 4 * lhi, p.N - 4));
    }
    const bool ln_any = LNF && p.ln_stats != nullptr;
    const bool ln_in = ln_any && !p.ln_swapped;    // row form: the normalised rows are the tile rows (one per lane)
    const bool ln_col = ln_any && p.ln_swapped;    // operand-swapped form (V^T = Wv X_b^T): they are the tile COLUMNS
    // consumer: (mean, rstd) of the lane's rows from the producer's partial sums.  Every load of every row is in flight at
    // once - ONE round trip to L2 (a chain of loads per row, row after row, cost 15 us per launch: ten round trips in the
    // epilogue of every block).  The two half-waves hold the same rows: lanes 0..31 sum chains 0 and 1 of ln_finalize_kernel's
    // four interleaved chains, lanes 32..63 chains 2 and 3, and one cross-half exchange gives (0 + 1) + (2 + 3) - the very
    // bits of the finalize launch (fp32 addition commutes).
    float ln_mean[MI], ln_rstd[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ln_mean[mi] = 0.f, ln_rstd[mi] = 1.f;
    if constexpr (LNF) {
        if (ln_in) {
            const int strips = p.ln_nstrips > 0 ? p.ln_nstrips : (p.K >> 6);
            float sa[MI][2], qa[MI][2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) sa[mi][0] = sa[mi][1] = qa[mi][0] = qa[mi][1] = 0.f;
            constexpr int JJ = MI == 1 ? 6 : 3;   // strips per chain and round: 24 (K = 1280 in one round, 20 or 24 entries) or, with two rows per lane, 12
            for (int base = 0; base < strips; base += 4 * JJ) {
                f32x2 t[MI][2][JJ];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int m = min(m0 + wm * WR + mi * 32 + l31, p.M - 1);
                    const f32x2* part = reinterpret_cast<const f32x2*>(p.ln_stats) + m;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                        for (int jj = 0; jj < JJ; ++jj) {
                            const int j = base + 2 * lhi + cc + 4 * jj;
                            t[mi][cc][jj] = part[(long)min(j, strips - 1) * p.M];
                        }
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
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
            for (int mi = 0; mi < MI; ++mi) {
                const float s2 = sa[mi][0] + sa[mi][1], q2 = qa[mi][0] + qa[mi][1];
                const float s = s2 + __shfl_xor(s2, 32, 64), q = q2 + __shfl_xor(q2, 32, 64);
                ln_mean[mi] = s * inv_c;
                ln_rstd[mi] = rsqrtf(fmaxf(fmaf(-ln_mean[mi], ln_mean[mi], q * inv_c), 0.f) + p.ln_eps);
            }
        }
    }
    // operand-swapped form: the statistics of the block's 128 columns = rows bz * ln_bstride + n0 .. of the normalised matrix are
    // finalised once by threads 0..127 (one L2 round trip per 24 entries) into a table behind the staging tile
    f32x2* const ln_tab = reinterpret_cast<f32x2*>(smem + HR * CS_STRIDE);
    if constexpr (LNF) {
        if (ln_col) {
            if (tid < BN) {
                const int strips = p.ln_nstrips > 0 ? p.ln_nstrips : (p.K >> 6);
                const long row = min(bz * p.ln_bstride + n0 + tid, p.ln_rows - 1);
                const f32x2* part = reinterpret_cast<const f32x2*>(p.ln_stats) + row;
                float sa[4] = {0.f, 0.f, 0.f, 0.f}, qa[4] = {0.f, 0.f, 0.f, 0.f};
                for (int base = 0; base < strips; base += 24) {
                    f32x2 t[4][6];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int jj = 0; jj < 6; ++jj) t[q][jj] = part[(long)min(base + q + 4 * jj, strips - 1) * p.ln_rows];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int jj = 0; jj < 6; ++jj) {
                            const bool in = base + q + 4 * jj < strips;
                            sa[q] += in ? t[q][jj][0] : 0.f;
                            qa[q] += in ? t[q][jj][1] : 0.f;
                        }
                }
                const float inv_c = 1.0f / (float)p.K;
                const float mean = ((sa[0] + sa[1]) + (sa[2] + sa[3])) * inv_c;
                const float var = fmaxf(fmaf(-mean, mean, ((qa[0] + qa[1]) + (qa[2] + qa[3])) * inv_c), 0.f);
                ln_tab[tid] = f32x2{mean, rsqrtf(var + p.ln_eps)};
            }
            __syncthreads();
        }
    }
    h8 cq[2][4];  // fused LayerNorm, consumer: (-c hi, -c lo) of the lane's 4 columns per (ni, g)
    if constexpr (LNF) {
        if (ln_in) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    cq[ni][g] = *reinterpret_cast<const h8*>(p.ln_c + 2 * min(n0 + wn * 64 + ni * 32 + 8 * g + 4 * lhi, p.N - 4));
        }
    }
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ml = wm * WR + mi * 32 + l31;  // row within the block tile
            if (ml / HR != half) continue;                   // wave-uniform: (wm, mi) decide the half
            const int ms = ml - half * HR;
            const int m = m0 + ml;
            const int grp = p.rowbias ? ((m < p.M ? m : p.M - 1) / p.rows_per_group) : 0;
            const float mean = ln_mean[mi], rstd = ln_rstd[mi];
            float nc_row = 0.f, bp_row = 0.f;   // operand-swapped form: -c and b' of the lane's ROW
            if constexpr (LNF) {
                if (ln_col) {
                    const h4 cb = *reinterpret_cast<const h4*>(p.ln_c + 4 * (long)(m < p.M ? m : p.M - 1));
                    nc_row = (float)cb[0] + (float)cb[1];
                    bp_row = (float)cb[2] + (float)cb[3];
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * 64 + ni * 32 + 8 * g + 4 * lhi;
                    const int n = n0 + nl;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e] + (float)bq[ni][g][e];
                    if constexpr (LNF) {
                        if (ln_in) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float nc = (float)cq[ni][g][2 * e] + (float)cq[ni][g][2 * e + 1];   // -c_n
                                v[e] = fmaf(fmaf(mean, nc, acc[mi][ni][4 * g + e]), rstd, (float)bq[ni][g][e]);
                            }
                        }
                        if (ln_col) {
                            const f32x4 t0 = *reinterpret_cast<const f32x4*>(ln_tab + nl), t1 = *reinterpret_cast<const f32x4*>(ln_tab + nl + 2);
                            v[0] = fmaf(fmaf(t0[0], nc_row, acc[mi][ni][4 * g + 0]), t0[1], bp_row);
                            v[1] = fmaf(fmaf(t0[2], nc_row, acc[mi][ni][4 * g + 1]), t0[3], bp_row);
                            v[2] = fmaf(fmaf(t1[0], nc_row, acc[mi][ni][4 * g + 2]), t1[1], bp_row);
                            v[3] = fmaf(fmaf(t1[2], nc_row, acc[mi][ni][4 * g + 3]), t1[3], bp_row);
                        }
                    }
                    if (n < p.N) {
                        if (p.rowbias) {
                            const h4 bv = *reinterpret_cast<const h4*>(p.rowbias + (long)grp * p.rowbias_ld + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
                        }
                    }
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<h4*>(sC + ms * CS_STRIDE + nl * 2) = o;
                }
            }
        }
        __syncthreads();
        const int mh = m0 + half * HR;
        if (p.epi == EPI_GEGLU) {
            // packed weight rows: each 128-row tile = 64 "hidden" columns followed by their 64 "gate" columns
            constexpr int IT = HR * 8 / NT;
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int id = tid + NT * j;
                const int row = id >> 3, c = id & 7;
                const int m = mh + row, n = (n0 >> 1) + c * 8;
                if (m < p.M && n < (p.N >> 1)) {
                    const h8 hv = *reinterpret_cast<const h8*>(sC + row * CS_STRIDE + c * 16);
                    const h8 gv = *reinterpret_cast<const h8*>(sC + row * CS_STRIDE + 128 + c * 16);
                    h8 o;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 ge = ds_gelu_erf2(f32x2{(float)gv[e], (float)gv[e + 1]});
                        o[e] = (half_t)((float)hv[e] * (float)(half_t)ge[0]);
                        o[e + 1] = (half_t)((float)hv[e + 1] * (float)(half_t)ge[1]);
                    }
                    *reinterpret_cast<h8*>(Cg + (long)m * p.ldc + n) = o;
                }
            }
        } else {
            constexpr int IT = HR * 16 / NT;
            // the residual pieces of the whole pass are requested up front (clamped addresses; the accumulators are
            // dead by now, so the registers are there) - not one load / wait / store round trip per 16 bytes
            h8 rvs[IT];
            if (Rg) {
#pragma unroll
                for (int j = 0; j < IT; ++j) {
                    const int id = tid + NT * j;
                    const int m = min(mh + (id >> 4), p.M - 1), n = min(n0 + (id & 15) * 8, p.N - 8);
                    rvs[j] = *reinterpret_cast<const h8*>(Rg + (long)m * p.ldr + n);
                }
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int id = tid + NT * j;
                const int row = id >> 4, c = id & 15;
                const int m = mh + row, n = n0 + c * 8;
                float sv = 0.f, qv = 0.f;
                if (m < p.M && n < p.N) {
                    h8 v = *reinterpret_cast<const h8*>(sC + row * CS_STRIDE + c * 16);
                    if (p.epi == EPI_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 ge = ds_gelu_erf2(f32x2{(float)v[e], (float)v[e + 1]});
                            v[e] = (half_t)ge[0];
                            v[e + 1] = (half_t)ge[1];
                        }
                    } else if (p.epi == EPI_QUICK_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            v[e] = (half_t)(f / (1.0f + __expf(-1.702f * f)));
                        }
                    }
                    if (Rg) {
                        const h8 rv = rvs[j];
                        v = v + rv;   // v_pk_add_f16: the same number as (f16)((float)a + (float)b) (tests/test_f16_add_equivalence.py)
                    }
                    *reinterpret_cast<h8*>(Cg + (long)m * p.ldc + n) = v;
                    if constexpr (LNF) {
                        if (p.stats_out) {
                            float s1 = 0.f, q1 = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float f = (float)v[e];
                                s1 += f;
                                q1 = fmaf(f, f, q1);
                            }
                            sv = s1, qv = q1;
                        }
                    }
                }
                if constexpr (LNF) {
                    // the 8 lanes c & 7 of a row's 64-column strip: every lane of the wave takes part (rows past M carry zeros)
                    if (p.stats_out) {
                        sv += ds_dpp_f32<0xB1>(sv);   // lane ^ 1, lane ^ 2, then lane j <-> 7 - j of every 8
                        qv += ds_dpp_f32<0xB1>(qv);
                        sv += ds_dpp_f32<0x4E>(sv);
                        qv += ds_dpp_f32<0x4E>(qv);
                        sv += ds_dpp_f32<0x141>(sv);
                        qv += ds_dpp_f32<0x141>(qv);
                        if ((c & 7) == 0 && m < p.M && n < p.N) {
                            f32x2 o2 = {sv, qv};
                            *reinterpret_cast<f32x2*>(p.stats_out + 2 * ((long)(n >> 6) * p.M + m)) = o2;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- glds pipeline
// STAGES = 1: one 32-KiB LDS buffer, three (four at BM 64) blocks per CU - independent blocks hide each other's DMA latency;
// STAGES = 2: two buffers, tile t+1 / t+2 in flight; STAGES >= 3: a ring of buffers with STAGES-1 tiles in flight and ONE
// barrier per k-tile - for grids too small to give a CU several blocks (num_samples 1: M = 2048 rows), where the
// one-buffer kernel spends every k-tile waiting out the full L2/HBM latency.  Equal to the one-buffer kernel when the
// operands are MALL-hot (profiles/r02_small_batch_gemm_ab.txt), 6 % faster per launch inside the sampler where every weight
// is read cold from HBM (31.2 vs 33.3 us, +4.8 % panels/s at num_samples 1: profiles/r02_ring_in_pipeline_ab.txt).
template <int BM, bool CONV, int STAGES = 2>
__global__ __launch_bounds__(256, (BM > 128 ? 1 : (STAGES >= 4 ? 1 : (STAGES == 1 ? (BM == 64 ? 4 : 3) : 2)))) void gemm_glds_kernel(const GemmParams p) {
    constexpr int MI = BM / 64;
    constexpr int ASEG = BM / 32;  // 1-KiB (8-row) A segments per wave per k-tile
    constexpr int BSEG = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + STAGES * BM * 128;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const long bz = blockIdx.z;
    const int lrow = lane >> 3, slot = lane & 7;

    // per-lane source descriptors: lane writes LDS slot `slot` of row `row`, so it must fetch the chunk that the
    // swizzle maps there: chunk = slot ^ ((row>>1)&7)
    const half_t* a1[ASEG];
    const half_t* a2[ASEG];
    int a_pb[ASEG], a_oy[ASEG], a_ox[ASEG], a_ch[ASEG];
#pragma unroll
    for (int j = 0; j < ASEG; ++j) {
        const int row = (wave * ASEG + j) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int m = min(m0 + row, p.M - 1);
        a_ch[j] = chunk * 8;
        if constexpr (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            a_oy[j] = rem / p.Wout;
            a_ox[j] = rem - a_oy[j] * p.Wout;
            a_pb[j] = b * p.Hin * p.Win;
            a1[j] = a2[j] = p.A;
        } else {
            a1[j] = p.A + bz * p.sA + (long)m * p.lda + chunk * 8;
            a2[j] = p.A2 ? p.A2 + bz * p.sA2 + (long)m * p.lda2 + chunk * 8 : a1[j];
            a_pb[j] = a_oy[j] = a_ox[j] = 0;
        }
    }
    const half_t* wrow[BSEG];
#pragma unroll
    for (int j = 0; j < BSEG; ++j) {
        const int row = (wave * BSEG + j) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int n = min(n0 + row, p.N - 1);
        wrow[j] = p.W + bz * p.sW + (long)n * p.ldw + chunk * 8;
    }

    auto issue = [&](int kt, int buf) {
        const int k0 = kt * BK;
        char* dA = sA + buf * BM * 128 + wave * ASEG * 1024;
        char* dB = sB + buf * BN * 128 + wave * BSEG * 1024;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;
            const int ci0 = k0 - tap * p.Cin;
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int j = 0; j < ASEG; ++j) {
                int iy, ix;
                bool ok;
                if (p.upsample) {
                    const int uy = a_oy[j] + ky - 1, ux = a_ox[j] + kx - 1;
                    ok = (uy >= 0) & (uy < p.Hout) & (ux >= 0) & (ux < p.Wout);
                    iy = nearest_src(uy, p.up_sy, p.Hin);
                    ix = nearest_src(ux, p.up_sx, p.Win);
                } else {
                    iy = a_oy[j] * p.cstride + ky - 1;
                    ix = a_ox[j] * p.cstride + kx - 1;
                    ok = (iy >= 0) & (iy < p.Hin) & (ix >= 0) & (ix < p.Win);
                }
                const long off = ((long)a_pb[j] + (long)iy * p.Win + ix) * p.Cin + ci0 + a_ch[j];
                const void* src = ok ? (const void*)(p.A + off) : (const void*)g_zero_page;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dA + j * 1024), 16, 0, 0);
            }
        } else {
            const bool first = k0 < p.K1;
#pragma unroll
            for (int j = 0; j < ASEG; ++j) {
                const half_t* src = first ? a1[j] + k0 : a2[j] + (k0 - p.K1);
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dA + j * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < BSEG; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(wrow[j] + k0), (lds_void*)(dB + j * 1024), 16, 0, 0);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Schedule (per k-tile, measured: profiles/r01_gemm_ablation.txt): LDS-DMA writes and ds_reads contend for the LDS,
    // so they are kept apart in time.  (1) counted vmcnt + barrier: tile kt has landed everywhere; (2) ALL fragments of
    // tile kt go LDS -> registers; (3) barrier: the buffer is drained, so (4) the DMA of tile kt+2 is issued into it
    // (tile kt+1 is already in flight in the other buffer: two tiles of latency cover with two buffers); (5) the 16
    // MFMAs run out of registers while the DMA lands.  Raw s_barrier: __syncthreads() would drain vmcnt to 0.
    constexpr int PIECES = ASEG + BSEG;  // LDS-DMA instructions per wave per tile
    if constexpr (STAGES >= 3) {
        // ---- ring: tiles kt+1 .. kt+STAGES-2 stay in flight while tile kt is consumed.  Passing the barrier of iteration
        // kt means every wave has retired its fragment reads of tile kt-1, so that buffer is refilled right away.
        auto wait_newer = [&](int tiles) {  // all LDS-DMA older than the `tiles` newest k-tiles has landed
            if (tiles >= 2) {
                if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (PIECES == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else if (tiles == 1) {
                if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if constexpr (PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        };
        static_assert(STAGES <= 4, "wait_newer covers at most two newer tiles in flight");
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) issue(s, s);
        int buf = 0, fill = STAGES - 1;  // buffer of tile kt / buffer the next issued tile goes to
        for (int kt = 0; kt < nk; ++kt) {
            wait_newer(min(STAGES - 2, nk - 1 - kt));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, fill);
            const char* cA = sA + buf * BM * 128;
            const char* cB = sB + buf * BN * 128;
            h8 af[4][MI], bf[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ch = kk * 2 + lhi;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int r = wm * (BM / 2) + mi * 32 + l31;
                    af[kk][mi] = *reinterpret_cast<const h8*>(cA + r * 128 + swz(r, ch));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int r = wn * 64 + ni * 32 + l31;
                    bf[kk][ni] = *reinterpret_cast<const h8*>(cB + r * 128 + swz(r, ch));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk][ni], af[kk][mi], acc[mi][ni], 0, 0, 0);
            buf = buf + 1 == STAGES ? 0 : buf + 1;
            fill = fill + 1 == STAGES ? 0 : fill + 1;
        }
        __syncthreads();
        epilogue<BM, BM / 2, 256, !CONV>(p, acc, smem, m0, n0, bz, wm, wn, l31, lhi, tid);
        return;
    }
    const bool deep = STAGES == 2 && !(p.debug & 8);  // debug 8: one barrier per tile, DMA one tile ahead (A/B switch)
    issue(0, 0);
    if (nk > 1 && deep) issue(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = STAGES == 2 ? (kt & 1) : 0;
        if (kt + 1 < nk && deep) {
            if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* cA = sA + buf * BM * 128;
        const char* cB = sB + buf * BN * 128;
        h8 af[4][MI], bf[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + lhi;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int r = wm * (BM / 2) + mi * 32 + l31;
                af[kk][mi] = *reinterpret_cast<const h8*>(cA + r * 128 + swz(r, ch));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int r = wn * 64 + ni * 32 + l31;
                bf[kk][ni] = *reinterpret_cast<const h8*>(cB + r * 128 + swz(r, ch));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STAGES == 1) {
            if (kt + 1 < nk) {  // the only buffer is drained once everyone has its fragments: refill it under the MFMAs
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue(kt + 1, 0);
            }
        } else if (deep) {
            if (kt + 2 < nk) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue(kt + 2, buf);
            }
        } else if (kt + 1 < nk) {
            issue(kt + 1, buf ^ 1);  // the other buffer was drained before this iteration's first barrier
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[kk][ni], af[kk][mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    epilogue<BM, BM / 2, 256, !CONV>(p, acc, smem, m0, n0, bz, wm, wn, l31, lhi, tid);
}

// ---------------------------------------------------------------------------------------------- register staging
template <int BM, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(const GemmParams p) {
    constexpr int MI = BM / 64;   // 32-row fragments per wave along M
    constexpr int ACH = BM / 32;  // 16-byte A chunks per thread per k-tile
    constexpr int BCH = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + 2 * BM * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const long bz = blockIdx.z;

    const int c8 = tid & 7;   // 16-byte chunk within the 128-byte k-row
    const int r0 = tid >> 3;  // first row handled by this thread (then +32 per chunk)

    const half_t* a1[ACH];
    const half_t* a2[ACH];
    bool a_ok[ACH];
    int a_pb[ACH], a_oy[ACH], a_ox[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        const int m = m0 + r0 + 32 * j;
        a_ok[j] = m < p.M;
        const int mc = a_ok[j] ? m : 0;
        if constexpr (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = mc / hw, rem = mc - b * hw;
            a_oy[j] = rem / p.Wout;
            a_ox[j] = rem - a_oy[j] * p.Wout;
            a_pb[j] = b * p.Hin * p.Win;
            a1[j] = p.A;
            a2[j] = p.A;
        } else {
            a1[j] = p.A + bz * p.sA + (long)mc * p.lda + c8 * 8;
            a2[j] = p.A2 ? p.A2 + bz * p.sA2 + (long)mc * p.lda2 + c8 * 8 : a1[j];
            a_pb[j] = a_oy[j] = a_ox[j] = 0;
        }
    }
    const half_t* wrow[BCH];
    bool w_ok[BCH];
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        const int n = n0 + r0 + 32 * j;
        w_ok[j] = n < p.N;
        wrow[j] = p.W + bz * p.sW + (long)(w_ok[j] ? n : 0) * p.ldw + c8 * 8;
    }

    h8 ra[ACH], rb[BCH];
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int kc = k0 + c8 * 8;
        const bool k_ok = kc < p.K;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;
            const int ci = k0 - tap * p.Cin + c8 * 8;
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                int iy, ix;
                bool ok;
                if (p.upsample) {
                    const int uy = a_oy[j] + ky - 1, ux = a_ox[j] + kx - 1;
                    ok = (uy >= 0) & (uy < p.Hout) & (ux >= 0) & (ux < p.Wout);
                    iy = nearest_src(uy, p.up_sy, p.Hin);
                    ix = nearest_src(ux, p.up_sx, p.Win);
                } else {
                    iy = a_oy[j] * p.cstride + ky - 1;
                    ix = a_ox[j] * p.cstride + kx - 1;
                    ok = (iy >= 0) & (iy < p.Hin) & (ix >= 0) & (ix < p.Win);
                }
                ok = ok & a_ok[j] & k_ok;
                const long off = ((long)a_pb[j] + (long)iy * p.Win + ix) * p.Cin + ci;
                ra[j] = ok ? *reinterpret_cast<const h8*>(p.A + off) : zero8;
            }
        } else {
            const bool first = kc < p.K1;
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const half_t* src = first ? a1[j] + k0 : a2[j] + (k0 - p.K1);
                ra[j] = (a_ok[j] & k_ok) ? *reinterpret_cast<const h8*>(src) : zero8;
            }
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j)
            rb[j] = (w_ok[j] & k_ok) ? *reinterpret_cast<const h8*>(wrow[j] + k0) : zero8;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int row = r0 + 32 * j;
            *reinterpret_cast<h8*>(sA + buf * BM * 128 + row * 128 + swz(row, c8)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int row = r0 + 32 * j;
            *reinterpret_cast<h8*>(sB + buf * BN * 128 + row * 128 + swz(row, c8)) = rb[j];
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        mma_tile<BM / 2>(sA + buf * BM * 128, sB + buf * BN * 128, acc, wm, wn, l31, lhi);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    epilogue<BM, BM / 2, 256>(p, acc, smem, m0, n0, bz, wm, wn, l31, lhi, tid);
}

template <int BM, bool CONV, bool GLDS>
int launch(const GemmParams& p0, int batch, hipStream_t stream) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = 2 * BM * 128 + 2 * BN * 128;
    auto kern = GLDS ? gemm_glds_kernel<BM, CONV> : gemm_f16_kernel<(BM > 128 ? 128 : BM), CONV>;
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    }
    dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

template <int BM, bool CONV>
int launch_glds1(const GemmParams& p0, int batch, hipStream_t stream) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = 128 * CS_STRIDE + 1024;  // >= (BM + BN) * 128: the epilogue staging tile is the larger tenant (+ the column statistics of the fused LayerNorm's operand-swapped form)
    dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
    hipLaunchKernelGGL((gemm_glds_kernel<BM, CONV, 1>), grid, dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

// ring-buffered variant for small grids: STAGES buffers of (64 + 128) x 128 B; 4 stages = 96 KiB (one block per CU),
// 3 stages = 72 KiB (two blocks per CU)
template <int STAGES>
int launch_ring(const GemmParams& p0, int batch, hipStream_t stream) {
    constexpr int BM = 64;
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = (size_t)STAGES * (BM + BN) * 128;
    auto kern = gemm_glds_kernel<BM, false, STAGES>;
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    }
    dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

enum Kind { K_REG, K_GLDS2, K_GLDS1, K_PP, K_HALO, K_RING, K_T160, K_G320 };
struct Choice {
    Kind kind;
    int bm;   // rows of the block tile
};

// Dispatch (measured on MI355X, profiles/r01_gemm_variants_microbench.txt).  What pays on this chip is the number of
// INDEPENDENT blocks resident per CU (their load / fragment / MFMA phases de-phase for free), not prefetch depth:
//   plain GEMM: one 32-KiB LDS buffer, 128x128 tiles, 3 blocks per CU (+5..28 % over two buffers / 2 blocks),
//               except the long-K (K >= 4096), narrow-N FF down-projection where the 2-deep DMA wins;
//   3x3 conv:   one buffer, 64x128 tiles, 4 blocks per CU (+15..25 %); small grids: 64-row tiles so they cover the CUs.
// g_gemm_variant != 0 forces one family for A/B runs.
Choice choose(const GemmParams& p, int batch) {
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
    const bool conv = p.conv != 0;
    const bool small = tiles128 < 384 || p.M <= 64;
    const bool dma_ok = p.K % 64 == 0;
    Choice c;
    c.bm = small ? 64 : 128;
    if (!dma_ok || g_gemm_variant == 1) {
        c.kind = K_REG;
        return c;
    }
    // stride-1 3x3 convolution on 8 x 16 pixel patches: A staged once per channel slice (conv_halo.hip)
    if (conv && (g_gemm_variant == 0 || g_gemm_variant == 10) && ds_conv_halo_applicable(p)) {
        c.kind = K_HALO;
        c.bm = 128;
        return c;
    }
    switch (g_gemm_variant) {
        case 3:
            if (ds_gemm_pp_applicable(p)) { c.kind = K_PP; c.bm = 256; return c; }
            break;
        case 11:
            if (ds_gemm_t160_possible(p, batch)) { c.kind = K_T160; c.bm = 64; return c; }
            break;
        case 2: c.kind = K_GLDS2; return c;
        case 7: c.kind = K_GLDS2; c.bm = 64; return c;
        case 8: c.kind = K_GLDS1; return c;
        case 9: c.kind = K_GLDS1; c.bm = 64; return c;
        default: break;
    }
    if (g_gemm_variant != 0) {  // forced family did not apply to this shape
        c.kind = K_GLDS2;
        return c;
    }
    if (conv) {
        c.kind = K_GLDS1;
        c.bm = 64;
    } else if (ds_gemm_t160_applicable(p, batch)) {
        // 64 x 160 tiles = exactly one block per CU where the 64 x 128 grid leaves a quarter of the CUs with two (gemm_t160.hip)
        c.kind = K_T160;
        c.bm = 64;
    } else if (ds_gemm_g320_plain_applicable(p, batch)) {
        // 256 x 320 tiles = one block per CU where the 256 x 256 grid is 1.25 rounds (gemm_g320.hip: q|k at M = 8192, N = 2560)
        c.kind = K_G320;
        c.bm = 256;
    } else if (small) {
        c.kind = K_GLDS1;
        // grids that leave a CU with at most two 64 x 128 blocks: nothing hides the DMA latency of the one-buffer kernel
        // (33 us per launch at M = 2048, profiles/r02_ring_in_pipeline_ab.txt) -> ring of 4 (one block per CU) or 3 (two) buffers
        const long blocks64 = (long)((p.M + 63) / 64) * ((p.N + 127) / 128) * batch;
        if (g_gemm_ring != 1 && p.K >= 256 && blocks64 <= 512) {  // every 64-row grid (<= 768 blocks): -1 % (profiles/r02_ring_wide_ab.txt)
            c.kind = K_RING;
            c.bm = blocks64 <= 256 ? 4 : 3;   // bm carries the ring depth for this kind
        }
    } else {
        c.kind = (p.K >= 4096 && p.N <= 2048) ? K_GLDS2 : K_GLDS1;
        // 256 x 256 persistent ping-pong kernel (gemm_pp.hip), grid = ceil(tiles / rounds) blocks (every round full).  It
        // pays when the useful fraction of the rounds x 256 CU-tiles it occupies is high enough (interleaved A/B at UNet
        // batches 6..16, profiles/r02_pp_dispatch_ab.txt; whole rounds: profiles/r01_gemm_pp_microbench.txt):
        //   * more than one round: >= 75 % useful (M = 10240 x N = 2560 = 400 tiles: 77 us vs 89 us), or >= 60 % with a
        //     long K (>= 4096: the FF down-projection at 320 tiles: 229 us vs 244 us); N = 640 counts its 17 % column padding;
        //   * a single partial round: >= 144 tiles with K, N >= 1280 (num_samples 4: 160 tiles, 111 us vs 145 us; 120 tiles
        //     lose: 115 us vs 90 us);
        //   * the short-K, narrow-N projection (N, K <= 640) only where its ragged third tile column takes the branch-free
        //     epilogue (plain epilogue, N % 64 == 0, whole 256-row tiles: gemm_pp.hip `strips_ok`, round 6).  On the generic
        //     epilogue it lost 6..20 % to the 128 x 128 kernels (rounds 1-5); knob "gemm_pp_narrow" 1 restores that rule (A/B).
        // Batched problems (the V^T projections, one GEMM per image with a shared A = Wv): their items are folded into the
        // kernel's tile walk, so the rules apply to the whole batch (round 3: 20 tiles x 32 images = 2.5 rounds run as 3).
        if (ds_gemm_pp_applicable(p)) {
            const long t = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
            const long rounds = (t + 255) / 256;
            const double useful = (double)p.M * p.N * batch / ((double)rounds * 256 * 65536);
            const bool multi = t > 256 && (useful >= 0.75 || (p.K >= 4096 && useful >= 0.6));
            const bool single = t >= 144 && t <= 256 && p.K >= 1280 && p.N >= 1280;
            const bool strips = g_pp_narrow != 1 && p.N % 64 == 0 && p.M % 256 == 0 && p.epi == EPI_NONE && !p.rowbias &&
                                (g_gemm_debug & 4096) == 0;
            if ((multi || single) && (!(p.N <= 640 && p.K <= 640) || strips)) {
                c.kind = K_PP;
                c.bm = 256;
            }
        }
    }
    return c;
}

}  // namespace

// Would ds_launch_gemm run this plain f16 GEMM on gemm_pp_kernel with every tile on a branch-free epilogue?  (The launch-plan
// builder asks before it replaces a LayerNorm launch by the fused producer / consumer pair.)
bool ds_gemm_pp_fast_path(int M, int N, int K, int batch, int epi) {
    return ds_gemm_ln_kind(M, N, K, batch, epi) == 1;
}

// Which fused-LayerNorm implementation would ds_launch_gemm use for this plain f16 GEMM: 1 = gemm_pp_kernel (every tile on
// a branch-free epilogue: M, N multiples of 256; consumers read the (mean, rstd) of ds_launch_ln_finalize), 2 = the 128-wide
// LDS-DMA kernels (their shared epilogue; consumers finalise the partial sums themselves: GemmParams::ln_partial), 0 = neither
// (register-staged fallback, ragged N, a forced A/B family).
int ds_gemm_ln_kind(int M, int N, int K, int batch, int epi) {
    GemmParams p;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N; p.epi = epi; p.K1 = K;
    // the 320-packed GEGLU projection (gemm_g320_kernel) consumes partial sums like the 128-wide kernels
    if (epi == EPI_GEGLU320) return (M > 0 && batch == 1 && N > 0 && N % 320 == 0 && K > 0 && K % 64 == 0) ? 2 : 0;
    if (M <= 0 || N <= 0 || K <= 0 || batch < 1 || (epi != EPI_NONE && epi != EPI_GEGLU)) return 0;
    // gemm_pp_kernel's fused epilogues: whole 256-row tiles; whole 256-column tiles, or - plain epilogue, unbatched (the batched
    // problem is the operand-swapped consumer, whose statistics run along the tile columns) - whole 64-column strips
    const bool cols_ok = N % 256 == 0 || (N % 64 == 0 && epi == EPI_NONE && batch == 1 && (g_gemm_debug & 4096) == 0);
    // (rows: whole tiles; the operand-swapped consumer - the batched problem - skips 32-row pieces past M)
    const bool rows_ok = M % 256 == 0 || (batch > 1 && M % 32 == 0);
    if (g_gemm_variant == 3) return (rows_ok && cols_ok && ds_gemm_pp_applicable(p)) ? 1 : 0;
    if (g_gemm_variant != 0) return 0;
    const Kind k = choose(p, batch).kind;
    if (k == K_PP) return (rows_ok && cols_ok) ? 1 : 0;
    if ((k == K_GLDS1 || k == K_GLDS2 || k == K_RING || k == K_T160 || k == K_G320) && K % 64 == 0) return 2;   // (a PRODUCER also needs N % 128 == 0 and no batch)
    return 0;
}

const char* ds_gemm_kernel_name(const GemmParams& p, int batch) {
    if (p.epi == EPI_GEGLU320) return "gemm_g320_kernel";
    // the fused-LayerNorm instantiations of gemm_pp_kernel are kernels of their own in a rocprofv3 trace (template argument FUSE)
    if (p.ln_stats && !p.ln_partial) return p.ln_swapped ? "gemm_pp_kernel<0,4>" : p.epi == EPI_GEGLU ? "gemm_pp_kernel<0,9>" : "gemm_pp_kernel<0,1>";
    if (p.stats_out && ds_gemm_ln_kind(p.M, p.N, p.K, batch, p.epi) == 1) return "gemm_pp_kernel<0,2>";
    const Choice c = choose(p, batch);
    const bool conv = p.conv != 0;
    switch (c.kind) {
        case K_PP: return "gemm_pp_kernel<0,0>";
        case K_HALO: return "conv_halo_kernel";
        case K_RING: return c.bm == 4 ? "gemm_glds_kernel<64,false,4>" : "gemm_glds_kernel<64,false,3>";
        case K_T160: return ds_gemm_t160_rows(p.M, p.N, p.K, batch) == 128 ? "gemm_t160_kernel<128 rows>" : "gemm_t160_kernel";
        case K_G320: return "gemm_g320_kernel<plain>";
        case K_GLDS1:
            if (c.bm == 128) return conv ? "gemm_glds_kernel<128,true,1>" : "gemm_glds_kernel<128,false,1>";
            return conv ? "gemm_glds_kernel<64,true,1>" : "gemm_glds_kernel<64,false,1>";
        case K_GLDS2:
            if (c.bm == 128) return conv ? "gemm_glds_kernel<128,true,2>" : "gemm_glds_kernel<128,false,2>";
            return conv ? "gemm_glds_kernel<64,true,2>" : "gemm_glds_kernel<64,false,2>";
        default:
            if (c.bm == 128) return conv ? "gemm_f16_kernel<128,true>" : "gemm_f16_kernel<128,false>";
            return conv ? "gemm_f16_kernel<64,true>" : "gemm_f16_kernel<64,false>";
    }
}

// GroupNorm statistics out of a 3x3 convolution: only the halo-patch kernels emit them, so the answer follows the dispatch
int ds_gemm_conv_gn_chunks(const GemmParams& p) {
    if (!p.conv || choose(p, 1).kind != K_HALO) return 0;
    return ds_conv_halo_gn_chunks(p);
}

int ds_launch_gemm(const GemmParams& p_in, int batch, hipStream_t stream) {
    GemmParams p = p_in;
    p.debug = g_gemm_debug;
    DS_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    DS_REQUIRE(p.N % 8 == 0 && p.K % 8 == 0, "gemm: N (%d) and K (%d) must be multiples of 8", p.N, p.K);
    DS_REQUIRE(p.ldc % 8 == 0 && p.ldw % 8 == 0, "gemm: ldc/ldw must be multiples of 8");
    const bool conv = p.conv != 0;
    if (conv) {
        DS_REQUIRE(p.Cin % 64 == 0, "conv3x3: Cin (%d) must be a multiple of 64", p.Cin);
        DS_REQUIRE(p.K == 9 * p.Cin, "conv3x3: K (%d) != 9*Cin (%d)", p.K, 9 * p.Cin);
        DS_REQUIRE(batch == 1, "conv3x3: batch is folded into M");
    } else {
        DS_REQUIRE(p.lda % 8 == 0, "gemm: lda must be a multiple of 8");
        DS_REQUIRE(p.A2 == nullptr || (p.K1 % 64 == 0 && p.lda2 % 8 == 0), "gemm: split-A needs K1 %% 64 == 0");
    }
    if (p.epi == EPI_GEGLU) DS_REQUIRE(p.N % 128 == 0, "geglu: packed N (%d) must be a multiple of 128", p.N);
    if (p.epi == EPI_GEGLU320) {   // W packed in 320-row groups (engine.pack_geglu320): one kernel reads that layout
        DS_REQUIRE(ds_gemm_g320_possible(p, batch), "geglu320: plain f16 GEMM, batch 1, N %% 320 == 0, K %% 64 == 0, no residual (M=%d N=%d K=%d)", p.M, p.N, p.K);
        return ds_launch_gemm_g320(p, stream);
    }
    Choice c = choose(p, batch);
    DS_REQUIRE(!p.gn_partial || (conv && c.kind == K_HALO && p.dtype == DS_DTYPE_F16),
               "conv3x3: GroupNorm statistics come out of the halo-patch kernels only (ask ds_conv3x3_gn_chunks first)");
    if (p.ln_stats || p.ln_c || p.stats_out) {
        // Fused LayerNorm.  The planner asks ds_gemm_ln_fusable which form the dispatch gives a shape and builds the pair
        // accordingly; a direct call is served by whichever family implements the role it names:
        //   consumer of finalised statistics, operand-swapped consumer: gemm_pp_kernel (whole 256 x 256 tiles);
        //   consumer of partial sums: the 128-wide kernels (shared epilogue of this file);
        //   producer: the consumer's family if the call is both, else what the dispatch picks, else whichever can.
        DS_REQUIRE(!conv && p.dtype == DS_DTYPE_F16 && !p.A2 && !p.rowbias, "gemm: fused LayerNorm is a plain f16 GEMM feature");
        const bool pp_ok = ds_gemm_pp_applicable(p) &&
                           (p.ln_swapped || (p.M % 256 == 0 && (p.N % 256 == 0 || (p.N % 64 == 0 && p.epi == EPI_NONE && (g_gemm_debug & 4096) == 0))));
        const bool wide_ok = batch == 1 && p.N % 128 == 0 && p.K % 64 == 0;
        const bool wide_kind = c.kind == K_GLDS1 || c.kind == K_GLDS2 || c.kind == K_RING || c.kind == K_T160 || c.kind == K_G320;
        auto force_wide = [&]() {
            if (!wide_kind) { c.kind = K_GLDS1; c.bm = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) < 384 ? 64 : 128; }
        };
        if (p.ln_stats && p.ln_partial) {
            DS_REQUIRE(p.K % 64 == 0 && p.ln_c && (p.ln_swapped ? (p.ln_rows > 0 && p.ln_bstride > 0 && !p.bias) : batch == 1),
                       "gemm: a consumer of partial LayerNorm sums needs K %% 64 == 0 (M=%d N=%d K=%d)", p.M, p.N, p.K);
            force_wide();
        } else if (p.ln_stats) {
            DS_REQUIRE(pp_ok && p.ln_c, "gemm: a consumer of finalised LayerNorm statistics needs the 256 x 256 kernel (M=%d N=%d K=%d)", p.M, p.N, p.K);
            c.kind = K_PP;
        }
        if (p.stats_out) {
            DS_REQUIRE(p.epi == EPI_NONE && !p.ln_swapped, "gemm: LayerNorm statistics come out of the plain epilogue only");
            if (p.ln_stats) {
                DS_REQUIRE(c.kind == K_PP || wide_ok, "gemm: LayerNorm statistics cannot be emitted for M=%d N=%d K=%d", p.M, p.N, p.K);
            } else if (c.kind == K_PP ? !pp_ok : !(wide_kind && wide_ok)) {
                DS_REQUIRE(pp_ok || wide_ok, "gemm: LayerNorm statistics cannot be emitted for M=%d N=%d K=%d", p.M, p.N, p.K);
                if (pp_ok) c.kind = K_PP; else force_wide();
            }
        }
    }
    if (p.dtype != DS_DTYPE_F16) {  // bf16 (VAE decoder): only the two kernels that are templated on the element type
        if (conv) {
            DS_REQUIRE(ds_conv_halo_applicable(p), "conv3x3 bf16: needs stride 1 and Cin %% 64 == 0");
            c.kind = K_HALO;
        } else {
            DS_REQUIRE(ds_gemm_pp_applicable(p), "gemm bf16: needs M, N %% 16 == 0 and K %% 128 == 0 (M=%d N=%d K=%d)", p.M, p.N, p.K);
            c.kind = K_PP;
        }
    }
    switch (c.kind) {
        case K_PP: return ds_launch_gemm_pp(p, batch, stream);
        case K_HALO: return ds_launch_conv_halo(p, stream);
        case K_RING: return c.bm == 4 ? launch_ring<4>(p, batch, stream) : launch_ring<3>(p, batch, stream);
        case K_T160: return ds_launch_gemm_t160(p, stream);
        case K_G320: return ds_launch_gemm_g320(p, stream);
        case K_GLDS1:
            if (c.bm == 128)
                return conv ? launch_glds1<128, true>(p, batch, stream) : launch_glds1<128, false>(p, batch, stream);
            return conv ? launch_glds1<64, true>(p, batch, stream) : launch_glds1<64, false>(p, batch, stream);
        case K_GLDS2:
            if (c.bm == 128) return conv ? launch<128, true, true>(p, batch, stream) : launch<128, false, true>(p, batch, stream);
            return conv ? launch<64, true, true>(p, batch, stream) : launch<64, false, true>(p, batch, stream);
        default:
            if (c.bm == 128) return conv ? launch<128, true, false>(p, batch, stream) : launch<128, false, false>(p, batch, stream);
            return conv ? launch<64, true, false>(p, batch, stream) : launch<64, false, false>(p, batch, stream);
    }
}

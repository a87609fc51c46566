// fp16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+bias, +row-group bias, +residual, GEGLU) and the
// implicit-GEMM 3x3 convolution over NHWC activations that shares its main loop.
//
// Replaces, on the DiffSensei UNet hot path (reference src/models/unet.py:206,244-338 -> diffusers blocks):
//   * every nn.Linear (to_q/to_k/to_v/to_out, proj_in/proj_out, GEGLU ff.net.0.proj, ff.net.2)
//     reference src/models/attention_processor.py:56,63,64,84,207,225,226,245,246,261
//   * every 3x3 conv of ResnetBlock2D / Downsample2D / Upsample2D and the 1x1 conv_shortcut
//
// CDNA4 design: 256 threads = 4 waves (2x2), block tile BM x 128 x 64, v_mfma_f32_32x32x16_f16 with the
// operands swapped (D[n][m]) so each lane ends with 4 consecutive output columns of one row; tiles are
// staged global -> VGPR -> LDS (issue-early / write-late) into two LDS buffers whose 16-byte chunks are
// XOR-swizzled (chunk ^= (row>>1)&7) so every ds_read_b128 lane group hits 16 distinct slots; the
// epilogue is staged through LDS so global stores are full 16 B/lane rows.  Convolution gathers its A
// tile straight from the NHWC tensor (zero-filled halo, optional stride 2, optional fused nearest x2
// upsample) — no im2col buffer ever exists in HBM.
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr int BN = 128;
constexpr int BK = 64;
constexpr int CS_STRIDE = 272;  // bytes per row of the epilogue staging tile (128 f16 + 8 pad)

__device__ __forceinline__ int swz(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

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
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const long bz = blockIdx.z;

    const int c8 = tid & 7;       // 16-byte chunk within the 128-byte k-row
    const int r0 = tid >> 3;      // first row handled by this thread (then +32 per chunk)

    // ---- per-thread A row descriptors
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
                    iy = uy >> 1;
                    ix = ux >> 1;
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
        const char* cA = sA + buf * BM * 128;
        const char* cB = sB + buf * BN * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            h8 af[MI], bf[2];
            const int ch = kk * 2 + lhi;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int r = wm * (BM / 2) + mi * 32 + l31;
                af[mi] = *reinterpret_cast<const h8*>(cA + r * 128 + swz(r, ch));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int r = wn * 64 + ni * 32 + l31;
                bf[ni] = *reinterpret_cast<const h8*>(cB + r * 128 + swz(r, ch));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue, stage 1: bias / row-group bias, round to f16, park the tile in LDS as [m][n]
    // D layout (operands swapped): lane holds row m_local = ..+(lane&31); regs r -> n = (r&3)+8*(r>>2)+4*(lane>>5)
    char* sC = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ml = wm * (BM / 2) + mi * 32 + l31;
        const int m = m0 + ml;
        const int grp = p.rowbias ? ((m < p.M ? m : p.M - 1) / p.rows_per_group) : 0;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + ni * 32 + 8 * g + 4 * lhi;
                const int n = n0 + nl;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
                if (n < p.N) {
                    if (p.bias) {
                        const h4 bv = *reinterpret_cast<const h4*>(p.bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
                    }
                    if (p.rowbias) {
                        const h4 bv = *reinterpret_cast<const h4*>(p.rowbias + (long)grp * p.rowbias_ld + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
                    }
                }
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
                *reinterpret_cast<h4*>(sC + ml * CS_STRIDE + nl * 2) = o;
            }
        }
    }
    __syncthreads();

    // ---- epilogue, stage 2: coalesced 16-byte rows out of LDS (+ residual, or GEGLU pairing)
    half_t* Cg = p.C + bz * p.sC;
    if (p.epi == EPI_GEGLU) {
        // packed weight rows: each 128-row tile = 64 "hidden" columns followed by their 64 "gate" columns
        constexpr int IT = BM * 8 / 256;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int id = tid + 256 * j;
            const int row = id >> 3, c = id & 7;
            const int m = m0 + row, n = (n0 >> 1) + c * 8;
            if (m < p.M && n < (p.N >> 1)) {
                const h8 hv = *reinterpret_cast<const h8*>(sC + row * CS_STRIDE + c * 16);
                const h8 gv = *reinterpret_cast<const h8*>(sC + row * CS_STRIDE + 128 + c * 16);
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const half_t ge = (half_t)ds_gelu_erf((float)gv[e]);
                    o[e] = (half_t)((float)hv[e] * (float)ge);
                }
                *reinterpret_cast<h8*>(Cg + (long)m * p.ldc + n) = o;
            }
        }
    } else {
        constexpr int IT = BM * 16 / 256;
        const half_t* Rg = p.residual ? p.residual + bz * p.sR : nullptr;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int id = tid + 256 * j;
            const int row = id >> 4, c = id & 15;
            const int m = m0 + row, n = n0 + c * 8;
            if (m < p.M && n < p.N) {
                h8 v = *reinterpret_cast<const h8*>(sC + row * CS_STRIDE + c * 16);
                if (p.epi == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)ds_gelu_erf((float)v[e]);
                } else if (p.epi == EPI_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        v[e] = (half_t)(f / (1.0f + __expf(-1.702f * f)));
                    }
                }
                if (Rg) {
                    const h8 rv = *reinterpret_cast<const h8*>(Rg + (long)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
                }
                *reinterpret_cast<h8*>(Cg + (long)m * p.ldc + n) = v;
            }
        }
    }
}

template <int BM, bool CONV>
int launch(const GemmParams& p0, int batch, hipStream_t stream) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = 2 * BM * 128 + 2 * BN * 128;
    static bool attr_set = false;
    if (!attr_set) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<BM, CONV>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
    hipLaunchKernelGGL((gemm_f16_kernel<BM, CONV>), grid, dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// pick BM: small problems get 64-row tiles so the grid covers the 256 CUs
bool ds_gemm_uses_small_tile(const GemmParams& p, int batch) {
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
    return tiles128 < 384 || p.M <= 64;
}

int ds_launch_gemm(const GemmParams& p, int batch, hipStream_t stream) {
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
    const bool small = ds_gemm_uses_small_tile(p, batch);
    if (conv) return small ? launch<64, true>(p, batch, stream) : launch<128, true>(p, batch, stream);
    return small ? launch<64, false>(p, batch, stream) : launch<128, false>(p, batch, stream);
}

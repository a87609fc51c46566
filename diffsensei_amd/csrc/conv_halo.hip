// 3x3 convolution (stride 1, optional fused nearest x2 upsample) over NHWC f16 activations as an implicit GEMM whose
// A operand is staged ONCE per 64-channel slice: the large-resolution path of ds_launch_gemm's conv mode
// (ResnetBlock2D conv1/conv2 and Upsample2D.conv on the DiffSensei UNet path, reference src/models/unet.py:244-338 ->
// diffusers blocks; same GemmParams / epilogue semantics as gemm.hip's conv instantiation).
//
// Why.  gemm.hip's conv kernel re-gathers the 64 x 64-channel A tile from L2 for each of the 9 taps; its 64 x 128 tile
// then moves 24 KiB per 1 MFLOP through L2 -> LDS and sits on that path's limit (~38 B/clk/CU, ~650 TFLOP/s).  Here
// a block owns an 8 x 16 patch of output pixels (M tile = 128) x 128 output channels, and per channel slice it
// loads the 10 x 18 halo patch (180 pixels x 128 B = 22.5 KiB) once; the nine taps then read their A fragments from
// that patch at a uniform row shift (ky*18 + kx).  Per slice: 22.5 KiB of A + 9 x 16 KiB of W for 18.9 MFLOP =
// 113 flop/B instead of 43.  LDS: patch 24 KiB + one W tile 16 KiB = 40 KiB -> the three independent blocks per CU that
// the 128 x 128 GEMM relies on (gemm.hip, STAGES = 1 schedule: wait, barrier, all fragments to registers, barrier,
// refill the single W buffer under the 16 MFMAs).
//
// The patch is written by LDS-DMA (lane-linear destination), so it is dense [row][64 ch]; 16-byte chunk c of patch pixel
// q = (qy, qx) sits at slot c ^ ((qx>>1)&7) - swizzled by the patch COLUMN.  A fragment read touches 16 + 16 pixels of two
// adjacent patch rows (q jumps by 18 between lanes 15 and 16), so the row-index swizzle of the other tiles, (q>>1)&7, put
// two lanes of every 16-lane ds_read_b128 group on the same banks: 8 LDS cycles per fragment instead of 4, 40 % of the
// kernel's LDS cycles were conflicts (PMC: profiles/r02_pmc_conv_attn_summary.txt; the bank model in
// tools/lds_bank_model.py reproduces the 40 % and shows the column swizzle conflict-free for all nine taps).  The tap
// shift changes qx per lane, so the reader rebuilds it per tap: base = q*128 + ((lhi ^ ((qx0+kx)>>1)&7) << 4), k-step kk
// at base ^ (kk << 5).
// Halo pixels outside the image come from a zero page.  With upsample the patch is gathered from input pixel
// nearest_src(uy), nearest_src(ux) (= uy>>1, ux>>1 for the plain x2 case): the upsampled tensor never exists.
//
// K order: channel slices outer, taps inner (k = tap*Cin + ci in the weight rows) - the sum is the same set of
// products as the tap-major kernel's, in a different order, so results agree to fp32-accumulation rounding.
#include <type_traits>

#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr int PH = 8, PW = 16;            // output pixels per block: 8 rows x 16 columns = 128 GEMM rows
constexpr int HWD = PW + 2;               // halo patch width
constexpr int HROWS = (PH + 2) * HWD;     // 180 patch pixels
constexpr int PROWS = 192;                // staged rows (24 LDS-DMA pieces of 8 rows)
constexpr int BN = 128;
constexpr int CS_STRIDE = 272;            // bytes per row of the epilogue staging tile (128 f16 + 8 pad)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __attribute__((aligned(256))) char g_halo_zero_page[256];

__device__ __forceinline__ int swz(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

// GroupNorm statistics of the block's output tile (GemmParams::gn_partial).  Stage 2 of the epilogue gives thread t the 8
// channels of chunk t & 15 on rows (t >> 4) + 16 j: it sums the values it STORES (f16, after the residual) and their squares in
// fp32; the four lanes of a wave that share a chunk meet in two shuffles, the four waves in `red` (4 KiB of LDS behind the
// staging tile), and threads 0..127 write one (sum, sum of squares) per channel: partial[b][tile][n0 + ch] - the layout of
// gn_stats_kernel, so gn_finalize_kernel / gn_apply_kernel run unchanged.  A fixed summation order: deterministic, and an
// image's statistics do not depend on its position in the batch.
template <typename V8>
__device__ __forceinline__ void gn_accumulate(const V8& v, float (&gs)[8], float (&gq)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = (float)v[e];
        gs[e] += f;
        gq[e] = fmaf(f, f, gq[e]);
    }
}
__device__ __forceinline__ void gn_emit(const GemmParams& p, float (&gs)[8], float (&gq)[8], char* red_bytes, int tid, int b,
                                        int tile_in_image, int n0) {
    float* const red = reinterpret_cast<float*>(red_bytes);
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gs[e] += __shfl_xor(gs[e], 16, 64);
        gq[e] += __shfl_xor(gq[e], 16, 64);
        gs[e] += __shfl_xor(gs[e], 32, 64);
        gq[e] += __shfl_xor(gq[e], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(wave * 128 + lane * 8 + e) * 2] = gs[e];
            red[(wave * 128 + lane * 8 + e) * 2 + 1] = gq[e];
        }
    }
    __syncthreads();
    if (tid < 128 && n0 + tid < p.N) {
        const float s = (red[tid * 2] + red[(128 + tid) * 2]) + (red[(256 + tid) * 2] + red[(384 + tid) * 2]);
        const float q = (red[tid * 2 + 1] + red[(128 + tid) * 2 + 1]) + (red[(256 + tid) * 2 + 1] + red[(384 + tid) * 2 + 1]);
        f32x2 o2 = {s, q};
        *reinterpret_cast<f32x2*>(p.gn_partial + (((long)b * p.gn_chunks + tile_in_image) * p.N + n0 + tid) * 2) = o2;
    }
}

// Epilogue of the two 8 x 16-pixel kernels (conv_halo_kernel, conv_halo_deep_kernel): the caller has passed its last barrier.
template <typename T>
__device__ __forceinline__ void conv_halo_epilogue8(const GemmParams& p, f32x16 (&acc)[2][2], char* smem, int tid, int wm, int wn,
                                                    int l31, int lhi, int b, int oy0, int ox0, int n0, int tile_in_image) {
    typedef typename Elt<T>::v8 V8;
    typedef typename Elt<T>::v4 V4;
    // ---- epilogue: bias (+ per-image bias), round to f16, park the 128 x 128 tile in LDS, then whole 16-byte pieces
    // of output rows (+ residual).  D layout (operands swapped): lane holds row ..+(lane&31); register r is column
    // (r&3) + 8*(r>>2) + 4*(lane>>5) of its 32-column fragment.
    // (no branch / s_waitcnt per 4 values: biases fetched once with clamped addresses, residual pieces requested in
    // batches of four before they are needed - see conv_halo256_kernel)
    char* const sC = smem;
    V4 b0[2][4], b1[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) b0[ni][g] = b1[ni][g] = V4{0, 0, 0, 0};
    if (p.bias) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b0[ni][g] = *reinterpret_cast<const V4*>(p.bias + min(n0 + wn * 64 + ni * 32 + 8 * g + 4 * lhi, p.N - 4));
    }
    if (p.rowbias) {
        const half_t* rb = p.rowbias + (long)b * p.rowbias_ld;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b1[ni][g] = *reinterpret_cast<const V4*>(rb + min(n0 + wn * 64 + ni * 32 + 8 * g + 4 * lhi, p.N - 4));
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int ms = wm * 64 + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + ni * 32 + 8 * g + 4 * lhi;
                V4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[mi][ni][4 * g + e];
                    v += (float)b0[ni][g][e];
                    v += (float)b1[ni][g][e];
                    o[e] = (T)v;
                }
                *reinterpret_cast<V4*>(sC + ms * CS_STRIDE + nl * 2) = o;
            }
    }
    __syncthreads();
    const T* const Rp = reinterpret_cast<const T*>(p.residual);
    float gs[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
#pragma unroll 1
    for (int j0 = 0; j0 < 8; j0 += 4) {
        long mrow[4];
        int ncol[4];
        bool ok[4];
        V8 rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + 256 * (j0 + u);
            const int row = id >> 4, c = id & 15;
            const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
            ok[u] = (n0 + c * 8 < p.N) & (oy < p.Hout) & (ox < p.Wout);
            mrow[u] = ((long)b * p.Hout + min(oy, p.Hout - 1)) * p.Wout + min(ox, p.Wout - 1);
            ncol[u] = min(n0 + c * 8, p.N - 8);
            if (Rp) rv[u] = *reinterpret_cast<const V8*>(Rp + mrow[u] * p.ldr + ncol[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + 256 * (j0 + u);
            const int row = id >> 4, c = id & 15;
            V8 v = *reinterpret_cast<const V8*>(sC + row * CS_STRIDE + c * 16);
            if (Rp) {
                // f16: one v_pk_add_f16 per two values - the same number as (f16)((float)a + (float)b), which never rounds twice
                // (tests/test_f16_add_equivalence.py); bf16 (VAE decoder) keeps the f32 form
                if constexpr (std::is_same<T, half_t>::value) v = v + rv[u];
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (T)((float)v[e] + (float)rv[u][e]);
                }
            }
            if (ok[u]) *reinterpret_cast<V8*>(reinterpret_cast<T*>(p.C) + mrow[u] * p.ldc + ncol[u]) = v;
            if (p.gn_partial && ok[u]) gn_accumulate(v, gs, gq);
        }
    }
    if (p.gn_partial) gn_emit(p, gs, gq, smem + 128 * CS_STRIDE, tid, b, tile_in_image, n0);
}

template <typename T>  // half_t (UNet) or bf16_t (VAE decoder)
__global__ __launch_bounds__(256, 3) void conv_halo_kernel(const GemmParams p) {
    typedef typename Elt<T>::v8 V8;
    typedef typename Elt<T>::v4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sP = smem;
    char* const sW = smem + PROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int lrow = lane >> 3, slot = lane & 7;
    const bool rowswz = (p.debug & 1024) != 0;  // A/B only: the row-index patch swizzle (2-way bank conflicts on every A read)

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int tiles_x = (p.Wout + PW - 1) / PW, tiles_y = (p.Hout + PH - 1) / PH;  // ragged edges: masked stores
    const int tx = tm % tiles_x, ty = (tm / tiles_x) % tiles_y, b = tm / (tiles_x * tiles_y);
    const int oy0 = ty * PH, ox0 = tx * PW, n0 = tn * BN;

    // ---- patch descriptors: this wave stages patch rows (6w + j) * 8 + lrow, j < 6
    int poff[6];  // element offset of the lane's 16-byte chunk at channel 0, or -1: zero page (halo / pad rows)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = (wave * 6 + j) * 8 + lrow;
        const int qy = q / HWD, qx = q - qy * HWD;
        const int uy = oy0 - 1 + qy, ux = ox0 - 1 + qx;  // output-resolution pixel
        const bool ok = (q < HROWS) & (uy >= 0) & (uy < p.Hout) & (ux >= 0) & (ux < p.Wout);
        const int iy = p.upsample ? nearest_src(uy, p.up_sy, p.Hin) : uy, ix = p.upsample ? nearest_src(ux, p.up_sx, p.Win) : ux;
        const int chunk = slot ^ (((rowswz ? q : qx) >> 1) & 7);  // swizzle by patch COLUMN (see the header)
        poff[j] = ok ? ((b * p.Hin + iy) * p.Win + ix) * p.Cin + chunk * 8 : -1;
    }
    int woff[4];  // element offset of the lane's chunk in the weight matrix at k = 0 (Cout * 9 * Cin < 2^31)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int n = min(n0 + row, p.N - 1);
        woff[j] = n * (int)p.ldw + chunk * 8;
    }
    auto issue_patch = [&](int ci0) {
        char* d = sP + wave * 6 * 1024;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const void* src = poff[j] >= 0 ? (const void*)(p.A + poff[j] + ci0) : (const void*)g_halo_zero_page;
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(d + j * 1024), 16, 0, 0);
        }
    };
    auto issue_w = [&](int k0) {
        char* d = sW + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(p.W + k0 + woff[j]), (lds_void*)(d + j * 1024), 16, 0, 0);
    };

    // ---- A fragment rows: GEMM row r = 64 wm + 32 mi + l31  <->  patch pixel (r >> 4, r & 15), tap (0,0) at q0
    int q0[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) q0[mi] = (wm * 4 + mi * 2 + (l31 >> 4)) * HWD + (l31 & 15);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int slices = p.Cin / 64;
    issue_patch(0);
    issue_w(0);
    for (int s = 0; s < slices; ++s) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int shift = ky * HWD + kx;
            V8 af[4][2], bf[4][2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int q = q0[mi] + shift;
                const int base = q * 128 + ((lhi ^ (((rowswz ? q : (l31 & 15) + kx) >> 1) & 7)) << 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) af[kk][mi] = *reinterpret_cast<const V8*>(sP + (base ^ (kk << 5)));
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int r = wn * 64 + ni * 32 + l31;
                    bf[kk][ni] = *reinterpret_cast<const V8*>(sW + r * 128 + swz(r, kk * 2 + lhi));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const bool last = (s + 1 == slices) & (tap == 8);
            if (!last) {  // the W buffer (and after tap 8 the patch) is drained once everyone holds its fragments
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (tap == 8) {
                    issue_patch((s + 1) * 64);
                    issue_w((s + 1) * 64);
                } else {
                    issue_w((tap + 1) * p.Cin + s * 64);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = Elt<T>::mfma(bf[kk][ni], af[kk][mi], acc[mi][ni]);
        }
    }
    __syncthreads();

    conv_halo_epilogue8<T>(p, acc, smem, tid, wm, wn, l31, lhi, b, oy0, ox0, n0, ty * tiles_x + tx);
}

// ---------------------------------------------------------------------------------------------------------------
// Large-batch variant: 16 x 16 output pixels (M tile = 256) x 128 channels per block.  The 8 x 16 kernel above still
// moves 36 B/clk/CU through L2 -> LDS (its W tile serves only 128 pixels), i.e. it sits on the same fill limit as the
// 128 x 128 GEMMs; doubling the pixels per W tile takes that to 20 B/clk/CU (204 flop/B).  Waves 2 x 2, each
// 128 pixels x 64 channels (acc 128 VGPRs, 0.75 KiB of fragment reads per MFMA instead of 1).  LDS: 18 x 18 halo patch
// 44 KiB + TWO W buffers (the next k-tile's W is in flight under the current tile's MFMAs, one barrier per k-tile) =
// 76 KiB -> two blocks per CU.  The patch is single-buffered: at a slice boundary the block waits for the new patch
// while the co-resident block computes.
// ---------------------------------------------------------------------------------------------------------------
static_assert(128 * CS_STRIDE + 4096 <= PROWS * 128 + BN * 128, "conv_halo_kernel: staging tile + GroupNorm partials exceed its LDS");

// ---------------------------------------------------------------------------------------------------------------
// Small-grid variant of the 8 x 16 kernel (round 6): the same tile (8 x 16 output pixels x 128 channels, 4 waves), but the W
// k-tiles go through a RING of three 16-KiB buffers (two k-tiles in flight) and the halo patch is double-buffered, so one block
// ALONE on a CU never waits out a DMA round trip per k-tile.  Why: conv_halo_kernel hides its single W buffer's latency behind
// the two other blocks of the CU (3 per CU); at UNet batch 2 the 1280-channel level is 160 blocks - one per CU on 160 CUs -
// and every one of its 180 k-tiles costs a full L2 round trip: 175 us per convolution, 344 TFLOP/s, a third of the batch-2
// forward's convolution time in thirteen launches (profiles/r06_conv_deep_ab_b2.txt).  One barrier per k-tile, at its LAST
// k-step: every fragment of the k-tile is in registers by then (they are requested one k-step ahead, two register sets, as in
// conv_halo256_kernel), so: counted vmcnt (W(kt+1) has landed; W(kt+2) and - for two k-tiles per slice - the next slice's patch
// stay in flight), barrier, W(kt+3) into the buffer k-tile kt has just left, the first fragments of k-tile kt+1, the last four
// MFMAs of k-tile kt.  (First version: all sixteen fragment reads at the top of a k-tile, then its sixteen MFMAs - with one wave
// per SIMD nothing overlapped the reads: 120.6 us per 1280 -> 1280 convolution at batch 2 instead of 175.8; this form: see
// profiles/r06_conv_deep_ab_b2.txt.)
// LDS: 2 x 24 KiB patch + 3 x 16 KiB W = 96 KiB -> one block per CU: chosen only for grids of at most one block per CU.
// (Ring depth, measured per 1280 -> 1280 convolution at UNet batch 2 against the single-buffer kernel on the same box: three
// buffers 115.5 vs 176.7 us (0.65), five buffers - 64 KiB in flight - 130.4 vs 178.6 (0.73): more in flight does not help, the
// k-tile is not waiting for its DMA any more; DEEP_NW stays a constant of the source.  Eight waves per block (4 x 2, 32 pixels x
// 64 channels each, two per SIMD) measured the same again: 116.7 vs 177.7 us (profiles/r06_conv_deep_8waves_ab_b2.txt).  What is
// left is the CU's fill path itself: 18.7 KiB per k-tile in 0.64 us = 29 GB/s, between the 21-25 GB/s a CU draws from HBM and
// the 35-45 it draws from the Infinity Cache / L2 whatever is in flight (tools/ubench/lds_fill_rate.hip) - the weights of a
// batch-1 request are read cold.)
// (Later in round 6, three timing experiments on the 1280 -> 1280 convolution at UNet batch 2, 117 us: the same time with hot
// weights - the launch repeated back to back - as inside the forward; half / a quarter of the W pieces issued (garbage results):
// 113.6 / 110.0 us; ALL sixteen fragments of the next k-tile requested ahead of a k-tile's MFMAs (two 64-register sets, the last
// k-tile peeled so the compiler keeps counted lgkmcnt waits): 114.7 us, bit-identical, not kept.  So neither cold weights, nor the
// bytes, nor the LDS round trips hold the 0.64 us per k-tile against 0.24 us of MFMAs; what is left is the chain wait -> barrier ->
// DMA issue -> landing of one block alone on its CU - the guide's "landing cadence 0.64 us per 16-KiB fill" of a single loader.)
// Same products in the same order as conv_halo_kernel: bit-identical results (tests/test_gpu_ops.py).
// ---------------------------------------------------------------------------------------------------------------
constexpr int DEEP_NW = 3;                            // W buffers in the ring: DEEP_NW - 1 k-tiles (32 KiB) in flight per CU
constexpr int DEEP_W0 = 2 * PROWS * 128;              // byte offset of the W ring
constexpr int DEEP_LDS = DEEP_W0 + DEEP_NW * BN * 128;  // 96 KiB

template <int N>
__device__ __forceinline__ void deep_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T>
__global__ __launch_bounds__(256, 1) void conv_halo_deep_kernel(const GemmParams p) {
    typedef typename Elt<T>::v8 V8;
    typedef typename Elt<T>::v4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int lrow = lane >> 3, slot = lane & 7;

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int tiles_x = (p.Wout + PW - 1) / PW, tiles_y = (p.Hout + PH - 1) / PH;
    const int tx = tm % tiles_x, ty = (tm / tiles_x) % tiles_y, b = tm / (tiles_x * tiles_y);
    const int oy0 = ty * PH, ox0 = tx * PW, n0 = tn * BN;

    int poff[6];  // element offset of the lane's 16-byte chunk at channel 0, or -1: zero page (halo / pad rows)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = (wave * 6 + j) * 8 + lrow;
        const int qy = q / HWD, qx = q - qy * HWD;
        const int uy = oy0 - 1 + qy, ux = ox0 - 1 + qx;
        const bool ok = (q < HROWS) & (uy >= 0) & (uy < p.Hout) & (ux >= 0) & (ux < p.Wout);
        const int iy = p.upsample ? nearest_src(uy, p.up_sy, p.Hin) : uy, ix = p.upsample ? nearest_src(ux, p.up_sx, p.Win) : ux;
        const int chunk = slot ^ ((qx >> 1) & 7);  // swizzle by patch COLUMN (see the file header)
        poff[j] = ok ? ((b * p.Hin + iy) * p.Win + ix) * p.Cin + chunk * 8 : -1;
    }
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int n = min(n0 + row, p.N - 1);
        woff[j] = n * (int)p.ldw + chunk * 8;
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    auto issue_patch = [&](int ci0, int pbuf) {
        const unsigned d = lds0 + pbuf * (PROWS * 128) + wave * 6 * 1024;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const void* src = poff[j] >= 0 ? (const void*)(p.A + poff[j] + ci0) : (const void*)g_halo_zero_page;
            lds_dma16_v(src, d + j * 1024);
        }
    };
    auto issue_w = [&](int k0, int wbuf) {
        const unsigned d = lds0 + DEEP_W0 + wbuf * (BN * 128) + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_v(p.W + k0 + woff[j], d + j * 1024);
    };

    int q0[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) q0[mi] = (wm * 4 + mi * 2 + (l31 >> 4)) * HWD + (l31 & 15);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int slices = p.Cin / 64;
    const int nkt = slices * 9;
    auto w_k0 = [&](int kt) {   // k-tile kt = (slice, tap): column offset of its W tile (k = tap * Cin + ci)
        const int sl = kt / 9, tp = kt - sl * 9;
        return tp * p.Cin + sl * 64;
    };
    auto set_abase = [&](int (&ab)[2], int tp) {
        const int ky = tp / 3, kx = tp - 3 * ky;
        const int shift = ky * HWD + kx;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int q = q0[mi] + shift;
            ab[mi] = q * 128 + ((lhi ^ ((((l31 & 15) + kx) >> 1) & 7)) << 4);
        }
    };
    auto load = [&](V8 (&af)[2], V8 (&bf)[2], const int (&ab)[2], const char* cp, const char* cw, int kk) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const V8*>(cp + (ab[mi] ^ (kk << 5)));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int r = wn * 64 + ni * 32 + l31;
            bf[ni] = *reinterpret_cast<const V8*>(cw + r * 128 + swz(r, kk * 2 + lhi));
        }
    };
    auto mfmas = [&](const V8 (&af)[2], const V8 (&bf)[2]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Elt<T>::mfma(bf[ni], af[mi], acc[mi][ni]);
    };
    // counted wait: everything older than the `w` newest W k-tiles (4 pieces per wave each) and - if `pn` - the patch issued
    // among them (6 pieces) has landed; vmcnt retires in issue order, so the count is exact
    auto wait_newer = [&](int w, bool pn) {
        static_assert(DEEP_NW <= 6, "wait_newer covers up to five newer W k-tiles");
        switch (w * 2 + (pn ? 1 : 0)) {
            case 0: deep_wait_vm<0>(); break;
            case 1: deep_wait_vm<6>(); break;
            case 2: deep_wait_vm<4>(); break;
            case 3: deep_wait_vm<10>(); break;
            case 4: deep_wait_vm<8>(); break;
            case 5: deep_wait_vm<14>(); break;
            case 6: deep_wait_vm<12>(); break;
            case 7: deep_wait_vm<18>(); break;
            case 8: deep_wait_vm<16>(); break;
            case 9: deep_wait_vm<22>(); break;
            case 10: deep_wait_vm<20>(); break;
            default: deep_wait_vm<26>(); break;
        }
    };
    // ---- prologue: patch(0), W(0) .. W(NW - 1), patch(1) - the order every later slice repeats (its patch goes out behind the
    // W k-tile that is NW ahead of the previous slice's last k-tile)
    issue_patch(0, 0);
#pragma unroll
    for (int j = 0; j < DEEP_NW; ++j) issue_w(w_k0(j), j);          // nkt >= 9 > NW
    if (slices > 1) issue_patch(64, 1);
    wait_newer(DEEP_NW - 1, slices > 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int s = 0, tap = 0, wb = 0;   // slice, tap, W buffer of k-tile kt
    int abase[2];
    set_abase(abase, 0);
    const char* cP = smem;
    const char* cW = smem + DEEP_W0;
    V8 af[2][2], bf[2][2];
    load(af[0], bf[0], abase, cP, cW, 0);
    // ---- k-loop: the fragments of k-step kk + 1 are requested BEFORE the MFMAs of k-step kk (two register sets); at the tile's
    // last k-step every fragment of the tile is in registers, so its W buffer goes to the k-tile three ahead and the first
    // fragments of the next k-tile load under the last four MFMAs of this one.  ONE barrier per k-tile.
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                load(af[(kk + 1) & 1], bf[(kk + 1) & 1], abase, cP, cW, kk + 1);
            } else if (kt + 1 < nkt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of this k-tile has returned
                // W(kt + 1) has landed: newer = W(kt + 2) .. W(kt + NW - 1) and, at taps 0 .. NW - 2 of a slice that has a successor,
                // that successor's patch (issued behind W(first k-tile of the slice + NW - 1))
                wait_newer(max(0, min(DEEP_NW - 2, nkt - 2 - kt)), tap <= DEEP_NW - 2 && s + 1 < slices);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + DEEP_NW < nkt) issue_w(w_k0(kt + DEEP_NW), wb);   // the buffer this k-tile has just left
                const int ntap = tap == 8 ? 0 : tap + 1, ns = tap == 8 ? s + 1 : s;
                // entering slice ns: its successor's patch goes out now, behind W(first k-tile of ns + 2) - the prologue's order -,
                // into the buffer the slice just finished has left
                if (ntap == 0 && ns + 1 < slices) issue_patch((ns + 1) * 64, (ns + 1) & 1);
                tap = ntap;
                s = ns;
                wb = wb == DEEP_NW - 1 ? 0 : wb + 1;
                cP = smem + (s & 1) * (PROWS * 128);
                cW = smem + DEEP_W0 + wb * (BN * 128);
                set_abase(abase, tap);
                load(af[0], bf[0], abase, cP, cW, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // the reads go out FIRST
            mfmas(af[kk & 1], bf[kk & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    conv_halo_epilogue8<T>(p, acc, smem, tid, wm, wn, l31, lhi, b, oy0, ox0, n0, ty * tiles_x + tx);
}

constexpr int PH2 = 16;
constexpr int HROWS2 = (PH2 + 2) * HWD;   // 324 patch pixels
constexpr int NPIECE2 = (HROWS2 + 7) / 8;  // 41 LDS-DMA pieces of 8 rows
constexpr int PPW2 = (NPIECE2 + 3) / 4;    // 11 per wave
constexpr int PBYTES2 = PPW2 * 4 * 1024;   // 44 KiB

static_assert(256 * CS_STRIDE + 4096 <= PBYTES2 + 2 * BN * 128, "conv_halo256_kernel: staging tile + GroupNorm partials exceed its LDS");

template <int V>
struct IC2 {
    static constexpr int value = V;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_halo256_kernel(const GemmParams p) {
    typedef typename Elt<T>::v8 V8;
    typedef typename Elt<T>::v4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sP = smem;
    char* const sW = smem + PBYTES2;  // two buffers of BN * 128 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int lrow = lane >> 3, slot = lane & 7;
    const bool rowswz = (p.debug & 1024) != 0;  // A/B only: the row-index patch swizzle (2-way bank conflicts on every A read)

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    const int tiles_x = (p.Wout + PW - 1) / PW, tiles_y = (p.Hout + PH2 - 1) / PH2;
    const int tx = tm % tiles_x, ty = (tm / tiles_x) % tiles_y, b = tm / (tiles_x * tiles_y);
    const int oy0 = ty * PH2, ox0 = tx * PW, n0 = tn * BN;

    int poff[PPW2];  // element offset of the lane's chunk at channel 0; -1: zero page (halo / pad rows)
#pragma unroll
    for (int j = 0; j < PPW2; ++j) {
        const int q = (wave * PPW2 + j) * 8 + lrow;
        const int qy = q / HWD, qx = q - qy * HWD;
        const int uy = oy0 - 1 + qy, ux = ox0 - 1 + qx;
        const bool ok = (q < HROWS2) & (uy >= 0) & (uy < p.Hout) & (ux >= 0) & (ux < p.Wout);
        const int iy = p.upsample ? nearest_src(uy, p.up_sy, p.Hin) : uy, ix = p.upsample ? nearest_src(ux, p.up_sx, p.Win) : ux;
        const int chunk = slot ^ (((rowswz ? q : qx) >> 1) & 7);  // swizzle by patch COLUMN (see the header)
        poff[j] = ok ? ((b * p.Hin + iy) * p.Win + ix) * p.Cin + chunk * 8 : -1;
    }
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int n = min(n0 + row, p.N - 1);
        woff[j] = n * (int)p.ldw + chunk * 8;
    }
    // LDS-DMA from inline asm (`lds_dma16_v`, ds_common.h): behind the builtin the compiler turns every LDS wait of the kernel
    // into lgkmcnt(0), and the fragment read-ahead of the k-loop below would wait for the reads it has just issued
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    auto issue_patch = [&](int ci0) {
        const unsigned d = lds0 + wave * PPW2 * 1024;
#pragma unroll
        for (int j = 0; j < PPW2; ++j) {
            if (wave * PPW2 + j >= NPIECE2) continue;  // wave-uniform: the last wave owns fewer pieces
            const void* src = poff[j] >= 0 ? (const void*)(p.A + poff[j] + ci0) : (const void*)g_halo_zero_page;
            lds_dma16_v(src, d + j * 1024);
        }
    };
    auto issue_w = [&](int k0, int buf) {
        const unsigned d = lds0 + PBYTES2 + buf * (BN * 128) + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_v(p.W + k0 + woff[j], d + j * 1024);
    };

    char* const sC = smem;   // epilogue staging tile (over the patch / W buffers once the k-loop is done)
    auto body = [&](auto tailc) {
        // TAIL: the block's channel tile has at most 64 valid columns (Cout = 320: every third tile).  The four waves then split
        // the 256 pixels four ways over ONE 64-column strip (wave tile 64 x 64, 16 MFMAs per k-tile instead of 32) instead of
        // two of them computing 64 columns nobody stores: the tile costs half a tile, bit-identical values.
        constexpr bool TAIL = decltype(tailc)::value != 0;
        constexpr int MI = TAIL ? 2 : 4;
        const int rbase = TAIL ? wave * 64 : wm * 128, cbase = TAIL ? 0 : wn * 64;
        int q0[MI];  // GEMM row r = rbase + 32 mi + l31  <->  patch pixel (r >> 4, r & 15)
    #pragma unroll
        for (int mi = 0; mi < MI; ++mi) q0[mi] = ((rbase >> 4) + mi * 2 + (l31 >> 4)) * HWD + (l31 & 15);

        f32x16 acc[MI][2];
    #pragma unroll
        for (int i = 0; i < MI; ++i)
    #pragma unroll
            for (int j = 0; j < 2; ++j)
    #pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // ---- k-loop (round 5).  The fragments of k-step kk + 1 are requested BEFORE the MFMAs of k-step kk (two register sets),
        // so a read has 8 MFMAs (~260 cycles) to return; the k-tile's one barrier sits in front of its LAST k-step, where every
        // fragment read of the tile has returned - the W buffer is then free for the k-tile after next, and the first fragments
        // of the next k-tile load under the last 8 MFMAs of this one.  (Rounds 1-4: reads one or two MFMAs ahead of their use
        // and, because the staging went through __builtin_amdgcn_global_load_lds, every wait an lgkmcnt(0): a fragment round
        // trip exposed per four MFMAs.)
        const int slices = p.Cin / 64;
        const int nkt = slices * 9;
        auto set_abase = [&](int (&ab)[MI], int tp) {
            const int ky = tp / 3, kx = tp - 3 * ky;
            const int shift = ky * HWD + kx;
    #pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int q = q0[mi] + shift;
                ab[mi] = q * 128 + ((lhi ^ (((rowswz ? q : (l31 & 15) + kx) >> 1) & 7)) << 4);
            }
        };
        auto load = [&](V8 (&af)[MI], V8 (&bf)[2], const int (&ab)[MI], const char* cw, int kk) {
    #pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const V8*>(sP + (ab[mi] ^ (kk << 5)));
    #pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int r = cbase + ni * 32 + l31;
                bf[ni] = *reinterpret_cast<const V8*>(cw + r * 128 + swz(r, kk * 2 + lhi));
            }
        };
        auto mfmas = [&](const V8 (&af)[MI], const V8 (&bf)[2]) {
    #pragma unroll
            for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Elt<T>::mfma(bf[ni], af[mi], acc[mi][ni]);
        };
        auto landed_everywhere = [&]() {   // this wave's LDS-DMA pieces have landed, and so have everybody else's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto w_k0 = [&](int tp, int sl) { return tp * p.Cin + sl * 64; };

        issue_patch(0);
        issue_w(0, 0);
        landed_everywhere();
        if (nkt > 1) issue_w(w_k0(1, 0), 1);
        int s = 0, tap = 0;
        int abase[MI];
        set_abase(abase, 0);
        const char* cW = sW;
        V8 af[2][MI], bf[2][2];
        load(af[0], bf[0], abase, cW, 0);
        for (int kt = 0; kt + 1 < nkt; ++kt) {   // (the last k-tile is peeled: no conditional between the next tile's reads and
    #pragma unroll                              //  the MFMAs, so the compiler keeps its LDS waits counted there too)
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < 3) {
                    load(af[(kk + 1) & 1], bf[(kk + 1) & 1], abase, cW, kk + 1);
                } else {
                    const int ntap = tap == 8 ? 0 : tap + 1, ns = tap == 8 ? s + 1 : s;
                    const int ntap2 = ntap == 8 ? 0 : ntap + 1, ns2 = ntap == 8 ? ns + 1 : ns;
                    // every fragment read of this k-tile has returned (the last ones were requested a k-step ago)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ntap == 0) {
                        // new channel slice: the single patch buffer is free once every wave is here (the co-resident block
                        // computes while this one waits for the refill)
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        issue_patch(ns * 64);
                    }
                    landed_everywhere();                           // W(kt + 1) (and a new patch); nobody reads buffer kt & 1 any more
                    if (kt + 2 < nkt) issue_w(w_k0(ntap2, ns2), kt & 1);
                    tap = ntap;
                    s = ns;
                    cW = sW + ((kt + 1) & 1) * (BN * 128);
                    set_abase(abase, tap);
                    load(af[0], bf[0], abase, cW, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // the reads go out FIRST: left alone the scheduler sinks them between the MFMAs
                mfmas(af[kk & 1], bf[kk & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    #pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) load(af[(kk + 1) & 1], bf[(kk + 1) & 1], abase, cW, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(af[kk & 1], bf[kk & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();

        // ---- epilogue: as the 8 x 16 kernel, 256 rows - but without a branch (and the s_waitcnt vmcnt(0) the compiler puts
        // behind each conditional load) per 4 values: the bias + per-image bias of the lane's 8 column groups are fetched
        // once (clamped addresses: masked columns are never stored), and the residual rows are loaded four 16-byte pieces
        // at a time before they are needed.
        V4 b0[2][4], b1[2][4];
    #pragma unroll
        for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
            for (int g = 0; g < 4; ++g) b0[ni][g] = b1[ni][g] = V4{0, 0, 0, 0};
        if (p.bias) {
    #pragma unroll
            for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
                for (int g = 0; g < 4; ++g)
                    b0[ni][g] = *reinterpret_cast<const V4*>(p.bias + min(n0 + cbase + ni * 32 + 8 * g + 4 * lhi, p.N - 4));
        }
        if (p.rowbias) {
            const half_t* rb = p.rowbias + (long)b * p.rowbias_ld;
    #pragma unroll
            for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
                for (int g = 0; g < 4; ++g)
                    b1[ni][g] = *reinterpret_cast<const V4*>(rb + min(n0 + cbase + ni * 32 + 8 * g + 4 * lhi, p.N - 4));
        }
    #pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ms = rbase + mi * 32 + l31;
    #pragma unroll
            for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = cbase + ni * 32 + 8 * g + 4 * lhi;
                    V4 o;
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[mi][ni][4 * g + e];
                        v += (float)b0[ni][g][e];  // same order as the generic kernels: bias, then the per-image bias
                        v += (float)b1[ni][g][e];
                        o[e] = (T)v;
                    }
                    *reinterpret_cast<V4*>(sC + ms * CS_STRIDE + nl * 2) = o;
                }
        }
    };
    if (p.N - n0 <= 64 && (p.debug & 2048) == 0) body(IC2<1>{});   // gemm_debug bit 11: A/B, tail tiles as full tiles
    else body(IC2<0>{});
    __syncthreads();
    const T* const Rp = reinterpret_cast<const T*>(p.residual);
    float gs[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
#pragma unroll 1
    for (int j0 = 0; j0 < 16; j0 += 4) {
        long mrow[4];
        int ncol[4];
        bool ok[4];
        V8 rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + 256 * (j0 + u);
            const int row = id >> 4, c = id & 15;
            const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
            ok[u] = (n0 + c * 8 < p.N) & (oy < p.Hout) & (ox < p.Wout);
            mrow[u] = ((long)b * p.Hout + min(oy, p.Hout - 1)) * p.Wout + min(ox, p.Wout - 1);
            ncol[u] = min(n0 + c * 8, p.N - 8);
            if (Rp) rv[u] = *reinterpret_cast<const V8*>(Rp + mrow[u] * p.ldr + ncol[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = tid + 256 * (j0 + u);
            const int row = id >> 4, c = id & 15;
            V8 v = *reinterpret_cast<const V8*>(sC + row * CS_STRIDE + c * 16);
            if (Rp) {
                // f16: one v_pk_add_f16 per two values - the same number as (f16)((float)a + (float)b), which never rounds twice
                // (tests/test_f16_add_equivalence.py); bf16 (VAE decoder) keeps the f32 form
                if constexpr (std::is_same<T, half_t>::value) v = v + rv[u];
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (T)((float)v[e] + (float)rv[u][e]);
                }
            }
            if (ok[u]) *reinterpret_cast<V8*>(reinterpret_cast<T*>(p.C) + mrow[u] * p.ldc + ncol[u]) = v;
            if (p.gn_partial && ok[u]) gn_accumulate(v, gs, gq);
        }
    }
    if (p.gn_partial) gn_emit(p, gs, gq, smem + 256 * CS_STRIDE, tid, b, ty * tiles_x + tx, n0);
}

static thread_local int g_deep_max_blocks_per_cu = 1;   // ring-buffered 8 x 16 kernel for grids of <= this many blocks per CU (A/B: "conv_deep_blocks")
static thread_local int g_halo_variant = 0;  // 0 auto, 1 force the 8 x 16 kernel, 2 force the 16 x 16 kernel (where its shape rule holds), 3 force the ring-buffered 8 x 16 kernel, 4 never use it

}  // namespace

void ds_conv_halo_set_variant(int v) { g_halo_variant = v; }
void ds_conv_halo_set_deep_blocks(int v) { g_deep_max_blocks_per_cu = v; }

// Shapes the kernel takes: stride 1 (optionally the fused x2 upsample), Cin % 64 == 0, plain epilogue, an input small
// enough for 32-bit element offsets; any output height / width (edge patches are masked).
bool ds_conv_halo_applicable(const GemmParams& p) {
    if (!p.conv || p.cstride != 1 || p.epi != EPI_NONE || p.Hout <= 0 || p.Wout <= 0) return false;
    if (p.Cin % 64 != 0 || p.N % 8 != 0) return false;
    const long batch = p.M / ((long)p.Hout * p.Wout);
    return batch * p.Hin * p.Win * p.Cin < (1L << 31);
}

// 16 x 16 patches once they still fill the chip twice over (two blocks per CU): large batches / resolutions
static bool halo_big(const GemmParams& p) {
    const int batch = p.M / (p.Hout * p.Wout);
    const int ty16 = (p.Hout + PH2 - 1) / PH2, txs = (p.Wout + PW - 1) / PW;
    const long tiles256 = (long)batch * ty16 * txs * ((p.N + BN - 1) / BN);
    return g_halo_variant == 2 || ((g_halo_variant == 0 || g_halo_variant == 4) && tiles256 >= 1024);
}

// Pixel tiles per image of the variant ds_launch_conv_halo runs on this problem = partial-sum chunks the GroupNorm behind it
// has to add up; 0 when the statistics cannot come from the convolution (the GroupNorm workspace holds 128 chunks per image).
int ds_conv_halo_gn_chunks(const GemmParams& p) {
    if (!ds_conv_halo_applicable(p) || p.dtype != DS_DTYPE_F16) return 0;
    const int ph = halo_big(p) ? PH2 : PH;
    const long n = (long)((p.Hout + ph - 1) / ph) * ((p.Wout + PW - 1) / PW);
    return n <= 128 ? (int)n : 0;
}

int ds_launch_conv_halo(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    DS_REQUIRE(ds_conv_halo_applicable(p), "conv_halo: shape not supported");
    const int batch = p.M / (p.Hout * p.Wout);
    p.tiles_n = (p.N + BN - 1) / BN;
    const int ty8 = (p.Hout + PH - 1) / PH, ty16 = (p.Hout + PH2 - 1) / PH2, txs = (p.Wout + PW - 1) / PW;
    const bool big = halo_big(p);
    if (p.gn_partial) {
        p.gn_chunks = ds_conv_halo_gn_chunks(p);
        DS_REQUIRE(p.gn_chunks > 0, "conv_halo: GroupNorm statistics requested for a problem with more than 128 pixel tiles per image");
    }
    if (big) {
        p.tiles_m = batch * ty16 * txs;
        const size_t lds2 = PBYTES2 + 2 * BN * 128;  // 76 KiB; the 256 x 272 B epilogue tile (68 KiB) + 4 KiB of GroupNorm partials fit inside
        static unsigned long long attr_devs = 0;
        if (ds_first_on_device(attr_devs)) {
            DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo256_kernel<half_t>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo256_kernel<bf16_t>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        }
        dim3 grid2(p.tiles_m * p.tiles_n);
        if (p.dtype == DS_DTYPE_BF16) hipLaunchKernelGGL(conv_halo256_kernel<bf16_t>, grid2, dim3(256), lds2, stream, p);
        else hipLaunchKernelGGL(conv_halo256_kernel<half_t>, grid2, dim3(256), lds2, stream, p);
        DS_LAUNCH_CHECK();
        return 0;
    }
    p.tiles_m = batch * ty8 * txs;
    // Grids of at most one 8 x 16 block per CU (UNet batch 2 at the 1280-channel level: 160 blocks): the ring-buffered variant -
    // nothing else on the CU hides the single-buffer kernel's DMA round trip per k-tile.  f16 only (the bf16 VAE decoder never
    // has such a grid).  Same bits out.
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        DS_HIP(hipGetDevice(&dev));
        DS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus <= 0) cus = 256;
    }
    const long blocks8 = (long)p.tiles_m * p.tiles_n;
    if (p.dtype == DS_DTYPE_F16 && (g_halo_variant == 3 || (g_halo_variant == 0 && blocks8 <= g_deep_max_blocks_per_cu * (long)cus))) {
        static unsigned long long attr_devs_deep = 0;
        if (ds_first_on_device(attr_devs_deep))
            DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_deep_kernel<half_t>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEEP_LDS));
        hipLaunchKernelGGL(conv_halo_deep_kernel<half_t>, dim3(p.tiles_m * p.tiles_n), dim3(256), (size_t)DEEP_LDS, stream, p);
        DS_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = PROWS * 128 + BN * 128;  // 40 KiB; the 128 x 272 B epilogue tile (34 KiB) + 4 KiB of GroupNorm partials fit inside
    dim3 grid(p.tiles_m * p.tiles_n);
    if (p.dtype == DS_DTYPE_BF16) hipLaunchKernelGGL(conv_halo_kernel<bf16_t>, grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(conv_halo_kernel<half_t>, grid, dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

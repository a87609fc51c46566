// Attention cores for the DiffSensei UNet (head_dim 64, fp16 in / fp32 softmax / fp16 out).
//
//  self_attn_kernel   flash attention, replaces F.scaled_dot_product_attention at
//                     reference src/models/attention_processor.py:76-78 (AttnProcessor2_0)
//  ip_attn_kernel     ONE kernel for the whole MaskedIPAttnProcessor2_0 core, reference
//                     src/models/attention_processor.py:235-258: text SDPA (77 keys) + region-masked IP SDPA
//                     (16 dummy + 4x16 character keys) + `text + scale*ip`.  The additive mask the reference
//                     materialises as [B,heads,N,80] every layer of every step (:115-169, a Python double loop
//                     with a host sync per box) is evaluated analytically per query row from the 4 boxes.
//
// CDNA4 mapping (both kernels): a wave owns 32 query rows.  Scores are computed TRANSPOSED,
// S^T = K * Q^T with v_mfma_f32_32x32x16_f16, so lane (l&31) holds one query row's scores in registers:
// row max / row sum / the bbox test are lane-local (one cross-half exchange with lane^32).  The same
// registers, exponentiated and packed to f16, are directly the B operand of O^T = V^T * P^T; V is kept
// key-contiguous ("V transposed", produced that way by the projection GEMM) so its A fragments are
// plain 8-byte LDS reads.  O^T keeps the query row on the lane as well, so the online-softmax rescale is
// one multiply per accumulator register.
#include "ds_common.h"
#include "ds_kernels.h"

static thread_local int g_attn_variant = 0;  // 0 auto, 1 force 32 query rows per wave (QB = 1), 2 force 64 rows per wave (QB = 2)
void ds_attn_set_variant(int v) { g_attn_variant = v; }
static thread_local long g_ip_min_blocks = 1024;  // ip_attn: double the query tiles per block while the grid keeps this many blocks
void ds_ip_attn_set_min_blocks(int v) { g_ip_min_blocks = v; }
static thread_local int g_ip_variant = 0;  // ip_attn: 0 auto, 1 force the 4-wave register-staged kernel, 2 force the 8-wave LDS-DMA ring kernel
void ds_ip_attn_set_variant(int v) { g_ip_variant = v; }

namespace {
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ int swz(int row, int chunk) { return ((chunk ^ ((row >> 1) & 7)) << 4); }

// 2^x as ONE v_exp_f32.  exp2f() adds a denormal-range rescue (compare, select, ldexp: five more VALU slots per score);
// softmax arguments are <= 0 and anything below 2^-126 may flush to zero.  The softmax of the flash kernels is VALU
// bound (32 scores per lane per 16 MFMAs), so every slot per score shows up in the kernel time.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ h8 pack8(const f32x16& s, int base) {
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)s[base + e];
    return o;
}

// ------------------------------------------------------------------------------------------------
// Flash self-attention.  grid = (ceil(Nq/(128*QB)), B*heads), 256 threads, KV tile 64 keys, LDS 32 KiB.
// A wave owns QB blocks of 32 query rows; every K / V^T fragment read from LDS feeds QB MFMAs, so QB = 2 halves the
// LDS read traffic per flop (the limiter of the QB = 1 kernel: 1 KiB of ds_read per MFMA).
// ------------------------------------------------------------------------------------------------
template <int QB>
__global__ __launch_bounds__(256, 2) void self_attn_kernel(const SelfAttnParams p) {
    __shared__ __attribute__((aligned(16))) char sK[2][64 * 128];
    __shared__ __attribute__((aligned(16))) char sV[2][64 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
    const int q0 = blockIdx.x * (128 * QB) + wave * (32 * QB);

    // Q fragments (B operand of S^T): lane holds Q[q][kk*16 + lhi*8 .. +7]
    h8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + qb * 32 + l31, p.Nq - 1);
        const half_t* qp = p.q + (long)b * p.sq + (long)qrow * p.ldq + h * 64 + lhi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[qb][kk] = *reinterpret_cast<const h8*>(qp + kk * 16);
    }

    const half_t* kbase = p.k + (long)b * p.sk + h * 64;
    const half_t* vbase = p.vt + ((long)(b * p.heads + h) * 64) * p.ldv;
    const int c8 = tid & 7, r0 = tid >> 3;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    h8 rk[2], rv[2];
    auto load_tile = [&](int t) {
        const int key0 = t * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = r0 + 32 * j;
            const int key = key0 + row;
            rk[j] = key < p.Nk ? *reinterpret_cast<const h8*>(kbase + (long)key * p.ldk + c8 * 8) : zero8;
            const int kc = key0 + c8 * 8;  // row = d here
            rv[j] = kc < p.Nk ? *reinterpret_cast<const h8*>(vbase + (long)row * p.ldv + kc) : zero8;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = r0 + 32 * j;
            *reinterpret_cast<h8*>(&sK[buf][row * 128 + swz(row, c8)]) = rk[j];
            // V^T rows are stored with the four 4-key pieces of every 16-key group in the order 0 2 1 3: the P fragment
            // leaves the score MFMA holding keys {0-3, 8-11} (lanes 0-31) or {4-7, 12-15} (lanes 32-63) of a group, so
            // its V^T operand is then ONE 16-byte read (two 8-byte reads + four moves per fragment before)
            h4 lo, hi;
            lo[0] = rv[j][0]; lo[1] = rv[j][1]; lo[2] = rv[j][2]; lo[3] = rv[j][3];
            hi[0] = rv[j][4]; hi[1] = rv[j][5]; hi[2] = rv[j][6]; hi[3] = rv[j][7];
            const int g2 = c8 & ~1, odd8 = (c8 & 1) * 8;
            *reinterpret_cast<h4*>(&sV[buf][row * 128 + swz(row, g2) + odd8]) = lo;
            *reinterpret_cast<h4*>(&sV[buf][row * 128 + swz(row, g2 + 1) + odd8]) = hi;
        }
    };

    f32x16 ot[QB][2];
    float m_run[QB], l_part[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = NEG_BIG;
        l_part[qb] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[qb][d][r] = 0.f;
    }
    const float c = p.scale * LOG2E;

    const int nt = (p.Nk + 63) / 64;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        // ---- S^T = K Q^T  (two 32-key blocks per q-block; each K fragment feeds QB MFMAs)
        f32x16 st[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[qb][kb][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const h8 kf = *reinterpret_cast<const h8*>(&sK[buf][row * 128 + swz(row, kk * 2 + lhi)]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    st[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][kk], st[qb][kb], 0, 0, 0);
            }
        }
        const bool ragged = t * 64 + 64 > p.Nk;  // last tile: keys past Nk never contribute
        // ---- online softmax, query row = lane&31 (the other 16 keys of each block live on lane^32)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (ragged) {
                // keep this a (rarely taken) branch: if-converted or hoisted it costs up to 4 VALU slots per score
                int nk_here = p.Nk;
                asm volatile("" : "+s"(nk_here)::"memory");
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        if (key >= nk_here) st[qb][kb][r] = NEG_BIG;
                        asm volatile("" : "+v"(st[qb][kb][r]));  // pins the select inside the branch
                    }
            }
            float mloc = NEG_BIG;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[qb][kb][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run[qb], mloc);
            const float alpha = fast_exp2((m_run[qb] - m_new) * c);
            const bool moved = __builtin_amdgcn_ballot_w64(m_new != m_run[qb]) != 0;  // wave-uniform
            m_run[qb] = m_new;
            const float mc = m_new * c;
            // two scores per VALU slot where the ISA allows it: v_pk_fma_f32 for the scale-and-shift, v_pk_add_f32
            // for the row sum (the exponential itself has no packed form)
            f32x2 psum2 = {0.f, 0.f};
            const f32x2 c2 = {c, c}, mc2 = {-mc, -mc};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = {st[qb][kb][r], st[qb][kb][r + 1]};
                    v = __builtin_elementwise_fma(v, c2, mc2);
                    f32x2 e = {fast_exp2(v[0]), fast_exp2(v[1])};
                    st[qb][kb][r] = e[0];
                    st[qb][kb][r + 1] = e[1];
                    psum2 += e;
                }
            const float psum = psum2[0] + psum2[1];
            l_part[qb] = fmaf(l_part[qb], alpha, psum);
            if (moved) {  // once the running maxima have settled (a few tiles in) alpha == 1 on every lane
                asm volatile("" ::: "memory");
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[qb][d][r] *= alpha;
            }
        }
        // ---- O^T += V^T P^T  (each V^T fragment feeds QB MFMAs)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                h8 pf[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) pf[qb] = pack8(st[qb][kb], hb * 8);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int row = db * 32 + l31;
                    const h8 vf = *reinterpret_cast<const h8*>(&sV[buf][row * 128 + swz(row, kb * 4 + hb * 2 + lhi)]);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        ot[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb], ot[qb][db], 0, 0, 0);
                }
            }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l = l_part[qb] + __shfl_xor(l_part[qb], 32, 64);
        const float inv = 1.0f / l;
        const int qrow = q0 + qb * 32 + l31;
        // 16-byte stores (round 6; the store shape was measured on ip_attn_kernel below): a lane holds columns 8g + 4 lhi + {0..3}
        // of its query row, lane ^ 32 the other four of every group of eight; one v_permlane32_swap per dword and group pair
        // leaves the lower lane with columns 16j .. 16j+7, the upper one with 16j+8 .. 16j+15.  The swaps run on every lane
        // (clamped row address past Nq); only the stores are masked.
        half_t* op = p.o + (long)b * p.so + (long)min(qrow, p.Nq - 1) * p.ldo + h * 64 + lhi * 8;
        const bool row_ok = qrow < p.Nq;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            unsigned og[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[qb][db][4 * g + e] * inv);
                const u32x2_t t2 = __builtin_bit_cast(u32x2_t, o);
                og[g][0] = t2[0], og[g][1] = t2[1];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(og[2 * j][0]), "+v"(og[2 * j + 1][0]));
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(og[2 * j][1]), "+v"(og[2 * j + 1][1]));
                const u32x4_t v = {og[2 * j][0], og[2 * j][1], og[2 * j + 1][0], og[2 * j + 1][1]};
                if (row_ok) *reinterpret_cast<u32x4_t*>(op + db * 32 + j * 16) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Region test of the reference mask builder, evaluated per query token.
// reference src/models/attention_processor.py:145-163: grid = torch.linspace(0,1,W) x linspace(0,1,H)
// (inclusive end points), token idx -> (row idx / W, col idx % W); inside box k iff x1<=x<=x2 && y1<=y<=y2.
// torch.linspace (fp32): step = 1/(n-1); v[i] = i<n/2 ? i*step : 1 - (n-1-i)*step  (no fma).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace01(int i, int n) {
    if (n <= 1) return 0.f;
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? __fmul_rn(step, (float)i) : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}
// bit k set <=> token inside box k
__device__ __forceinline__ unsigned region_flags(const float* __restrict__ bbox_b, int max_ips, int idx, int mh,
                                                 int mw) {
    const int yi = idx / mw, xi = idx - yi * mw;
    const float x = linspace01(xi, mw), y = linspace01(yi, mh);
    unsigned f = 0;
    for (int k = 0; k < max_ips; ++k) {
        const float x1 = bbox_b[4 * k + 0], y1 = bbox_b[4 * k + 1], x2 = bbox_b[4 * k + 2], y2 = bbox_b[4 * k + 3];
        if (x >= x1 && x <= x2 && y >= y1 && y <= y2) f |= 1u << k;
    }
    return f;
}

__global__ void ip_region_flags_kernel(const float* bbox, uint8_t* flags, int B, int N, int max_ips, int mh, int mw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, idx = i - b * N;
    flags[i] = (uint8_t)region_flags(bbox + (long)b * max_ips * 4, max_ips, idx, mh, mw);
}

// ------------------------------------------------------------------------------------------------
// Fused text + region-masked IP cross-attention.  Keys padded to LP = 96 rows (three 32-key blocks).
// grid = (ceil(N/(128*qt)), B*heads); each block keeps the (b,h) text/IP K and V^T panels in LDS
// (4 x 12 KiB) and walks `qt` 128-row query tiles.
// ------------------------------------------------------------------------------------------------
constexpr int LP = 96;
constexpr int VSTR = 200;  // bytes per V^T row in LDS (96 keys * 2 B + 8 pad): conflict-free 8-byte reads

// (202 VGPRs -> two blocks per CU.  A build limited to 168 VGPRs for three blocks per CU spills 63 registers and is 20 %
// slower: 128 vs 106 us at B = 32, heads 20, N = 1024 - profiles/r02_ipattn_occupancy.txt.)
//
// RING (round 4; an experiment kept for A/B - ip_attn_variant 2 - and never chosen automatically: see ds_launch_ip_attn).
// Hypothesis: the kernel's arithmetic is tiny (192 keys per query row) and what it waits for is its own input and output - PMC
// (profiles/r03_pmc_conv_attn_ip_summary.txt): 45 % of the wave cycles in s_waitcnt, 2.2 TB/s; row-per-lane Q loads and O stores
// touch 32 cache lines per instruction for 32 bytes each (every 128-byte row slice took four load and eight store instructions),
// and Q was requested only one tile (~1.4 us) ahead of its use with two waves per SIMD.  Outcome: +6-9 % back to back, 0.35 ms
// per forward SLOWER inside the UNet forward - the kernel is VALU-bound (96 scores per lane and tile), not I/O-bound.
// RING = true: eight waves share the panels (one block per CU, same two waves per SIMD), and every wave owns a ring of three
// 4-KiB LDS slots: Q tiles arrive by LDS-DMA as WHOLE rows (8 lanes x 16 B per 128-byte row, 8 rows per instruction,
// chunk-swizzled on the source address), two tiles ahead, with counted vmcnt waits and no barrier (the slots are wave-private);
// the fragments are ds_read_b128.  The finished O tile is transposed through the slot its Q tile came from and leaves as whole
// rows as well (16 bytes per lane, 4 instructions instead of 8).  Same arithmetic in the same order: bit-identical to RING = false.
constexpr int IP_PANEL_BYTES = 2 * LP * 128 + 2 * 64 * VSTR;   // sKt | sKi | sVt | sVi
constexpr int IP_SLOT = 4096;                                  // one wave's 32 x 64 f16 tile
#ifndef IP_DIRECT_STORE
#define IP_DIRECT_STORE 1
#endif
template <int V>
struct IPC {
    static constexpr int value = V;
};

// T16 (round 6): both key counts lie in (64, 80] (the model's 77 text / 80 image keys), so keys 80..95 of the third 32-key block -
// registers 8..15 of its score block on every lane - are padding for every row: their scale / maximum / exponential /
// normalisation work, their V fragments and their P V MFMAs (probability exactly 0: bit-identical) are not issued at all -
// a sixth of the VALU work per score and of the P V MFMAs of a kernel whose time IS that work (header above).
template <int NW, bool RING, bool T16 = false>
__global__ __launch_bounds__(NW * 64, 2) void ip_attn_kernel(const IPAttnParams p, int qt) {
    extern __shared__ __attribute__((aligned(16))) char ip_smem[];   // ONE LDS object (a second one de-pipelines LDS-DMA waits)
    char* const sKt = ip_smem;
    char* const sKi = ip_smem + LP * 128;
    char* const sVt = ip_smem + 2 * LP * 128;
    char* const sVi = ip_smem + 2 * LP * 128 + 64 * VSTR;
    constexpr int ROWS = NW * 32;   // query rows per block and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
    char* const ring = ip_smem + IP_PANEL_BYTES + wave * (3 * IP_SLOT);   // RING only
    const int ntile = min(qt, (p.N - (int)blockIdx.x * qt * ROWS + ROWS - 1) / ROWS);   // tiles this block really has

    // Q fragments (B operand of S^T) are fetched one query tile ahead: the kernel runs two waves per SIMD, too few to
    // hide a global-load round trip inside the tile loop, and the first tile's rows are requested before the panels.
    h8 qn[4];
    auto load_q = [&](int it) {
        const int qrow = min((int)(blockIdx.x * qt + it) * ROWS + wave * 32 + l31, p.N - 1);
        const half_t* qp = p.q + ((long)b * p.N + qrow) * p.ldq + h * 64 + lhi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qn[kk] = *reinterpret_cast<const h8*>(qp + kk * 16);
    };
    // RING: tile `it` of this wave -> slot it % 3, four 1-KiB LDS-DMA pieces of 8 whole rows each; LDS row r keeps its 16-byte
    // chunk c at position c ^ ((r >> 1) & 7) (the swizzle is applied to the lane's SOURCE chunk, the DMA destination is linear)
    auto dma_q = [&](int it) {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        char* dst = ring + (it % 3) * IP_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = j * 8 + (lane >> 3);
            const int qrow = min((int)(blockIdx.x * qt + it) * ROWS + wave * 32 + r, p.N - 1);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            // (inline asm, round 5: behind the builtin the compiler turns every LDS wait of the kernel - the K / V panel fragment
            // reads of the tile loop - into lgkmcnt(0): ds_common.h, `lds_dma16`)
            lds_dma16_v(p.q + ((long)b * p.N + qrow) * p.ldq + h * 64 + chunk * 8, (unsigned)(size_t)(lds_void*)(dst + j * 1024));
        }
    };
    if constexpr (RING) {
        dma_q(0);
        if (ntile > 1) dma_q(1);
    } else {
        load_q(0);
    }
    // (the boxes are requested here, ahead of the panel loads, and broadcast behind the barrier below: one more round trip that
    // used to sit on its own between the staging and the first tile)
    float bv[4] = {2.f, 2.f, -1.f, -1.f};
    {
        const float* bbox_b = p.bbox + (long)b * p.max_ips * 4;
        if (lane < p.max_ips) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) bv[c4] = bbox_b[4 * lane + c4];
        }
    }
    // ---- stage the four panels once.  All 4 x ITER global loads are REQUESTED first, then the LDS writes (round 6): as two
    // rolled loops of {load, load, wait, write, write} the staging cost a block 6 memory round trips back to back - fitted from the
    // grid-rule sweep of profiles/r06_ipattn_microbench.txt (time = rounds x (S + tiles T)): S = 7-9 us per block against
    // T = 4 us per query tile, i.e. a fifth of the kernel at UNet batch 64 and most of it at batch 2 (one tile per block).
    {
        constexpr int ITER = (LP * 8 + NW * 64 - 1) / (NW * 64);   // LP * 8 == 64 * 12 == 768 sixteen-byte pieces per panel
        const half_t* ktp = p.kt + (long)b * p.sk + h * 64;
        const half_t* kip = p.ki + (long)b * p.sk + h * 64;
        const half_t* vtp = p.vtt + (long)b * p.sv + (long)(h * 64) * LP;
        const half_t* vip = p.vti + (long)b * p.sv + (long)(h * 64) * LP;
        h8 rk[ITER][2], rv[ITER][2];
#pragma unroll
        for (int j = 0; j < ITER; ++j) {
            const int id = min(tid + j * NW * 64, LP * 8 - 1);
            const int row = id >> 3, c = id & 7;
            rk[j][0] = *reinterpret_cast<const h8*>(ktp + (long)row * p.ldk + c * 8);
            rk[j][1] = *reinterpret_cast<const h8*>(kip + (long)row * p.ldk + c * 8);
            const int vrow = id / 12, vc = id - vrow * 12;
            rv[j][0] = *reinterpret_cast<const h8*>(vtp + (long)vrow * LP + vc * 8);
            rv[j][1] = *reinterpret_cast<const h8*>(vip + (long)vrow * LP + vc * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < ITER; ++j) {
            const int id = tid + j * NW * 64;
            if (id >= LP * 8) continue;
            const int row = id >> 3, c = id & 7;
            *reinterpret_cast<h8*>(&sKt[row * 128 + swz(row, c)]) = rk[j][0];
            *reinterpret_cast<h8*>(&sKi[row * 128 + swz(row, c)]) = rk[j][1];
            const int vrow = id / 12, vc = id - vrow * 12;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const h8 a = rv[j][t];
                char* const sV = t ? sVi : sVt;
                h4 lo, hi;
                lo[0] = a[0]; lo[1] = a[1]; lo[2] = a[2]; lo[3] = a[3];
                hi[0] = a[4]; hi[1] = a[5]; hi[2] = a[6]; hi[3] = a[7];
                *reinterpret_cast<h4*>(&sV[vrow * VSTR + vc * 16]) = lo;
                *reinterpret_cast<h4*>(&sV[vrow * VSTR + vc * 16 + 8]) = hi;
            }
        }
    }
    __syncthreads();

    // The character boxes of this image, fetched ONCE per block: lane k reads box k and the eight (x1, y1, x2, y2) are
    // broadcast into scalar registers.  (Read inside the tile loop, the short-circuited test became a chain of two
    // dependent global loads per box, each behind an s_waitcnt vmcnt(0) that also waited for the next tile's Q rows
    // requested just before it - every query tile stalled for whole memory round trips.)  Boxes >= max_ips: x1 = 2,
    // which no grid point (x in [0, 1]) satisfies.
    float box[8][4];
    {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) box[k][c4] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv[c4]), k));
    }
    const float ip_scale = p.ip_scale_ptr ? *p.ip_scale_ptr : p.ip_scale;
    for (int it = 0; it < qt; ++it) {
        const int q0 = (blockIdx.x * qt + it) * ROWS + wave * 32;
        if constexpr (RING) {
            if (it >= ntile) break;   // block-uniform (a wave whose rows lie past N still walks its - clamped - tiles: the counts below stay fixed)
        } else {
            if (q0 >= p.N) break;     // wave-uniform
        }
        const int qidx = min(q0 + l31, p.N - 1);
        h8 qf[4];
        if constexpr (RING) {
            // Vector-memory instructions this wave has issued behind tile it's four DMA pieces, in order: [4 O stores of tile
            // it-2] 4 pieces of tile it+1 (requested at the top of tile it-1) [4 O stores of tile it-1] - vmcnt retires in issue
            // order, so "all but those" is exact.  (Tile 0: only tile 1's pieces; the last tile has no successor's pieces.)
            const int newer = (it + 1 < ntile ? 4 : 0) + (it >= 1 ? 4 : 0) + (it >= 2 ? 4 : 0);
            if (newer >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (newer >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (newer >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* src = ring + (it % 3) * IP_SLOT;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const h8*>(src + l31 * 128 + swz(l31, kk * 2 + lhi));
            // tile it+2 goes into the slot tile it-1 used (its Q fragments and its O transposition are long retired)
            if (it + 2 < ntile) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_q(it + 2);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
            if (it + 1 < qt) load_q(it + 1);  // next tile's Q rows land under this tile's MFMAs / softmax
        }
        unsigned inside = 0;  // bit k set <=> the token lies in box k (region_flags() with the boxes in registers)
        {
            const int yi = qidx / p.mask_w, xi = qidx - yi * p.mask_w;
            const float x = linspace01(xi, p.mask_w), y = linspace01(yi, p.mask_h);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                inside |= (unsigned)((x >= box[k][0]) & (x <= box[k][2]) & (y >= box[k][1]) & (y <= box[k][3])) << k;
        }
        // 96-bit "attendable IP key" set of this query row: dummy keys [0, n_dummy) iff the token lies in NO box,
        // character k's keys [n_dummy + k*tpi, +tpi) iff it lies in box k  (reference :155-163)
        unsigned open_ip[3] = {0u, 0u, 0u};
        {
            auto set_range = [&](int lo, int hi) {  // bits [lo, hi) of the 96-bit set
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const int a = max(lo - 32 * w, 0), b = min(hi - 32 * w, 32);
                    if (b > a) open_ip[w] |= (b - a == 32 ? 0xffffffffu : ((1u << (b - a)) - 1u)) << a;
                }
            };
            if (inside == 0) set_range(0, p.n_dummy);
            for (int k = 0; k < p.max_ips; ++k)
                if ((inside >> k) & 1u) set_range(p.n_dummy + k * p.tok_per_ip, p.n_dummy + (k + 1) * p.tok_per_ip);
        }

        f32x16 ot[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;

        // A masked key's probability is exactly 0 (exp(-10000 + ...) underflows in fp32 as it does in the reference's
        // fp16), so a 32-key block of the IP panel in which NO query row of this wave has an attendable key contributes
        // nothing and is skipped whole (wave-uniform): the unconditional half of the CFG batch only ever attends the
        // dummy tokens, a panel with two characters never touches the fourth character's block.  Every row keeps at
        // least one attendable key in an active block as long as dummy tokens exist (no box -> dummies).
        bool act_ip[3];
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
            act_ip[kb] = p.n_dummy == 0 || __builtin_amdgcn_ballot_w64(open_ip[kb] != 0u) != 0;
        // Round 6: every LDS fragment of a phase is REQUESTED before the phase's first MFMA.  Left to the compiler the tile loop was
        // a chain of {ds_read, s_waitcnt lgkmcnt(0), v_mfma} triples - the registers of one fragment reused for the next - i.e. ~60
        // exposed LDS round trips per query tile with two waves per SIMD to hide them (ISA of the round-5 kernel; PMC: 47 % of the
        // wave cycles in s_waitcnt, profiles/r05_pmc_conv_ip_attn_summary.txt).  Now per part: the K fragments of the active key
        // blocks (up to 12 ds_read_b128) -> one wait -> the S^T MFMAs; the V^T fragments (up to 24 ds_read_b64 pairs) are requested
        // right behind them and land under the softmax's VALU work; the IP part's K fragments are requested before the text part's
        // P V MFMAs.  The scheduling barriers keep the compiler from sinking the reads back to their uses.  Same products in the
        // same order per accumulator: bit-identical to the round-5 kernel.
        // The loads are unconditional (a fragment of a skipped key block is read and never used: cheaper than a value that is
        // defined under one branch and used under another - the compiler cannot correlate the two and keeps ~100 registers alive).
        auto run_part = [&](auto partc) {
        h8 kf[3][4], vf[3][2][2];
        f32x16 st[3];
        auto load_k = [&](auto partc, auto kb0c, auto kb1c) {
            constexpr int part = decltype(partc)::value;
            const char* sK = part ? sKi : sKt;
#pragma unroll
            for (int kb = decltype(kb0c)::value; kb < decltype(kb1c)::value; ++kb) {
                const int row = kb * 32 + l31;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kf[kb][kk] = *reinterpret_cast<const h8*>(sK + row * 128 + swz(row, kk * 2 + lhi));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto qk = [&](auto partc) {
            constexpr int part = decltype(partc)::value;
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                if (kb == 0) load_k(partc, IPC<2>{}, IPC<3>{});   // the last block's fragments land under the first two blocks' MFMAs
                if (part && !act_ip[kb]) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][kk], qf[kk], st[kb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto load_v = [&](auto partc, auto kb0c, auto kb1c) {
            constexpr int part = decltype(partc)::value;
            const char* sV = part ? sVi : sVt;
#pragma unroll
            for (int kb = decltype(kb0c)::value; kb < decltype(kb1c)::value; ++kb) {
#pragma unroll
                for (int hb = 0; hb < ((T16 && kb == 2) ? 1 : 2); ++hb)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int row = db * 32 + l31;
                        const int koff = (kb * 32 + hb * 16) * 2 + 8 * lhi;  // byte offset of this lane's first 4 keys
                        const h4 v0 = *reinterpret_cast<const h4*>(sV + row * VSTR + koff);
                        const h4 v1 = *reinterpret_cast<const h4*>(sV + row * VSTR + koff + 16);
                        h8 v;
                        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
                        v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
                        vf[kb][hb][db] = v;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // scale + additive region mask (-10000, like the reference) + padding keys removed, then softmax.  The whole
        // core is VALU bound (192 scores per query row against 48 MFMAs), so the per-score work is kept to packed
        // fp32 ops: s = fma(raw, scale, bias) with a per-REGISTER bias.  Register r of key block kb is key
        // 32 kb + (r&3) + 8 (r>>2) + 4 lhi, i.e. 16-key group 2 kb + (r>>3) on every lane; when the dummy / per-
        // character token counts are multiples of 16 (the model's: 16 / 16) a group is open or closed as a whole
        // and six per-lane biases replace 96 bit tests.  Unmasked scores are bit-identical to raw*scale.
        // A masked key's probability is exactly 0 (see act_ip above).  The probabilities are left in st[], normalised and weighted.
        auto softmax = [&](auto partc) {
            constexpr int part = decltype(partc)::value;
            const int L = part ? p.Li : p.Lt;
            const bool grouped = (p.n_dummy % 16 == 0) && (p.tok_per_ip % 16 == 0) && L > 64;
            float mloc = NEG_BIG;
            if (grouped) {
                float gb[6];
#pragma unroll
                for (int g = 0; g < 6; ++g)
                    gb[g] = (part && !((open_ip[g >> 1] >> ((g & 1) * 16)) & 1u)) ? -10000.0f : 0.0f;
                const f32x2 sc2 = {p.qk_scale, p.qk_scale};
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    if (part && !act_ip[kb]) continue;
#pragma unroll
                    for (int r = 0; r < ((T16 && kb == 2) ? 8 : 16); r += 2) {
                        const float g = gb[kb * 2 + (r >> 3)];
                        f32x2 v = {st[kb][r], st[kb][r + 1]};
                        v = __builtin_elementwise_fma(v, sc2, (f32x2){g, g});
                        if (kb == 2) {  // only the last block holds padding keys (L > 64)
                            const int key = 64 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                            if (key >= L) v[0] = NEG_BIG;
                            if (key + 1 >= L) v[1] = NEG_BIG;
                        }
                        st[kb][r] = v[0];
                        st[kb][r + 1] = v[1];
                        mloc = fmaxf(mloc, fmaxf(v[0], v[1]));
                    }
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    if (part && !act_ip[kb]) continue;
#pragma unroll
                    for (int r = 0; r < ((T16 && kb == 2) ? 8 : 16); ++r) {
                        const int kbit = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const int key = kb * 32 + kbit;
                        float sv = fmaf(st[kb][r], p.qk_scale, (part && !((open_ip[kb] >> kbit) & 1u)) ? -10000.0f : 0.0f);
                        if (key >= L) sv = NEG_BIG;
                        st[kb][r] = sv;
                        mloc = fmaxf(mloc, sv);
                    }
                }
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            f32x2 psum2 = {0.f, 0.f};
            {
                const f32x2 l2 = {LOG2E, LOG2E}, m2 = {-mloc * LOG2E, -mloc * LOG2E};
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    if (part && !act_ip[kb]) continue;
#pragma unroll
                    for (int r = 0; r < ((T16 && kb == 2) ? 8 : 16); r += 2) {
                        f32x2 v = {st[kb][r], st[kb][r + 1]};
                        v = __builtin_elementwise_fma(v, l2, m2);
                        const f32x2 e = {fast_exp2(v[0]), fast_exp2(v[1])};
                        st[kb][r] = e[0];
                        st[kb][r + 1] = e[1];
                        psum2 += e;
                    }
                }
            }
            float psum = psum2[0] + psum2[1];
            psum += __shfl_xor(psum, 32, 64);
            const float w = (part ? ip_scale : 1.0f) / psum;
            {
                const f32x2 w2 = {w, w};
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    if (part && !act_ip[kb]) continue;
#pragma unroll
                    for (int r = 0; r < ((T16 && kb == 2) ? 8 : 16); r += 2) {
                        f32x2 v = {st[kb][r], st[kb][r + 1]};
                        v *= w2;
                        st[kb][r] = v[0];
                        st[kb][r + 1] = v[1];
                    }
                }
            }
        };
        // O^T += V^T P^T
        auto pv = [&](auto partc) {
            constexpr int part = decltype(partc)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                if (kb == 0) load_v(partc, IPC<(RING ? 0 : 1)>{}, IPC<3>{});   // land under the first block's four MFMAs
                if (part && !act_ip[kb]) continue;
#pragma unroll
                for (int hb = 0; hb < ((T16 && kb == 2) ? 1 : 2); ++hb) {
                    const h8 pf = pack8(st[kb], hb * 8);
#pragma unroll
                    for (int db = 0; db < 2; ++db) ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kb][hb][db], pf, ot[db], 0, 0, 0);
                }
            }
        };
        load_k(partc, IPC<0>{}, IPC<2>{});
        qk(partc);
        load_v(partc, IPC<0>{}, IPC<(RING ? 0 : 1)>{});   // (more of them ahead of the softmax spills: 244 registers as it is; the ring variant has none to spare)
        softmax(partc);
        pv(partc);
        };
        run_part(IPC<0>{});
        run_part(IPC<1>{});
        if constexpr (RING) {
            // O^T -> the tile's own (consumed) Q slot, rows of 128 bytes with the same chunk swizzle, -> whole rows out
            char* ep = ring + (it % 3) * IP_SLOT;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)ot[db][4 * g + e];
                    const int c = db * 32 + 8 * g + 4 * lhi;   // first of the lane's 4 columns
                    *reinterpret_cast<h4*>(ep + l31 * 128 + swz(l31, c >> 3) + ((c >> 2) & 1) * 8) = o;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), ch = lane & 7;
                const h8 v = *reinterpret_cast<const h8*>(ep + row * 128 + swz(row, ch));
                // unconditional (the launcher only picks this variant for N % 256 == 0): the counted waits above assume four
                // store instructions per tile and wave, and a store whose lanes are all masked off is branched around
                *reinterpret_cast<h8*>(p.o + ((long)b * p.N + q0 + row) * p.ldo + h * 64 + ch * 8) = v;
            }
        } else if constexpr (IP_DIRECT_STORE == 0) {   // build-time A/B: eight 8-byte stores per row (rounds 1-5)
            if (q0 + l31 < p.N) {
                half_t* op = p.o + ((long)b * p.N + q0 + l31) * p.ldo + h * 64;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        h4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (half_t)ot[db][4 * g + e];
                        *reinterpret_cast<h4*>(op + db * 32 + 8 * g + 4 * lhi) = o;
                    }
            }
        } else {
            // A lane holds columns 8g + 4 lhi + {0..3} of its query row; its partner lane ^ 32 the other four of every group of
            // eight.  One v_permlane32_swap per dword and group pair leaves the lower lane with columns 16j .. 16j+7 and the upper one
            // with 16j+8 .. 16j+15: 16-byte stores, four per wave and 32-dim block instead of eight 8-byte ones (the hand-over of
            // gemm_pp_kernel's GEGLU epilogue, cdna_hip_programming.md T21).  The swaps run on every lane (rows past N included);
            // only the stores are masked.
            half_t* op = p.o + ((long)b * p.N + min(q0 + l31, p.N - 1)) * p.ldo + h * 64 + lhi * 8;
            const bool row_ok = q0 + l31 < p.N;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                unsigned og[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)ot[db][4 * g + e];
                    const u32x2_t t = __builtin_bit_cast(u32x2_t, o);
                    og[g][0] = t[0], og[g][1] = t[1];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(og[2 * j][0]), "+v"(og[2 * j + 1][0]));
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(og[2 * j][1]), "+v"(og[2 * j + 1][1]));
                    const u32x4_t v = {og[2 * j][0], og[2 * j][1], og[2 * j + 1][0], og[2 * j + 1][1]};
                    if (row_ok) *reinterpret_cast<u32x4_t*>(op + db * 32 + j * 16) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic small attention (any head_dim <= 256, any token counts) for the once-per-panel encoders:
// CLIP ViT-H (257 tokens, head_dim 80), ViT-MAE, the perceiver Resampler and the MLLM's QwenResamplers (head_dim 160)
// (reference src/models/resampler.py:67-72: (q*s)(k*s)^T, fp32 softmax).  One wavefront per query row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_attn_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                         const half_t* __restrict__ v, half_t* __restrict__ o,
                                                         long ldq, long ldk, long ldv, long ldo, long sq, long sk,
                                                         long sv, long so, int B, int heads, int Nq, int Nk, int D,
                                                         float scale, int causal) {
    extern __shared__ float sp[];  // [4 waves][Nk]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    if (row >= Nq) return;
    float* pw = sp + (size_t)wave * Nk;
    const half_t* qp = q + b * sq + (long)row * ldq + h * D;
    float mx = NEG_BIG;
    for (int j = lane; j < Nk; j += 64) {
        const half_t* kp = k + b * sk + (long)j * ldk + h * D;
        float s = 0.f;
        for (int d = 0; d < D; d += 8) {
            const h8 a = *reinterpret_cast<const h8*>(qp + d);
            const h8 c = *reinterpret_cast<const h8*>(kp + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf((float)a[e], (float)c[e], s);
        }
        s *= scale;
        if (causal && j > row) s = NEG_BIG;  // CLIP text encoders: token t attends to tokens <= t
        pw[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Nk; j += 64) {
        const float e = __expf(pw[j] - mx);
        pw[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();  // same-wave LDS writes are ordered before its later reads
    const float inv = 1.0f / sum;
    for (int d = lane; d < D; d += 64) {
        float acc = 0.f;
        const half_t* vp = v + b * sv + h * D + d;
        for (int j = 0; j < Nk; ++j) acc = fmaf((float)(half_t)(pw[j] * inv), (float)vp[(long)j * ldv], acc);
        o[b * so + (long)row * ldo + h * D + d] = (half_t)acc;
    }
}

}  // namespace

// 0: self_attn_kernel<1>, 1: self_attn_kernel<2>, 2: self_attn_sp_kernel (see the measurements in ds_launch_self_attn)
static int self_attn_choice(int B, int heads, int Nq, int Nk) {
    const long blocks_sp = (long)((Nq + 255) / 256) * B * heads;
    if (g_attn_variant >= 3 || (g_attn_variant == 0 && blocks_sp >= 128)) return 2;
    if (g_attn_variant == 2 || (blocks_sp >= 512 && Nk >= 2048 && g_attn_variant != 1)) return 1;
    return 0;
}
const char* ds_self_attn_kernel_name(int B, int heads, int Nq, int Nk) {
    static const char* names[] = {"self_attn_kernel<1>", "self_attn_kernel<2>", "self_attn_sp_kernel"};
    return names[self_attn_choice(B, heads, Nq, Nk)];
}

int ds_launch_self_attn(const SelfAttnParams& p, hipStream_t stream) {
    DS_REQUIRE(p.B > 0 && p.heads > 0 && p.Nq > 0 && p.Nk > 0, "self_attn: empty problem");
    // any Nk: V^T rows are read 8 keys at a time, so their stride must cover Nk rounded up to 8 (the pad columns may hold
    // anything finite: keys >= Nk are masked to probability 0 on the last tile)
    DS_REQUIRE(p.ldv % 8 == 0 && p.ldv >= (p.Nk + 7) / 8 * 8 && p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldo % 4 == 0,
               "self_attn: ld* alignment (Nk=%d ldv=%ld)", p.Nk, p.ldv);
    // From 128 blocks of 256 query rows on: the software-pipelined kernel (attention_sp.hip).  Round 5 (no running maximum,
    // hand-scheduled step, counted LDS waits; profiles/r05_self_attn_sp_ab.txt, same box, us per launch, sp vs the 64-row flash
    // kernel): B = 64, N = 1024 448 vs 499; B = 64, N = 4096 2914 vs 3481; B = 8, N = 1024 67 vs 74; B = 4, N = 1024 (320 blocks)
    // 40.3 vs 47.9; B = 2, N = 1024 (160 blocks) 28.8 vs 30.5; B = 2, heads 20, N = 4096 226 vs 252; B = 2, N = 16384 1462 vs
    // 1633 - it wins at every shape of the UNet at every batch (rounds 3-4: only from ~1000 blocks on).  Smaller grids keep the
    // plain kernels: 64 query rows per wave when that still leaves >= 2 blocks per CU, 32 rows per wave otherwise.
    int choice = self_attn_choice(p.B, p.heads, p.Nq, p.Nk);
    // self_attn_sp_kernel forms its K / V^T source offsets in 32 bits (lds_dma16 takes a base + an unsigned byte offset):
    // a problem whose per-head K panel or 64 V^T rows span 4 GiB or more keeps the 64-bit-addressed flash kernel
    if (choice == 2 && !((long)p.Nk * p.ldk * 2 < (1L << 32) && 64L * p.ldv * 2 < (1L << 32))) choice = 1;
    switch (choice) {
        case 2: {
            SelfAttnParams q = p;
            q.xcd_map = g_attn_variant == 4 ? 0 : 1;   // 4: A/B only - the plain block order
            return ds_launch_self_attn_sp(q, stream);
        }
        case 1: hipLaunchKernelGGL(self_attn_kernel<2>, dim3((p.Nq + 255) / 256, p.B * p.heads), dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL(self_attn_kernel<1>, dim3((p.Nq + 127) / 128, p.B * p.heads), dim3(256), 0, stream, p); break;
    }
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_ip_attn(const IPAttnParams& p0, hipStream_t stream) {
    IPAttnParams p = p0;
    if (p.ldk == 0) p.ldk = p.C;
    if (p.sk == 0) p.sk = (long)LP * p.C;
    if (p.sv == 0) p.sv = (long)LP * p.C;
    DS_REQUIRE(p.ldk % 8 == 0 && p.sk % 8 == 0 && p.sv % 8 == 0, "ip_attn: key/value panel strides must be multiples of 8");
    DS_REQUIRE(p.B > 0 && p.heads > 0 && p.N > 0, "ip_attn: empty problem");
    DS_REQUIRE(p.LP == LP, "ip_attn: key panels must be padded to %d rows (got %d)", LP, p.LP);
    DS_REQUIRE(p.Lt > 0 && p.Lt <= LP && p.Li > 0 && p.Li <= LP, "ip_attn: Lt=%d Li=%d exceed %d", p.Lt, p.Li, LP);
    DS_REQUIRE(p.C == p.heads * 64, "ip_attn: head_dim must be 64 (C=%d heads=%d)", p.C, p.heads);
    DS_REQUIRE(p.max_ips >= 1 && p.max_ips <= 8 && p.tok_per_ip > 0, "ip_attn: bad ip token layout");
    DS_REQUIRE(p.n_dummy + p.max_ips * p.tok_per_ip == p.Li, "ip_attn: Li (%d) != n_dummy + max_ips*tok_per_ip", p.Li);
    DS_REQUIRE(p.mask_h * p.mask_w == p.N, "ip_attn: mask grid %dx%d != N %d", p.mask_h, p.mask_w, p.N);
    DS_REQUIRE(p.ldq % 8 == 0 && p.ldo % 8 == 0, "ip_attn: ldq (%ld) and ldo (%ld) must be multiples of 8 (16-byte row pieces)", (long)p.ldq, (long)p.ldo);
    // walk several query tiles per block once there are plenty of blocks (amortises the K/V panel staging)
    // The 8-wave LDS-DMA ring variant, 256 query rows per block and tile: g_ip_variant 2 only (see below).
    {
        const int tiles8 = (p.N + 255) / 256;
        int qt8 = 1;
        while (qt8 < 8 && qt8 * 2 <= tiles8 && (long)((tiles8 + 2 * qt8 - 1) / (2 * qt8)) * p.B * p.heads >= g_ip_min_blocks) qt8 *= 2;
        const long blocks8 = (long)((tiles8 + qt8 - 1) / qt8) * p.B * p.heads;
        // Measured (profiles/r04_ipattn_ring_ab.txt): back to back on its own, B = 64 N = 1024 163 -> 154 us, N = 4096 269 -> 246 us (four
        // and eight tiles per block; with two tiles per block the ring has nothing to run ahead of and the larger block loses
        // 15-20 %) - but INSIDE the UNet forward, where Q has just been written by the to_q GEMM, the same launches take 0.35 ms
        // per forward LONGER than the register-staged kernel (in-situ A/B, three interleaved rounds; the rocprofv3 trace of a
        // whole call agrees: 214 vs 188 us per launch).  The kernel is VALU-bound, not bound by its loads and stores: the
        // variant is kept for A/B (ip_attn_variant 2, bit-identical) and never chosen automatically.
        (void)blocks8;
        if (p.N % 256 == 0 && p.ldo % 8 == 0 && p.ldq % 8 == 0 && g_ip_variant == 2) {
            const size_t lds = IP_PANEL_BYTES + 8 * 3 * IP_SLOT;
            auto kern = ip_attn_kernel<8, true>;
            static unsigned long long attr_devs = 0;
            if (ds_first_on_device(attr_devs))
                DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((tiles8 + qt8 - 1) / qt8, p.B * p.heads), dim3(512), lds, stream, p, qt8);
            DS_LAUNCH_CHECK();
            return 0;
        }
    }
    const int tiles = (p.N + 127) / 128;
    int qt = 1;
    while (qt < 8 && (long)((tiles + 2 * qt - 1) / (2 * qt)) * p.B * p.heads >= g_ip_min_blocks) qt *= 2;
    dim3 grid((tiles + qt - 1) / qt, p.B * p.heads);
    // keys 80..95 are padding for every row of both panels (the model's 77 / 80 keys): the T16 instantiation (ip_attn_variant 3: off, A/B)
    const bool t16 = g_ip_variant != 3 && p.Lt > 64 && p.Lt <= 80 && p.Li > 64 && p.Li <= 80;
    if (t16) hipLaunchKernelGGL((ip_attn_kernel<4, false, true>), grid, dim3(256), (size_t)IP_PANEL_BYTES, stream, p, qt);
    else hipLaunchKernelGGL((ip_attn_kernel<4, false, false>), grid, dim3(256), (size_t)IP_PANEL_BYTES, stream, p, qt);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_ip_region_flags(const float* bbox, uint8_t* flags, int B, int N, int max_ips, int mask_h, int mask_w,
                              hipStream_t stream) {
    DS_REQUIRE(mask_h * mask_w == N, "region_flags: grid %dx%d != N %d", mask_h, mask_w, N);
    hipLaunchKernelGGL(ip_region_flags_kernel, dim3((B * N + 255) / 256), dim3(256), 0, stream, bbox, flags, B, N,
                       max_ips, mask_h, mask_w);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_small_attn(const half_t* q, const half_t* k, const half_t* v, half_t* o, long ldq, long ldk, long ldv,
                         long ldo, long sq, long sk, long sv, long so, int B, int heads, int Nq, int Nk, int D,
                         float scale, hipStream_t stream, int causal) {
    DS_REQUIRE(D % 8 == 0 && D <= 256, "small_attn: head_dim %d unsupported", D);
    DS_REQUIRE(Nk * 4 * sizeof(float) <= 60000, "small_attn: Nk %d too large", Nk);
    dim3 grid((Nq + 3) / 4, B * heads);
    hipLaunchKernelGGL(small_attn_kernel, grid, dim3(256), (size_t)4 * Nk * sizeof(float), stream, q, k, v, o, ldq,
                       ldk, ldv, ldo, sq, sk, sv, so, B, heads, Nq, Nk, D, scale, causal);
    DS_LAUNCH_CHECK();
    return 0;
}

// Software-pipelined flash self-attention (head_dim 64, fp16 in / fp32 softmax / fp16 out): the large-problem variant of
// ds_launch_self_attn, replaces F.scaled_dot_product_attention at reference src/models/attention_processor.py:76-78.
//
// Why a second kernel.  self_attn_kernel (attention.hip) runs QK^T -> softmax -> PV strictly one after the other inside a
// wave: 32 MFMAs, then ~230 VALU instructions with the matrix pipe idle (PMC, profiles/r02_pmc_conv_attn_summary.txt: MFMA
// busy 0.41, 11 VALU per MFMA).  Here a wave still owns two 32-row query blocks, but the work is skewed by one (tile,
// query block) PAIR: while the VALU exponentiates the scores of pair k, the matrix pipe computes the scores of pair k+1 and
// the P V product of pair k-1 - 16 MFMAs beside ~100 VALU instructions in ONE scheduling region, pinned together with
// sched_group_barrier.  What it takes:
//   * no branch inside a step: the online-softmax rescale is deferred.  Scores leave the MFMA already shifted - the
//     accumulator input of S^T = K Q^T is a register block holding -m_ref of the lane's query row - and in base 2 (Q is
//     pre-multiplied by scale * log2 e once), so a probability is ONE v_exp_f32.
//   * NO running maximum in the steady state (round 5).  m_ref is set from the first tile's maximum and afterwards only raised
//     when it has to be: a step sums the f16 probabilities it has just packed (v_dot2c_f32_f16 against (1, 1): the row sum the
//     normalisation needs anyway, taken from the very values the P V MFMA multiplies), and only when some lane's 32-key partial
//     sum exceeds LIM = 2^14 (every p is then still < 2^14 << 65504) does a rare wave-uniform branch BEHIND the step take the
//     tile maximum, raise m_ref, rescale O and l and redo the step's probabilities from the scores, which are still intact.
//     Rounds 3-4 took every tile's maximum inside the step (16 v_max3 + a cross-half exchange + a ballot per step, and the
//     maximum had to wait for the step's own Q K^T MFMAs to finish - the one place the VALU stalled on the matrix pipe).
//     f16 keeps 11 bits whatever a probability's magnitude, l and O accumulate in f32: the result is as accurate as with p <= 1.
//   * K and V^T tiles arrive by LDS-DMA (global_load_lds, swizzle on the source address) into a ring of three buffers each:
//     no staging registers, no VALU for staging.  V^T is stored row-major exactly as it sits in HBM: the score MFMA's row
//     order is permuted instead (tile row i holds key pi(i), pi = swap bits 2 and 3), after which the eight f16 a lane packs
//     for one P V k-step are eight CONSECUTIVE keys - one 16-byte LDS read of V^T, as in self_attn_kernel, but without its
//     store-time shuffle.
// Not bit-identical to self_attn_kernel (deferred rescale, pre-scaled Q): compared with the fp32 reference and the golden
// fixture at the same tolerance (tests/test_gpu_ops.py).
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int TILE_B = 8192;      // one 64 x 64 f16 tile
constexpr int V_BASE = 3 * TILE_B;
constexpr float LIM = 16384.0f;   // a lane's 32-key partial row sum above this re-centres the row (p < 2^14 always)
constexpr float NEG_BIG = -1.0e30f;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_sp __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_sp __attribute__((ext_vector_type(4)));

// Debug counter (ds_debug_counter("attn_sp_recentre")): how often the rare re-centring branch behind a step ran, per wave and
// pair, since the last reset.  One atomic from one lane inside the rare branch - nothing in the steady state.  It exists so that
// a model-level test can PROVE its inputs drove the branch (tests/test_gpu_outlier_magnitudes.py).
__device__ unsigned long long g_sp_recentre_count = 0;

// every LDS-DMA this wave has issued has landed, every wave of the block is here, and nothing moves across the point
#define SP_SYNC()                                              \
    do {                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

__device__ __forceinline__ int pi32(int i) {  // swap bits 2 and 3 of a 5-bit index
    return (i & 19) | ((i & 4) << 1) | ((i & 8) >> 1);
}

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max over both half-waves of a per-lane value (lanes l and l^32 hold the two halves of one query row's scores)
__device__ __forceinline__ float cross_max(float x) {
    // v_permlane32_swap vdst, src exchanges vdst[32..63] with src[0..31]: with both operands = x, vdst ends up holding the
    // lower half's value in every lane and src the upper half's.  Inline asm on purpose: through
    // __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) folds fmaxf(result[0], result[1]) to result[0] - the second result
    // is never even extracted in the IR - and every lane then only saw the lower half's maximum (found on the GPU: rows
    // whose maximum sat in an upper-half key were not re-centred and overflowed f16).  The two v_nop are the wait states the
    // instruction needs after a VALU write of its operands.
    float a = x, b = x;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

__global__ __launch_bounds__(256, 2) void self_attn_sp_kernel(const SelfAttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE_B];  // K ring (3 tiles) | V^T ring (3 tiles): ONE object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // Block -> (head, query block): the dispatcher places block i on XCD i % 8.  With the plain (query block, head) grid the 16
    // query blocks of one head were spread over all eight XCDs and every one of them pulled that head's K / V^T through its own
    // L2: 2.85 GB of fabric reads per launch for 0.17 GB of K + V (PMC, profiles/r03_pmc_conv_attn_ip_summary.txt).  The 1-D
    // grid is remapped so that an XCD owns a contiguous run of work items = all query blocks of a few heads, back to back.
    const int nqb = (p.Nq + 255) / 256;
    const int item = p.xcd_map ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int bh = item / nqb, qblk = item - bh * nqb;
    const int b = bh / p.heads, h = bh % p.heads;
    const int q0 = qblk * 256 + wave * 64;
    const int nt = (p.Nk + 63) / 64;
    const bool ragged = (p.Nk & 63) != 0;

    // ---- LDS-DMA staging: per tile and operand two 4-KiB instructions; wave w, piece j covers tile rows (4 j + w) * 8 .. + 7
    const half_t* const kbase = p.k + (long)b * p.sk + h * 64;
    const half_t* const vbase = p.vt + ((long)(b * p.heads + h) * 64) * p.ldv;
    // (the lane's staging row / chunk are rebuilt from an opaque lane id inside every issue: hoisted out of the tile loop they
    // are eight registers this kernel does not have - they went to scratch, and a scratch reload is a vmcnt(0) round trip per tile)
    auto lane_now = [&]() -> int {
        unsigned z = 0;
        asm volatile("" : "+v"(z));
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    };
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;   // LDS byte address of the rings
    auto issue_k = [&](int t, int buf) {
        const int ln = lane_now(), srow = ln >> 3, sslot = ln & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (4 * j + wave) * 8 + srow;
            const int key = min(t * 64 + row, p.Nk - 1);                 // rows past Nk re-read the last key (masked later)
            const int chunk = sslot ^ ((row >> 1) & 7);
            lds_dma16(kbase, (unsigned)(key * (int)p.ldk + chunk * 8) * 2u, lds0 + buf * TILE_B + (4 * j + wave) * 1024);
        }
    };
    auto issue_v = [&](int t, int buf) {
        const int ln = lane_now(), srow = ln >> 3, sslot = ln & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (4 * j + wave) * 8 + srow;                   // d
            const int chunk = sslot ^ ((row >> 1) & 7);
            const int kc = min(t * 64 + chunk * 8, ((p.Nk - 1) >> 3) << 3);  // chunks past Nk re-read the last one (P = 0 there)
            lds_dma16(vbase, (unsigned)(row * (int)p.ldv + kc) * 2u, lds0 + V_BASE + buf * TILE_B + (4 * j + wave) * 1024);
        }
    };

    // ---- prologue, part 1: K(0), V(0), K(1) are requested BEFORE the Q rows, so the two round trips overlap
    issue_k(0, 0);
    issue_v(0, 0);
    if (nt > 1) issue_k(1, 1);

    // ---- Q fragments (B operand of S^T), pre-multiplied by scale * log2(e): scores arrive as base-2 logits
    h8 qf[2][4];
    {
        const float c = p.scale * LOG2E;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = min(q0 + qb * 32 + l31, p.Nq - 1);
            const half_t* qp = p.q + (long)b * p.sq + (long)qrow * p.ldq + h * 64 + lhi * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const h8 v = *reinterpret_cast<const h8*>(qp + kk * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[qb][kk][e] = (half_t)((float)v[e] * c);
            }
        }
    }

    // ---- fragment addresses (bytes into smem, without the ring-buffer base)
    unsigned koff[4], voff[4];
    {
        const int krow = pi32(l31), ksw = (krow >> 1) & 7, vsw = (l31 >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) koff[kk] = krow * 128 + (((kk * 2 + lhi) ^ ksw) << 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) voff[c] = V_BASE + l31 * 128 + (((c * 2 + lhi) ^ vsw) << 4);
    }

    f32x16 O[2][2], S0[2], S1[2];
    h8 P0[2][2], P1[2][2];
    float mref[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[qb][0][r] = O[qb][1][r] = 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int e = 0; e < 8; ++e) P1[kb][hb][e] = (half_t)0.f;

    // S^T block of one pair: S[kb] = K(kb) Q'^T - m_ref
    auto qk = [&](f32x16 (&S)[2], int qb, unsigned kb_base) {
        f32x16 negm;  // the accumulator input: -m_ref of the lane's query row in all 16 registers (one block, rebuilt per step)
        const f32x2 nm2 = {-mref[qb], -mref[qb]};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {   // pairs: v_pk_mov_b32 moves two registers per instruction
            negm[r] = nm2[0];
            negm[r + 1] = nm2[1];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const h8 kf = *reinterpret_cast<const h8*>(smem + kb_base + koff[kk] + kb * 4096);
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][kk], kk == 0 ? negm : S[kb], 0, 0, 0);
            }
    };
    // O^T(qb) += V^T P^T
    auto pv = [&](const h8 (&P)[2][2], int qb, unsigned vb_base) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const h8 vf = *reinterpret_cast<const h8*>(smem + vb_base + voff[kb * 2 + hb] + db * 4096);
                    O[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, P[kb][hb], O[qb][db], 0, 0, 0);
                }
    };
    auto tile_max = [&](const f32x16 (&S)[2]) -> float {
        float m = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(m, S[kb][r]), S[kb][r + 1]);
        return m;
    };
    // probabilities of one pair from its (shifted, base-2) scores: P = f16(exp2(S)), and the lane's partial row sum of the
    // PACKED values (v_dot2c_f32_f16 against (1, 1)).  S is consumed: nothing reads it afterwards, its registers free up as the
    // step proceeds (the rare re-centring behind a step recomputes the scores from K, which is still in its ring slot).
    auto softmax = [&](const f32x16 (&S)[2], h8 (&P)[2][2]) -> float {
        float sum[2] = {0.f, 0.f};   // two chains: the dot accumulates in place, one chain would serialise 16 of them
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const h2v pr = {(half_t)fexp2(S[kb][hb * 8 + e]), (half_t)fexp2(S[kb][hb * 8 + e + 1])};
                    P[kb][hb][e] = pr[0];
                    P[kb][hb][e + 1] = pr[1];
                    sum[hb] = __builtin_amdgcn_fdot2(pr, h2v{(half_t)1.f, (half_t)1.f}, sum[hb], false);
                }
            }
        return sum[0] + sum[1];
    };
    auto mask_ragged = [&](f32x16 (&S)[2], int t) {   // keys >= Nk of the last tile
        int nk_here = p.Nk;
        asm volatile("" : "+s"(nk_here)::"memory");
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + kb * 32 + pi32((r & 3) + 8 * (r >> 2) + 4 * lhi);
                if (key >= nk_here) S[kb][r] = NEG_BIG;
            }
    };
    // rare work at the top of a step (outside the scheduled region): the ragged last tile masks keys, the first tile sets m_ref
    auto prepare = [&](f32x16 (&S)[2], int qb, int t) {
        if (ragged && t == nt - 1) mask_ragged(S, t);
        if (t == 0) {   // scores arrived unshifted (m_ref = 0): centre the row on its first tile's maximum, whatever its sign
            const float mx = cross_max(tile_max(S));
            mref[qb] = mx;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[kb][r] -= mx;
        }
    };
    // rare work BEHIND a step: some lane's partial row sum outgrew LIM (or overflowed to inf).  The pair's scores are computed
    // again (K(t) has not left its ring slot), the rows whose tile maximum lies above m_ref are re-centred on it, what they have
    // accumulated is rescaled and the pair's probabilities are redone.  O[qb] holds pairs before (t, qb) only (this pair's P V
    // runs in the next step), so nothing of the redone pair is in it yet.
    auto recentre = [&](f32x16 (&S)[2], h8 (&P)[2][2], int qb, unsigned kc_base, int t, float& sum) {
        if (lane_now() == 0) atomicAdd(&g_sp_recentre_count, 1ull);
        qk(S, qb, kc_base);
        if (ragged && t == nt - 1) mask_ragged(S, t);
        const float delta = fmaxf(cross_max(tile_max(S)), 0.f);
        const float alpha = fexp2(-delta);
        mref[qb] += delta;
        lsum[qb] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kb][r] -= delta;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[qb][db][r] *= alpha;
        sum = softmax(S, P);
    };
    // One step: softmax of the CURRENT pair on the VALU beside the MFMAs of the NEXT pair's scores and the PREVIOUS pair's P V
    // (kc_base: the ring slot of the current pair's own K tile, for `recentre`).
    // Written out as 16 groups of {1 MFMA, the fragment read for the MFMA four groups ahead, 2 v_exp_f32, 1 pack, 1 dot}, each
    // closed by a sched_barrier: the order below IS the issue order.  Every fragment is requested four MFMAs (~130 cycles: an
    // LDS round trip) before the MFMA that consumes it, through a ring of four fragment registers, and is waited for with a
    // COUNTED lgkmcnt(3) - which the compiler only emits because the LDS-DMA pieces of this kernel are issued from inline asm
    // (`lds_dma16`, ds_common.h): behind a __builtin_amdgcn_global_load_lds it assumes a FLAT access that may bump lgkmcnt
    // and turns every LDS wait of the kernel into lgkmcnt(0).  (Rounds 3-4 described the same
    // mix to the scheduler with sched_group_barrier and let it place the reads: at 256 registers it put each read directly in
    // front of its MFMA, so half of the MFMAs of a step sat behind an s_waitcnt lgkmcnt(0) for their own fragment - the matrix pipe
    // measured 0.44 busy with a VALU load that explains 0.7.)  MFMA order: the two score blocks alternate (two independent
    // accumulator chains), then P V with the two output blocks alternating.
    auto step = [&](f32x16 (&Sc)[2], h8 (&Pc)[2][2], int qbc, unsigned kc_base, f32x16 (&Sn)[2], unsigned kb_base,
                    const h8 (&Pp)[2][2], unsigned vb_base, int t) {
        const int qbn = qbc ^ 1;
        auto frag = [&](int g) -> h8 {   // g: compile-time after unrolling
            if (g < 8) return *reinterpret_cast<const h8*>(smem + kb_base + koff[g >> 1] + (g & 1) * 4096);
            return *reinterpret_cast<const h8*>(smem + vb_base + voff[(g - 8) >> 1] + ((g - 8) & 1) * 4096);
        };
        h8 fr[4];   // ring of four fragment registers
#pragma unroll
        for (int g = 0; g < 4; ++g) fr[g] = frag(g);
        f32x16 negm;  // the accumulator input of the score MFMAs: -m_ref of the lane's query row in all 16 registers
        {
            const f32x2 nm2 = {-mref[qbn], -mref[qbn]};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                negm[r] = nm2[0];
                negm[r + 1] = nm2[1];
            }
        }
        float sum[2] = {0.f, 0.f};   // two chains: the dot accumulates in place
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g < 8) {
                const int kb = g & 1, kk = g >> 1;
                Sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[g & 3], qf[qbn][kk], kk == 0 ? negm : Sn[kb], 0, 0, 0);
            } else {
                const int c = (g - 8) >> 1, db = (g - 8) & 1;
                O[qbn][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[g & 3], Pp[c >> 1][c & 1], O[qbn][db], 0, 0, 0);
            }
            if (g + 4 < 16) fr[g & 3] = frag(g + 4);
            {   // probabilities of score pair g of the current pair (S is consumed: its registers free up as the step proceeds)
                const int kb = g >> 3, hb = (g >> 2) & 1, e = (g & 3) * 2;
                const h2v pr = {(half_t)fexp2(Sc[kb][hb * 8 + e]), (half_t)fexp2(Sc[kb][hb * 8 + e + 1])};
                Pc[kb][hb][e] = pr[0];
                Pc[kb][hb][e + 1] = pr[1];
                sum[g & 1] = __builtin_amdgcn_fdot2(pr, h2v{(half_t)1.f, (half_t)1.f}, sum[g & 1], false);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float rsum = sum[0] + sum[1];
        if (__builtin_amdgcn_ballot_w64(!(rsum <= LIM)) != 0)    // (!(<=): catches inf and NaN too)
            recentre(Sc, Pc, qbc, kc_base, t, rsum);
        lsum[qbc] += rsum;
    };

    // ---- prologue, part 2: scores of pair (0, qb 0)
    SP_SYNC();
    qk(S0, 0, 0u);

    int kcur = 0, knext = nt > 1 ? 1 : 0, vprev = 0, vcur = 0;  // ring-buffer indices of K(t), K(t+1), V(t-1), V(t)
    for (int t = 0; t < nt; ++t) {
        // K(t+1) and V(t) were requested a whole iteration ago; everybody is done with K(t-1) and V(t-2)
        SP_SYNC();
        if (t + 2 < nt) issue_k(t + 2, (t + 2) % 3);
        if (t + 1 < nt) issue_v(t + 1, (t + 1) % 3);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- step A: softmax(t, qb 0) | scores(t, qb 1) | P V (t-1, qb 1)
        prepare(S0, 0, t);
        step(S0, P0, 0, (unsigned)kcur * TILE_B, S1, (unsigned)kcur * TILE_B, P1, (unsigned)vprev * TILE_B, t);
        // ---- step B: softmax(t, qb 1) | scores(t+1, qb 0) | P V (t, qb 0)
        prepare(S1, 1, t);
        step(S1, P1, 1, (unsigned)kcur * TILE_B, S0, (unsigned)knext * TILE_B, P0, (unsigned)vcur * TILE_B, t);
        vprev = vcur;
        vcur = vcur == 2 ? 0 : vcur + 1;
        kcur = knext;
        if (t + 2 < nt) knext = knext == 2 ? 0 : knext + 1;   // past the last tile: re-score K(nt-1), result unused
    }
    pv(P1, 1, (unsigned)vprev * TILE_B);   // the last pair's P V

    // ---- epilogue: O / l
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = lsum[qb] + __shfl_xor(lsum[qb], 32, 64);
        const float inv = 1.0f / l;
        const int qrow = q0 + qb * 32 + l31;
        // 16-byte stores (round 6, as in ip_attn_kernel where the store shape was measured: attention.hip): a lane holds columns
        // 8g + 4 lhi + {0..3} of its query row, its partner lane ^ 32 the other four of every group of eight; one
        // v_permlane32_swap per dword and group pair leaves the lower lane with columns 16j .. 16j+7, the upper one with 16j+8 ..
        // 16j+15.  The swaps run on every lane (rows past Nq included); only the stores are masked.
        half_t* op = p.o + (long)b * p.so + (long)min(qrow, p.Nq - 1) * p.ldo + h * 64 + lhi * 8;
        const bool row_ok = qrow < p.Nq;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            unsigned og[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)(O[qb][db][4 * g + e] * inv);
                const u32x2_sp t2 = __builtin_bit_cast(u32x2_sp, o);
                og[g][0] = t2[0], og[g][1] = t2[1];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(og[2 * j][0]), "+v"(og[2 * j + 1][0]));
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(og[2 * j][1]), "+v"(og[2 * j + 1][1]));
                const u32x4_sp v = {og[2 * j][0], og[2 * j][1], og[2 * j + 1][0], og[2 * j + 1][1]};
                if (row_ok) *reinterpret_cast<u32x4_sp*>(op + db * 32 + j * 16) = v;
            }
        }
    }
}

}  // namespace

// reads (and optionally clears) the re-centring counter of the current device; synchronises the device (debug / test use only)
int ds_attn_sp_recentre_count(int reset, long long* value) {
    unsigned long long v = 0;
    DS_HIP(hipDeviceSynchronize());
    DS_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sp_recentre_count), sizeof(v)));
    if (reset) {
        const unsigned long long z = 0;
        DS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sp_recentre_count), &z, sizeof(z)));
    }
    if (value) *value = (long long)v;
    return 0;
}

int ds_launch_self_attn_sp(const SelfAttnParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(self_attn_sp_kernel, dim3(((p.Nq + 255) / 256) * p.B * p.heads), dim3(256), 0, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

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
//     pre-multiplied by scale * log2 e once), so a probability is ONE v_exp_f32.  m_ref is only raised when a tile's maximum
//     exceeds it by more than THR = 8 (p <= 256 fits f16 and f32 comfortably); that, the first tile and the ragged last tile
//     are rare wave-uniform branches at the top of a step, outside the scheduled region.
//   * the maximum of pair k+1's scores is taken at the end of step k (beside the P V MFMAs), so the decision at the top of step
//     k+1 is one cross-half exchange and a compare.
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
constexpr float THR = 8.0f;
constexpr float NEG_BIG = -1.0e30f;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

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
    const int srow = lane >> 3, sslot = lane & 7;
    auto issue_k = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (4 * j + wave) * 8 + srow;
            const int key = min(t * 64 + row, p.Nk - 1);                 // rows past Nk re-read the last key (masked later)
            const int chunk = sslot ^ ((row >> 1) & 7);
            __builtin_amdgcn_global_load_lds((glb_void*)(kbase + (long)key * p.ldk + chunk * 8),
                                             (lds_void*)(smem + buf * TILE_B + (4 * j + wave) * 1024), 16, 0, 0);
        }
    };
    auto issue_v = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (4 * j + wave) * 8 + srow;                   // d
            const int chunk = sslot ^ ((row >> 1) & 7);
            const int kc = min(t * 64 + chunk * 8, ((p.Nk - 1) >> 3) << 3);  // chunks past Nk re-read the last one (P = 0 there)
            __builtin_amdgcn_global_load_lds((glb_void*)(vbase + (long)row * p.ldv + kc),
                                             (lds_void*)(smem + V_BASE + buf * TILE_B + (4 * j + wave) * 1024), 16, 0, 0);
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
    // rare work at the top of a step (outside the scheduled region): ragged-key masking, the deferred rescale
    auto fixup = [&](f32x16 (&S)[2], int qb, float& mx, int t) {
        if (ragged && t == nt - 1) {
            int nk_here = p.Nk;
            asm volatile("" : "+s"(nk_here)::"memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kb * 32 + pi32((r & 3) + 8 * (r >> 2) + 4 * lhi);
                    if (key >= nk_here) S[kb][r] = NEG_BIG;
                }
            mx = cross_max(tile_max(S));
        }
        const bool first = t == 0;
        if (first || __builtin_amdgcn_ballot_w64(mx > THR) != 0) {
            const float delta = first ? mx : fmaxf(mx, 0.f);   // the lane's reference maximum moves to m_ref + delta
            const float alpha = first ? 1.0f : fexp2(-delta);
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
        }
    };
    // One step: softmax of the CURRENT pair on the VALU beside the MFMAs of the NEXT pair's scores and the PREVIOUS pair's P V
    auto step = [&](f32x16 (&Sc)[2], h8 (&Pc)[2][2], int qbc, f32x16 (&Sn)[2], unsigned kb_base, const h8 (&Pp)[2][2],
                    unsigned vb_base, float& mx_next) {
        const int qbn = qbc ^ 1;
        qk(Sn, qbn, kb_base);
        pv(Pp, qbn, vb_base);
        // row sum, two scores per v_pk_add_f32: 17 fewer VALU per step and the same kernel time as scalar adds (239.6 vs
        // 239.3 us at B = 32, N = 1024, profiles/r03_self_attn_sp.txt) - the step is not VALU-issue bound any more
        f32x2 s2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 e = {fexp2(Sc[kb][r]), fexp2(Sc[kb][r + 1])};
                s2 += e;
                Sc[kb][r] = e[0];
                Sc[kb][r + 1] = e[1];
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int e = 0; e < 8; ++e) Pc[kb][hb][e] = (half_t)Sc[kb][hb * 8 + e];
        }
        lsum[qbc] += s2[0] + s2[1];
        mx_next = tile_max(Sn);
        // pin this step's results HERE: the packed probabilities are only consumed by the next step's MFMAs, and the
        // compiler otherwise sinks the whole softmax below the branch between the steps - out of this scheduling region
        // (inputs only: a tied in/out operand made the register allocator copy every packed pair into place - 16 v_mov per step)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) asm volatile("" ::"v"(Pc[kb][hb]));
        asm volatile("" ::"v"(lsum[qbc]), "v"(mx_next));
        // issue order: the bias block and two fragments first (the first MFMA needs them), then 16 x {1 MFMA, 1 fragment
        // read, 2 exponentials, 5 other VALU}: the matrix pipe never waits for a block of VALU work
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);      // VALU: the 16 bias registers
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // DS read
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            if (g < 14) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);   // TRANS
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // VALU
        }
    };

    // ---- prologue, part 2: scores of pair (0, qb 0)
    SP_SYNC();
    qk(S0, 0, 0u);
    float mx0 = cross_max(tile_max(S0)), mx1 = 0.f;

    int kcur = 0, knext = nt > 1 ? 1 : 0, vprev = 0, vcur = 0;  // ring-buffer indices of K(t), K(t+1), V(t-1), V(t)
    for (int t = 0; t < nt; ++t) {
        // K(t+1) and V(t) were requested a whole iteration ago; everybody is done with K(t-1) and V(t-2)
        SP_SYNC();
        if (t + 2 < nt) issue_k(t + 2, (t + 2) % 3);
        if (t + 1 < nt) issue_v(t + 1, (t + 1) % 3);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- step A: softmax(t, qb 0) | scores(t, qb 1) | P V (t-1, qb 1)
        fixup(S0, 0, mx0, t);
        float mraw;
        step(S0, P0, 0, S1, (unsigned)kcur * TILE_B, P1, (unsigned)vprev * TILE_B, mraw);
        mx1 = cross_max(mraw);
        // ---- step B: softmax(t, qb 1) | scores(t+1, qb 0) | P V (t, qb 0)
        fixup(S1, 1, mx1, t);
        step(S1, P1, 1, S0, (unsigned)knext * TILE_B, P0, (unsigned)vcur * TILE_B, mraw);
        mx0 = cross_max(mraw);
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
        if (qrow < p.Nq) {
            half_t* op = p.o + (long)b * p.so + (long)qrow * p.ldo + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)(O[qb][db][4 * g + e] * inv);
                    *reinterpret_cast<h4*>(op + db * 32 + 8 * g + 4 * lhi) = o;
                }
        }
    }
}

}  // namespace

int ds_launch_self_attn_sp(const SelfAttnParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(self_attn_sp_kernel, dim3(((p.Nq + 255) / 256) * p.B * p.heads), dim3(256), 0, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

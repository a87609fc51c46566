// fp16 MFMA GEMM, 256 x 256 x 64 block tile, 8 waves in two staggered groups ("ping-pong"), persistent blocks:
// the large-problem path of ds_launch_gemm (same GemmParams and epilogue semantics as gemm.hip; the reference call
// sites are listed there).
//
// Why a second main loop.  The 128 x 128 kernels of gemm.hip move 32 KiB of operand tiles through L2 -> LDS per
// 2 MFLOP of work; a CU sustains ~25-40 B/clk on that path (profiles/r01_gemm_ablation.txt), which caps them near
// 0.8 PFLOP/s whatever the schedule.  A 256 x 256 tile halves the bytes per flop, but one workgroup then owns the CU,
// so the overlap that three independent 128 x 128 blocks got for free has to be built inside the block:
//
//   * 8 waves = 2 (M) x 4 (N); waves w and w+4 share a SIMD.  Waves 4..7 run one barrier behind waves 0..3, so on
//     every SIMD one wave is inside its MFMA cluster (s_setprio 1) while the other issues ds_reads and LDS-DMA.
//   * a k-tile is four phases; each phase = {ds_read one operand sub-tile, stage one 16-KiB half-tile of a later
//     k-tile with global_load_lds, s_barrier, 8 x v_mfma_f32_32x32x16_f16 on one 64 x 32 quadrant, s_barrier}.
//   * the tile is cut into half-tiles A0/A1 (tile rows 0..127 / 128..255) and B0/B1 so that each is read in exactly
//     one phase (A0: P1, B0: P1, B1: P2, A1: P3; B0's fragments stay in registers for P4).  A wave owns rows
//     {64 wr.., 128 + 64 wr..} and two 32-column strips, one out of each B half.
//   * LDS: 2 buffers x 4 half-tiles x 16 KiB = 128 KiB (+ 8 x 4 KiB wave-private epilogue staging = all 160 KiB).
//     Staging order P1: A1(t+1), P2: B0(t+2), P3: A0(t+2), P4: B1(t+2): every slot is refilled as soon as it is
//     free, and waited for (counted s_waitcnt vmcnt(10): all but the five newest half-tiles) one phase before its
//     own read, so ~80 KiB per CU are in flight in the steady state.  The k-loop has no branch: past the end of K the
//     stages re-fetch the last k-tile into slots nobody reads, so the counts never change.
//   Hazards (group 1 runs one barrier late, so a phase spans three barrier intervals):
//     RAW  data waited for in phase p is read from phase p+1 on.
//     WAR  a half-tile is restaged >= 2 phases after its last ds_read; B0 is restaged ONE phase after its read,
//          which is legal only because P1 issues the B0 reads first and retires them (lgkmcnt(8)) before P1's
//          first barrier.
//   * persistent: a block walks tiles id = round * grid + chunk-of-its-XCD, and the k-loop never drains between two of
//     them (round 5): the stages of a tile's last two k-tiles, which used to re-fetch the last k-tile into slots nobody
//     reads, fetch k-tiles 0 and 1 of the NEXT output tile instead - they ARE the next tile's pipeline fill, in the
//     steady-state order, slots and hazards, so a tile boundary issues no prologue, no extra wait and no extra barrier; the
//     next tile's coordinates and operand addresses (~190 scalar instructions with three integer divisions, on the critical
//     path of every boundary until round 4) are computed under the first k-tile of the current one, where the scalar unit idles.
//     For the fill to hide under the epilogue no wait of the epilogue may cover it (vmcnt retires in issue order): the tile's
//     bias slice is staged into LDS with the tile's first k-tiles and read from there, the residual rows of a 32-row
//     piece are requested before the previous piece's stores, and the C stores themselves keep draining under the next
//     tile's first k-tiles (EX_TAIL).  Measured (profiles/r02_pp_epilogue_ab.txt): M = 32768, N = 2560, K = 1280
//     222 -> 187 us (967 -> 1146 TFLOP/s, F.linear / hipBLASLt on the same box: 1041-1168); GEGLU 893 -> 772 us.
//
// Column ownership is chosen for the epilogue (the B half-tiles are just a permutation of the 256 tile columns):
//   plain  half h, LDS row r -> tile column (r>>5)*64 + h*32 + (r&31): a wave ends up with 64 adjacent columns, i.e.
//          whole 128-byte lines of C per row;
//   GEGLU  (packed weights: every 128 columns = 64 hidden | their 64 gates)  (r>>6)*128 + h*64 + (r&63): a lane holds
//          a hidden value (B0 strip) and its gate (B1 strip), so h * gelu(g) happens in registers.
// The epilogue transposes 32-row pieces through the wave's private 4 KiB of LDS (swizzled, conflict-free, no
// barrier) so every global store / residual load is 16 bytes per lane over whole rows.
//
// LayerNorm (round 4; the three LayerNorms of a BasicTransformerBlock [3P], reached from src/models/unet.py:244-338, were
// stand-alone passes: read h, write LN(h), 76 us each at UNet batch 64).  The normalisation is folded into the GEMM pair
// around it instead:
//   producer (the +residual projection that writes h): the branch-free plain epilogue also emits, per row and 64-column
//     strip, the (sum, sum of squares) of the f16 values it stores (`stats_out`, [N/64][M] float2; 8-lane DPP butterfly in
//     the row-contiguous layout the stores already use); a 5-us launch (`ln_finalize_kernel`) turns them into (mean, rstd);
//   consumer: A is the RAW h (the LDS-DMA staging cannot transform it), W is gamma (.) W packed at load time, and
//     LN(h) W^T + b = rstd_m (h_m . gw_n - mean_m c_n) + b'_n with c_n = sum_k gw_nk, b' = b + W beta.  The rank-1 term is
//     ONE extra MFMA per 32 x 32 accumulator block (k slots {-c hi, -c lo, -c hi} x {mean hi, mean hi, mean lo} in f16
//     pairs: the product is exact to 2^-22) and the row factor is the multiplier of the fma that used to be the bias add -
//     in this kernel a lane owns a tile ROW, so rstd is a per-lane scalar.  The tile's statistics / c slices are LDS-DMA
//     pieces staged with the tile's first k-tiles, like the bias.
#include <type_traits>

#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr int HT = 16384;  // bytes of one half-tile: 128 rows x 64 k of f16
// Every prologue is FOLLOWED by at least this many vector-memory instructions of the same wave before the tile's first
// k-tile: the C stores of the previous tile's last two 32-row pieces (GEGLU: of all four) in the branch-free epilogues,
// eight dummy LDS-DMA pieces (`pad_tail`) behind the first prologue and behind the generic epilogue, whose store count
// depends on the tile's edges.  The counted waits of the first k-tiles leave exactly this many outstanding: more issued
// than counted only waits longer, fewer would under-wait.
#ifndef PP_EX
#define PP_EX 16
#endif
// How many of a tile's trailing vector-memory instructions - its C stores - may still be in flight when the next tile's first
// k-tiles run.  16 = every store of a plain tile (round 5; 8 until round 4: the first half of a tile's stores had to be
// acknowledged before the next tile's first MFMA, and the whole chip writes its tiles in the same microseconds).  Build-time
// A/B (-DPP_EX=8, tools/build_variants.py), batch-64 forward, same box: gemm_pp_kernel 258.6 -> 257.4 ms
// (profiles/r05_pp_ex16_pf_ab.txt).
constexpr int EX_MAX = PP_EX;
// every epilogue path issues (or pads to) at least EX_TAIL vector-memory instructions behind the next tile's prologue: plain 16
// stores, unfused GEGLU 8 stores + 8 pad pieces, fused GEGLU (FUSE 9) 8 stores + bias + 2 LayerNorm pieces = 11, generic 16 pad
// pieces.  A PP_EX above 16 would make `wait_newer` under-wait: a tile would read LDS before its DMA has landed.
static_assert(EX_MAX >= 0 && EX_MAX <= 16, "PP_EX: no epilogue path leaves more than 16 vector-memory instructions in flight");
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int V>
struct IC {
    static constexpr int value = V;
};

// v_mov_b32 with a DPP control: 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1] (lane ^ 2),
// 0x141 = row_half_mirror (lane j of every 8 <-> lane 7 - j)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// v_permlane32_swap_b32 vdst, src: lanes 32..63 of vdst <-> lanes 0..31 of src.  Inline asm: through the builtin hipcc has folded
// uses of the second result away (attention_sp.hip, cross_max); the s_nop covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void swap32(unsigned& vdst, unsigned& src) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(vdst), "+v"(src));
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// PP_DIRECT_EPI (round 5): the branch-free GEGLU epilogue hands its 32 x 32 output pieces to memory WITHOUT the transposition
// through LDS.  A lane holds tile row l31; its two half-wave partners (l, l + 32) hold columns 8g + {0..3} and 8g + {4..7} of every
// group g, so one v_permlane32_swap per dword and group pair leaves the lower lane with columns 16j .. 16j+7 and the upper one
// with 16j+8 .. 16j+15 of the row: one 16-byte store per lane, 32 rows x 32 bytes per instruction - a 64-byte output row leaves in
// two halves (cdna_hip_programming.md T21).  Per wave and tile 16 ds_write_b64 + 8 ds_read_b128 and their write -> read round
// trips become 16 swaps: the GEGLU GEMMs -1.4 % (1542.8 -> 1521.6 us at M = 65536; profiles/r05_pp_direct_epilogue_ab.txt).
// The plain epilogues keep the LDS transposition: their 128-byte rows in four pieces measured +4 ... +15 % (same file).
// -DPP_DIRECT_EPI=0: the LDS transposition here too (A/B).
#ifndef PP_DIRECT_EPI
#define PP_DIRECT_EPI 1
#endif

#define PP_FENCE()                               \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#define PP_BARRIER()                             \
    do {                                         \
        PP_FENCE();                              \
        __builtin_amdgcn_s_barrier();            \
        PP_FENCE();                              \
    } while (0)

// Static priority (round 5; MI355X_MICROARCH.md, "static priority for the younger half"): waves 4..7 - dispatched second, the
// arbitration losers of every phase - run at priority 1 for the whole kernel, and nobody flips s_setprio around its MFMA
// clusters any more (rounds 1-4: s_setprio 1 / 0 around every cluster of every wave).  Build-time A/B, batch-64 forward, two
// interleaved rounds on one box: every instantiation -0.6...-0.9 %, forward 442.6 -> 440.9 ms, same bits
// (profiles/r05_pp_static_prio_ab.txt).  -DPP_PER_CLUSTER_PRIO restores the flips.
#ifndef PP_PER_CLUSTER_PRIO
#define PP_STATIC_PRIO 1
#define PP_PRIO(v) do { } while (0)
#else
#define PP_PRIO(v)                                              \
    do {                                                        \
        if constexpr ((DBG & 8) == 0) __builtin_amdgcn_s_setprio(v); \
    } while (0)
#endif

// FUSE (fused LayerNorm, see the header): 0 none - the instantiation every other GEMM runs, its code is untouched by the
// feature; 1 consumer with the plain epilogue, 9 consumer with the GEGLU epilogue (statistics + c pieces, one extra MFMA per
// accumulator block, rstd applied to the accumulators / as the multiplier of the bias fma); 4 consumer in the operand-swapped
// form; 2 producer (row statistics out of the plain epilogue).  Separate instantiations - each compiles only the epilogue it
// runs - because the kernel sits exactly at its 256-register budget.
template <typename T, int DBG, int FUSE = 0>  // T: half_t (UNet) or bf16_t (VAE decoder).  DBG: ablation builds only: 1 = no MFMA, 2 = no tile loads, 4 = no fragment reads (garbage results), 8 = no s_setprio, 16 = clock probe written over C[0..15], 128 = no epilogue; 0 in production
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
    typedef typename Elt<T>::v8 V8;
    typedef typename Elt<T>::v4 V4;
    // the fused GEGLU epilogue issues 8 stores + the bias piece + 2 LayerNorm pieces per tile: its count is capped at 11
    constexpr int EX_TAIL = (FUSE == 9 && EX_MAX > 11) ? 11 : EX_MAX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = p.K / 64;
    const bool geglu = p.epi == EPI_GEGLU;
    // batched problems (the V^T projections: one GEMM per image, shared A) are folded into the tile walk: logical id ->
    // (batch item, tile of that item), items outermost - an XCD's chunk of consecutive ids stays inside one or two items and
    // shares their operand panels; every item's tile order is the plain one
    const int per_item = p.tiles_m * p.tiles_n;
    const int ntiles = per_item * p.nbatch;

    // ---- persistent tile walk: round-major, then one contiguous chunk of the logical order per XCD (block b runs on
    // XCD b % 8), so the tiles an XCD has in flight share A / W panels in its L2
    const int G = gridDim.x;
    const int tile_local = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;

    // ---- staging: every wave moves LDS rows 16w .. 16w+15 of each half-tile (two 1-KiB LDS-DMA pieces).  The DMA
    // destination is lane-linear, so the XOR swizzle is applied to the lane's SOURCE chunk.
    // Everything derived from the lane id is REBUILT per output tile from an opaque v_mbcnt (derive_stage / derive_frag):
    // kept live across the epilogue these 13 registers (and `lane` itself) pushed the epilogue's own values - packed
    // bias, residual rows - into scratch, and a scratch reload waits with vmcnt(0).
    auto lane_id = [&]() -> int {
        unsigned z = 0;
        asm volatile("" : "+v"(z));  // opaque start value: the count is recomputed here, not kept in a register per kernel
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    };
    unsigned oA[2], oB[2];  // BYTE offsets, unsigned: (uniform base) + zext(lane offset) selects the SGPR-base address mode
    auto derive_stage = [&]() {
        const int ln = lane_id();
        const int lrow = ln >> 3, slot = ln & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = j * 8 + lrow;                     // row within the wave's 16
            const int chunk = slot ^ ((r >> 1) & 7);        // (16w + r) >> 1 & 7 == (r >> 1) & 7
            oA[j] = (unsigned)(r * (int)p.lda + chunk * 8) * 2u;
            oB[j] = (unsigned)(r * (int)p.ldw + chunk * 8) * 2u;
        }
    };
    // tile columns of this wave's staging rows, per B half (see the header): first column of its 16
    const int cb0 = geglu ? (wave >> 2) * 128 + (wave & 3) * 16 : (wave >> 1) * 64 + (wave & 1) * 16;
    const int cbh = geglu ? 64 : 32;
    const half_t* gA[2];   // current tile: first row of the wave's 16 staged rows per half-tile
    const half_t* gB[2];
    const half_t* nA[2];   // next tile of this block (the current one again when there is none: a harmless re-fetch)
    const half_t* nB[2];
    // tile id -> coordinates -> operand rows, in three pieces: at a tile boundary they are issued one piece per MFMA cluster of
    // k-tile 0 (`hide_next` below), a handful of scalar instructions between every two MFMAs
    auto tile_origin = [&](int id, int& m0, int& n0, int& bz) {
        int tm, tn;
        bz = id / per_item;   // (0 for an unbatched problem; unconditional: a branch here would split the MFMA cluster it is issued in)
        tile_coords(id - bz * per_item, p.tiles_m, p.tiles_n, tm, tn);
        if constexpr ((DBG & 32) != 0) tm = tn = 0;  // ablation: every block works on tile (0,0): all operand loads hit L2
        m0 = tm * 256;
        n0 = tn * 256;
    };
    // ragged edges: a wave's 16 rows are all inside or all outside (M, N multiples of 16); outside rows re-read the
    // tile's first row - their products are never stored
    auto rows_a = [&](int m0, int bz, const half_t* (&a)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ra = m0 + h * 128 + wave * 16;
            a[h] = p.A + (long)bz * p.sA + (long)(ra < p.M ? ra : m0) * p.lda;
        }
    };
    auto rows_b = [&](int n0, int bz, const half_t* (&b)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rb = n0 + cb0 + h * cbh;
            b[h] = p.W + (long)bz * p.sW + (long)(rb < p.N ? rb : n0) * p.ldw;
        }
    };
    auto set_tile = [&](int id, int& m0, int& n0, int& bz, const half_t* (&a)[2], const half_t* (&b)[2]) {
        tile_origin(id, m0, n0, bz);
        rows_a(m0, bz, a);
        rows_b(n0, bz, b);
    };
    char* const sdst = smem + wave * 2048;
    // crossc = IC<1>: kt may be nk or nk + 1 = k-tiles 0, 1 of the NEXT output tile (wave-uniform: scalar selects, no branch);
    // IC<0>: the caller knows kt < nk (the first two k-tiles of a tile with nk >= 4, whose stages must not depend on the
    // next tile's rows: those are still being computed beside them)
    auto stage = [&](int op, int half, int buf, int kt, auto crossc) {
        char* d = sdst + ((op * 2 + half) * 2 + buf) * HT;
        const char* s;
        if constexpr (decltype(crossc)::value != 0) {
            const bool nx = kt >= nk;
            const half_t* const base = op == 0 ? (nx ? nA[half] : gA[half]) : (nx ? nB[half] : gB[half]);
            s = reinterpret_cast<const char*>(base + (nx ? kt - nk : kt) * 64);
        } else {
            s = reinterpret_cast<const char*>((op == 0 ? gA[half] : gB[half]) + kt * 64);
        }
        const unsigned o0 = op == 0 ? oA[0] : oB[0], o1 = op == 0 ? oA[1] : oB[1];
        if constexpr ((DBG & 2) != 0) return;
        __builtin_amdgcn_global_load_lds((glb_void*)(s + (size_t)o0), (lds_void*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(s + (size_t)o1), (lds_void*)(d + 1024), 16, 0, 0);
    };
    // The block's FIRST tile only: all of k-tile 0 and B0/A0/B1 of k-tile 1 (its A1 is staged by k-tile 0's P1), in the
    // order the last two k-tiles of a tile stage them for every later one
    auto stage_prologue = [&]() {
        stage(1, 0, 0, 0, IC<0>{});
        stage(0, 0, 0, 0, IC<0>{});
        stage(1, 1, 0, 0, IC<0>{});
        stage(0, 1, 0, 0, IC<0>{});
        stage(1, 0, 1, 1, IC<0>{});   // nk >= 2: K is a multiple of 128
        stage(0, 0, 1, 1, IC<0>{});
        stage(1, 1, 1, 1, IC<0>{});
    };

    // Interior tiles without a per-row bias or activation take the branch-free epilogues (all tiles of the UNet's shapes).
    // Round 6: so do the plain tiles of a ragged last tile COLUMN when N is a multiple of 64 (the 640-channel level: 2.5 tile
    // columns) - a wave owns 64 adjacent columns, so it is wholly inside the problem (it runs the branch-free epilogue) or
    // wholly outside (it stores nothing and only pads its vector-memory count, `wave_cols_in`).  Until then those tiles took
    // the generic epilogue (a branch and an s_waitcnt vmcnt(0) per 4 values: 4-8 us per tile), which is why the N, K <= 640
    // projections stayed on the 128 x 128 kernels.  gemm_debug bit 12 (4096): ragged tiles on the generic epilogue (A/B).
    const bool strips_ok = !geglu && (p.N & 63) == 0 && (p.debug & 4096) == 0;
    auto is_fast = [&](int tm0, int tn0) -> bool {
        return tm0 + 256 <= p.M && (tn0 + 256 <= p.N || strips_ok) && !p.rowbias && (geglu || p.epi == EPI_NONE);
    };
    char* const ep = smem + 8 * HT + wave * 4096;  // wave-private transposition tile of the epilogue
    // One LDS-DMA piece: the bias values of the wave's 64 output columns (GEGLU: 32 hidden | their 32 gates) -> bytes
    // 0..127 of `ep` (lanes 8.. repeat them).  Issued behind the tile's prologue (and the previous tile's epilogue, which
    // still uses `ep`), so it is one of the EX_TAIL instructions and has landed long before the epilogue reads it.
    auto stage_bias = [&](int tn0) {
        const int l7 = lane_id() & 7;
        int col = geglu ? tn0 + (wc >> 1) * 128 + (wc & 1) * 32 + (l7 >> 2) * 64 + (l7 & 3) * 8 : tn0 + wc * 64 + l7 * 8;
        if (col >= p.N) col = 0;   // a wave outside a ragged tile column: a harmless piece, its count stays the same
        __builtin_amdgcn_global_load_lds((glb_void*)(p.bias + col), (lds_void*)ep, 16, 0, 0);
    };
    auto pad_tail = [&](auto nc) {  // harmless pieces into the idle last KiB of `ep` (KiB 1 and 2 hold the LayerNorm pieces)
#pragma unroll
        for (int j = 0; j < decltype(nc)::value; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(gA[0]) + (size_t)oA[0]),
                                             (lds_void*)(ep + 3072), 16, 0, 0);
    };
    // Fused LayerNorm, consumer side: two more LDS-DMA pieces per tile, issued where the bias piece is -
    //   ep + 1024: (-c hi, -c lo) of the wave's 64 output columns, 4 bytes each, in accumulator-strip order (lanes 16.. repeat);
    //   ep + 2048: (mean, rstd) of the wave's 128 rows, 8 bytes each: rows 64 wr.. at +0, rows 128 + 64 wr.. at +512.
    auto stage_ln = [&](int tm0, int tn0, int tbz) {
        const int ln = lane_id();
        if constexpr ((FUSE & 4) != 0) {
            // operand-swapped form (V^T = Wv X_b^T: the normalised rows are the rows of W, i.e. the tile COLUMNS):
            //   ep + 1024: (mean, rstd) of the wave's 64 output columns = rows tbz * ln_bstride + tn0 + 64 wc .. of the
            //              normalised matrix, 8 bytes each (lanes 32.. repeat);
            //   ep + 2048: (-c hi, -c lo, b' hi, b' lo) of the wave's 128 output rows, 8 bytes each, laid out like the
            //              statistics of the row form.
            const long col = (long)tbz * p.ln_bstride + tn0 + wc * 64 + (ln & 31) * 2;
            __builtin_amdgcn_global_load_lds((glb_void*)(p.ln_stats + 2 * col), (lds_void*)(ep + 1024), 16, 0, 0);
            int row = tm0 + (ln >> 5) * 128 + wr * 64 + (ln & 31) * 2;
            if (row >= p.M) row = 0;   // ragged last tile ROW (M % 32 == 0, the 640-channel level): rows never stored
            __builtin_amdgcn_global_load_lds((glb_void*)(p.ln_c + 4 * (long)row), (lds_void*)(ep + 2048), 16, 0, 0);
        } else {
            const int L = ln & 15;
            int col = geglu ? tn0 + (wc >> 1) * 128 + (wc & 1) * 32 + (L >> 3) * 64 + (L & 7) * 4 : tn0 + wc * 64 + L * 4;
            if (col >= p.N) col = 0;   // (see stage_bias)
            __builtin_amdgcn_global_load_lds((glb_void*)(p.ln_c + 2 * (long)col), (lds_void*)(ep + 1024), 16, 0, 0);
            const int row = tm0 + (ln >> 5) * 128 + wr * 64 + (ln & 31) * 2;
            __builtin_amdgcn_global_load_lds((glb_void*)(p.ln_stats + 2 * (long)row), (lds_void*)(ep + 2048), 16, 0, 0);
        }
    };

    // ---- fragment addresses: row r of a half-tile at r*128, 16-byte chunk c at ((c ^ ((r>>1)&7)) << 4)
    unsigned fA[4], fB[4];  // LDS byte offsets, one per k-step: half / buffer / row-block are immediate ds_read offsets
    auto derive_frag = [&]() {
        const int ln = lane_id();
        const int l31 = ln & 31, lhi = ln >> 5;
        const int sw = (l31 >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fA[kk] = (wr * 64 + l31) * 128 + (((kk * 2 + lhi) ^ sw) << 4);
            fB[kk] = 4 * HT + (wc * 32 + l31) * 128 + (((kk * 2 + lhi) ^ sw) << 4);
            asm volatile("" : "+v"(fA[kk]), "+v"(fB[kk]));  // keep the eight addresses resident: no VALU in the load phases
        }
    };
    V8 dummy = {};
    if constexpr ((DBG & 4) != 0) asm volatile("" : "+v"(dummy));
    auto readA = [&](int half, int buf, int mi, int kk) -> V8 {
        if constexpr ((DBG & 4) != 0) return dummy;  // ablation: no fragment reads
        return *reinterpret_cast<const V8*>(smem + fA[kk] + ((half * 2 + buf) * HT + mi * 4096));
    };
    auto readB = [&](int half, int buf, int kk) -> V8 {
        if constexpr ((DBG & 4) != 0) return dummy;
        return *reinterpret_cast<const V8*>(smem + fB[kk] + (half * 2 + buf) * HT);
    };

    // Counted wait: returns once every LDS-DMA older than the `n` newest half-tiles (2 instructions each) has landed.
    // Each half-tile is staged as soon as its slot is free (2 phases after the slot's read) and waited for one phase
    // before its own read, so 5 half-tiles (80 KiB) are in flight per CU in the steady state: ~10 barrier intervals
    // (~1.5 us) of latency cover.  (One wait per k-tile - vmcnt(6) in P4 - measured the same within noise.)
    // EX: further vector-memory instructions this wave has issued AFTER the half-tile waited for and that may stay in
    // flight as well - the previous output tile's last C stores (EX_TAIL).  vmcnt counts loads and stores alike and they
    // retire in issue order, so "all but the n newest" is exact whatever the mix.
    auto wait_newer = [&](auto nc, auto exc) {
        constexpr int N = 2 * decltype(nc)::value + decltype(exc)::value;
        static_assert(N <= 63, "vmcnt is a 6-bit field");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    };

    f32x16 acc[4][2];
    constexpr bool mma_on = (DBG & 1) == 0;
    // No branch anywhere in a k-tile (a taken scalar branch costs more than the slack a phase has): past the end of K the
    // stages fetch the next output tile's first two k-tiles (`stage`), so neither the counted waits nor the slot schedule
    // ever change - the last k-tile of one tile and the first of the next are two ordinary neighbours.
    // `beside(IC<phase>)`: scalar work issued INSIDE the phase's MFMA cluster, six scalar instructions behind every MFMA
    // (sched_group_barrier): the next tile's coordinates in k-tile 0, nothing anywhere else.
    auto nothing = [](auto) {};
    auto ktile = [&](auto bufc, auto exa, auto exb, auto exd, auto crossc, int kt, auto beside) {  // exa / exb / exd: EX of the P1 / P2 / P4 wait
        constexpr int B = decltype(bufc)::value;
        constexpr bool BESIDE = !std::is_same<decltype(beside), decltype(nothing)>::value;
        auto interleave = [&]() {
            if constexpr (BESIDE) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x004, 6, 0);   // SALU
                }
            }
        };
        const int kt1 = kt + 1, kt2 = kt + 2;
        V8 bl[4], br[4], a0[4][2], a1[4][2];
        // ---------------- P1: B0 strip (first: retired before the barrier, see WAR above) + A0 rows
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bl[kk] = readB(0, B, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a0[kk][0] = readA(0, B, 0, kk);
            a0[kk][1] = readA(0, B, 1, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
        stage(0, 1, B ^ 1, kt1, crossc);
        wait_newer(IC<5>{}, exa);  // B1(kt), read in P2: newer = A1(kt) + B0 A0 B1 A1 of kt+1
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_PRIO(1);
        beside(IC<1>{});
        if constexpr (mma_on) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][0] = Elt<T>::mfma(bl[kk], a0[kk][0], acc[0][0]);
                acc[1][0] = Elt<T>::mfma(bl[kk], a0[kk][1], acc[1][0]);
            }
        }
        interleave();
        PP_PRIO(0);
        PP_BARRIER();
        // ---------------- P2: B1 strip
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) br[kk] = readB(1, B, kk);
        __builtin_amdgcn_sched_barrier(0);
        stage(1, 0, B, kt2, crossc);
        wait_newer(IC<5>{}, exb);  // A1(kt), read in P3: newer = all of kt+1 + B0(kt+2)
        PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_PRIO(1);
        beside(IC<2>{});
        if constexpr (mma_on) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][1] = Elt<T>::mfma(br[kk], a0[kk][0], acc[0][1]);
                acc[1][1] = Elt<T>::mfma(br[kk], a0[kk][1], acc[1][1]);
            }
        }
        interleave();
        PP_PRIO(0);
        PP_BARRIER();
        // ---------------- P3: A1 rows
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a1[kk][0] = readA(1, B, 0, kk);
            a1[kk][1] = readA(1, B, 1, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
        stage(0, 0, B, kt2, crossc);
        PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_PRIO(1);
        beside(IC<3>{});
        if constexpr (mma_on) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[2][1] = Elt<T>::mfma(br[kk], a1[kk][0], acc[2][1]);
                acc[3][1] = Elt<T>::mfma(br[kk], a1[kk][1], acc[3][1]);
            }
        }
        interleave();
        PP_PRIO(0);
        PP_BARRIER();
        // ---------------- P4: no reads (B0 strip still in registers); the k-tile's one counted wait
        stage(1, 1, B, kt2, crossc);
        wait_newer(IC<5>{}, exd);  // B0 A0 (kt+1), read in the next P1: newer = B1 A1 (kt+1) + B0 A0 B1 (kt+2)
        PP_BARRIER();
        PP_PRIO(1);
        beside(IC<4>{});
        if constexpr (mma_on) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[2][0] = Elt<T>::mfma(bl[kk], a1[kk][0], acc[2][0]);
                acc[3][0] = Elt<T>::mfma(bl[kk], a1[kk][1], acc[3][0]);
            }
        }
        interleave();
        PP_PRIO(0);
        PP_BARRIER();
    };

    int id = tile_local;
    if (id >= ntiles) return;
#ifdef PP_STATIC_PRIO
    if (wr == 1) __builtin_amdgcn_s_setprio(1);
#endif
    unsigned long long probe_c0 = 0, probe_r0 = 0;
    if constexpr ((DBG & 16) != 0) {  // clock probe: shader cycles vs the constant 100 MHz reference
        probe_c0 = __builtin_readcyclecounter();
        probe_r0 = __builtin_amdgcn_s_memrealtime();
    }
    int m0, n0, bz;      // current tile
    int nm0, nn0, nbz;   // next tile of this block
    set_tile(id, m0, n0, bz, gA, gB);
    nA[0] = gA[0], nA[1] = gA[1], nB[0] = gB[0], nB[1] = gB[1];
    derive_stage();
    stage_prologue();
    if (p.bias && is_fast(m0, n0)) stage_bias(n0);
    if constexpr ((FUSE & 5) != 0) stage_ln(m0, n0, bz);
    pad_tail(IC<EX_TAIL>{});  // no epilogue yet behind the first prologue
    derive_frag();
    // A0 B0 of the first tile's k-tile 0 must have landed: all but the five newer half-tiles of the prologue and the EX_TAIL
    // instructions behind them.  Every later tile finds them waited for by the previous tile's last P4, like any k-tile.
    wait_newer(IC<5>{}, IC<EX_TAIL>{});
    PP_BARRIER();
    while (true) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // The previous tile's last C stores (EX_TAIL of them) keep draining under the first k-tiles.  (A vmcnt(0) here parked
        // every wave of the CU until its C stores were acknowledged - and the whole chip writes its 256 x 256 tiles in the same
        // few microseconds of each round.)  They are older than A1(k-tile 1), staged in k-tile 0's P1: the waits for it and for
        // everything younger (k-tile 1's P2 on) already imply them.  gemm_debug bit 8 (256): drain first (A/B).
        if ((p.debug & 256) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wr == 1) PP_BARRIER();  // group 1 runs one barrier behind group 0 from here on
        // ---- the NEXT tile's coordinates and operand rows (~190 scalar instructions, three integer divisions) are computed
        // INSIDE the MFMA clusters of k-tile 0 - origin in P1, A rows in P2, W rows in P3 - where the scalar unit has nothing
        // else to do; they are first read by the stages of k-tile 2.  No next tile: the current one again.  The opaque copies
        // pin each piece inside its cluster (the inputs are born there, the results are consumed there).
        const int nid = id + G;
        const bool more = nid < ntiles;
        auto hide_next = [&](auto ph) {
            constexpr int P = decltype(ph)::value;
            if constexpr (P == 1) {
                int t = more ? nid : id;
                asm volatile("" : "+s"(t));
                tile_origin(t, nm0, nn0, nbz);
                asm volatile("" : "+s"(nm0), "+s"(nn0), "+s"(nbz));
            } else if constexpr (P == 2) {
                rows_a(nm0, nbz, nA);
                asm volatile("" : "+s"(nA[0]), "+s"(nA[1]));
            } else if constexpr (P == 3) {
                rows_b(nn0, nbz, nB);
                asm volatile("" : "+s"(nB[0]), "+s"(nB[1]));
            }
        };
        // (Measured and not kept, round 5: pulling the tile's residual rows into L2 two k-tiles ahead of the epilogue with two
        // 4-byte-per-lane LDS-DMA pieces per wave - the +residual shapes did not move, 83.6 vs 81.9 ms per forward,
        // profiles/r05_pp_ex16_pf_ab.txt: their longer tile boundary is the residual ADD's instructions, not the rows' latency.)
        if (nk >= 4) {
            ktile(IC<0>{}, IC<EX_TAIL>{}, IC<EX_TAIL>{}, IC<EX_TAIL>{}, IC<0>{}, 0, hide_next);
            ktile(IC<1>{}, IC<EX_TAIL>{}, IC<0>{}, IC<0>{}, IC<0>{}, 1, nothing);
            for (int kt = 2; kt < nk; kt += 2) {  // nk is even; the last two k-tiles stage the next tile's first two
                ktile(IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<1>{}, kt, nothing);
                ktile(IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<1>{}, kt + 1, nothing);
            }
        } else {  // K = 128: the tile's only two k-tiles already stage the next tile - its rows have to be known up front
            set_tile(more ? nid : id, nm0, nn0, nbz, nA, nB);
            ktile(IC<0>{}, IC<EX_TAIL>{}, IC<EX_TAIL>{}, IC<EX_TAIL>{}, IC<1>{}, 0, nothing);
            ktile(IC<1>{}, IC<EX_TAIL>{}, IC<0>{}, IC<0>{}, IC<1>{}, 1, nothing);
        }
        if (wr == 0) PP_BARRIER();  // balance group 1's extra barrier: both groups run the epilogue side by side

        // ---- epilogue.  D layout (operands swapped): lane holds tile row ..+(lane&31); register r of a fragment is
        // column (r&3) + 8*(r>>2) + 4*(lane>>5) of its 32-column strip.
        // (the lane-derived epilogue constants are rebuilt from an opaque copy so they are not kept live - spilled -
        // across the main loop)
        bool fast = true;
        bool wave_cols_in = true;   // false: this wave's 64 columns lie past N (ragged last tile column): nothing to store
        bool rows_short = false;    // true: some 32-row piece of this wave lies past M (operand-swapped consumer only): fewer stores
        // DBG 128 (ablation build only): NO epilogue - the accumulators are only marked as used, nothing is converted, staged or
        // stored.  time(DBG 0) - time(DBG 128) is everything a perfect overlap of the tile boundary with MFMAs could recover
        // (tools/pp_boundary_ablation.py, profiles/r06_pp_boundary_ablation.txt).
        if constexpr ((DBG & 128) != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
            fast = false;   // -> pad_tail(EX_TAIL) below: no store was issued
        } else {
        const int cm0 = __builtin_amdgcn_readfirstlane(m0), cn0 = __builtin_amdgcn_readfirstlane(n0);
        const int cbz = __builtin_amdgcn_readfirstlane(bz);   // this tile's batch item (set_tile below moves on to the next tile's)
        T* const Cg = reinterpret_cast<T*>(p.C) + (long)cbz * p.sC;
        const T* const Rg = p.residual ? reinterpret_cast<const T*>(p.residual) + (long)cbz * p.sR : nullptr;
        const int lane_e = lane_id();
        const int l31 = lane_e & 31, lhi = lane_e >> 5;
        // Interior tiles (all of them on the UNet's shapes) take a branch-free epilogue: the generic code below tests
        // bias / row bias / bounds per 4 values, and every test became a branch with an s_waitcnt vmcnt(0) behind each
        // bias or residual load - loads, converts and stores ran strictly one after the other (4-8 us per tile, all of
        // it with the matrix pipe idle).  Here the bias values are fetched once per tile, and the residual loads of a
        // 32-row piece are in flight while the previous piece is transposed through LDS and stored.
        // (the fused-LayerNorm instantiations only ever see interior tiles without a row bias: the launcher checks it, and the
        // generic epilogues are not compiled into them - the kernel has no registers to spare for paths it never takes)
        fast = FUSE != 0 ? true : is_fast(cm0, cn0);
        const int nwg = cn0 + (wc >> 1) * 128 + (wc & 1) * 32;  // GEGLU: the wave's hidden strip; its gates 64 columns further
        const int nwp = cn0 + wc * 64;                          // plain: the wave's 64 adjacent columns
        // The bias slice of this tile has been sitting in the first 128 bytes of the wave's transposition tile since the
        // tile's first k-tiles (`stage_bias`): reading it is an LDS read.  (Loaded from memory here, its wait - vmcnt
        // retires in issue order - either drained the next tile's prologue or, requested ahead of it, was turned into a
        // vmcnt(0) by the compiler all the same.)  Kept packed: the accumulators still occupy 128 registers.
        // Fused LayerNorm (consumer): the rank-1 term - mean_m c_n goes into the accumulators as ONE more MFMA per 32 x 32
        // block; rstd_m (the lane's own tile row) becomes the multiplier of the bias fma below.
        if constexpr ((FUSE & 1) != 0) {
            V8 cf[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const auto cc = *reinterpret_cast<const typename Elt<T>::v2*>(ep + 1024 + (ni * 32 + l31) * 4);
                const T z = (T)0.f;
                cf[ni] = V8{cc[0], cc[1], cc[0], z, z, z, z, z};
                if (lhi) cf[ni] = V8{z, z, z, z, z, z, z, z};
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {  // one row block at a time: 4 + 8 fragment registers live beside the accumulators
                const f32x2 st = *reinterpret_cast<const f32x2*>(ep + 2048 + ((mi >> 1) * 64 + (mi & 1) * 32 + l31) * 8);
                const T z = (T)0.f, mh = (T)st[0], ml = (T)(st[0] - (float)mh);
                V8 mf = V8{mh, mh, ml, z, z, z, z, z};
                if (lhi) mf = V8{z, z, z, z, z, z, z, z};
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Elt<T>::mfma(cf[ni], mf, acc[mi][ni]);
                asm volatile("" : "+v"(mf));   // keep the blocks in sequence (the scheduler would hoist all four fragment builds)
            }
            // rstd of the lane's own tile row is the multiplier of the fma that is the bias add otherwise (both epilogues; until
            // round 4 the plain one scaled the accumulators in place first: 64 v_pk_mul_f32 per wave and tile).  The GEGLU
            // transposition only uses the first 2 KiB of `ep`, so it re-reads rstd from ep + 2048 per 32-row piece; the plain
            // one overwrites all 4 KiB and takes the four values into registers here.
        }
        float rs4[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FUSE == 1) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                rs4[mi] = *reinterpret_cast<const float*>(ep + 2048 + ((mi >> 1) * 64 + (mi & 1) * 32 + l31) * 8 + 4);
        }
        if constexpr ((FUSE & 4) != 0) {
            // operand-swapped form: the statistics run along the tile columns (first MFMA operand), c and b' along the rows
            V8 mfc[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float mu = *reinterpret_cast<const float*>(ep + 1024 + (ni * 32 + l31) * 8);
                const T z = (T)0.f, mh = (T)mu, ml = (T)(mu - (float)mh);
                mfc[ni] = V8{mh, mh, ml, z, z, z, z, z};
                if (lhi) mfc[ni] = V8{z, z, z, z, z, z, z, z};
            }
            float brow[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const V4 cb = *reinterpret_cast<const V4*>(ep + 2048 + ((mi >> 1) * 64 + (mi & 1) * 32 + l31) * 8);
                const T z = (T)0.f;
                V8 cfr = V8{cb[0], cb[1], cb[0], z, z, z, z, z};
                if (lhi) cfr = V8{z, z, z, z, z, z, z, z};
                brow[mi] = (float)cb[2] + (float)cb[3];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Elt<T>::mfma(mfc[ni], cfr, acc[mi][ni]);
                asm volatile("" : "+v"(cfr));
            }
            // y = rstd_n acc + b'_m: register r of strip ni is tile column (r & 3) + 8 (r >> 2) + 4 lhi of the strip
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                float rs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rs[r] = *reinterpret_cast<const float*>(ep + 1024 + (ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * 8 + 4);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaf(acc[mi][ni][r], rs[r], brow[mi]);
            }
        }
        V4 bq[8];
        if (fast) {
#pragma unroll
            for (int g = 0; g < 8; ++g) bq[g] = V4{0, 0, 0, 0};
            if (p.bias) {
#pragma unroll
                for (int g = 0; g < 8; ++g) bq[g] = *reinterpret_cast<const V4*>(ep + (g >> 2) * 64 + 2 * (8 * (g & 3) + 4 * lhi));
            }
        }
        PP_FENCE();

        // (the next tile's pipeline fill went out with the last two k-tiles: nothing to issue here)
        if (FUSE != 2 && FUSE != 1 && FUSE != 4 && fast && (geglu || FUSE == 9)) {
            const int no = (cn0 >> 1) + wc * 32;
            // store addresses = (uniform 64-bit base of the piece's rows) + zext(32-bit lane offset): the SGPR-base form of
            // global_store - per-lane 64-bit row * ldc products were 40 % of the branch-free epilogues' VALU instructions
            char* const Cw = reinterpret_cast<char*>(Cg) + ((long)(cm0 + wr * 64) * p.ldc + no) * 2;
            const unsigned vC = (unsigned)(((lane_e >> 2) * p.ldc + (lane_e & 3) * 8) * 2);
            const unsigned vCd = (unsigned)((l31 * p.ldc + lhi * 8) * 2);   // PP_DIRECT_EPI: the lane's own row, its half of a 16-column pair
            (void)vC, (void)vCd;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                unsigned og[4][2];
                (void)og;
                float rs = 1.0f;
                if constexpr ((FUSE & 8) != 0)
                    rs = *reinterpret_cast<const float*>(ep + 2048 + ((mi >> 1) * 64 + (mi & 1) * 32 + l31) * 8 + 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 8 * g + 4 * lhi;
                    V4 o;
                    float hq[4], gq[4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        // The sums are formed in f32 and rounded to f16 in a SEPARATE instruction, pair by pair: left alone, the
                        // compiler folds some of these (not all - it depends on the registers at hand) into v_fma_mixlo_f16,
                        // which rounds the exact sum once, and a row's bits then depend on the accumulator slot it sits in -
                        // found by tools/replicate_determinism.py (profiles/r04_determinism_bisect.txt).  The empty asm pins
                        // the f32 pair in registers between the two steps.
                        f32x2 hs, gs;
                        if constexpr ((FUSE & 8) != 0) {
                            const f32x2 r2 = {rs, rs};
                            hs = __builtin_elementwise_fma(f32x2{acc[mi][0][4 * g + e], acc[mi][0][4 * g + e + 1]}, r2,
                                                           f32x2{(float)bq[g][e], (float)bq[g][e + 1]});
                            gs = __builtin_elementwise_fma(f32x2{acc[mi][1][4 * g + e], acc[mi][1][4 * g + e + 1]}, r2,
                                                           f32x2{(float)bq[4 + g][e], (float)bq[4 + g][e + 1]});
                        } else {
                            hs = f32x2{acc[mi][0][4 * g + e], acc[mi][0][4 * g + e + 1]} + f32x2{(float)bq[g][e], (float)bq[g][e + 1]};
                            gs = f32x2{acc[mi][1][4 * g + e], acc[mi][1][4 * g + e + 1]} + f32x2{(float)bq[4 + g][e], (float)bq[4 + g][e + 1]};
                        }
                        asm("" : "+v"(hs), "+v"(gs));
                        hq[e] = (float)(T)hs[0], hq[e + 1] = (float)(T)hs[1];
                        gq[e] = (float)(T)gs[0], gq[e + 1] = (float)(T)gs[1];
                    }
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 ge = ds_gelu_erf2(f32x2{gq[e], gq[e + 1]});
                        o[e] = (T)(hq[e] * (float)(T)ge[0]);
                        o[e + 1] = (T)(hq[e + 1] * (float)(T)ge[1]);
                    }
#if PP_DIRECT_EPI
                    {
                        const u32x2 t = __builtin_bit_cast(u32x2, o);
                        og[g][0] = t[0], og[g][1] = t[1];
                    }
                }
                {
                    char* const Cr = Cw + (long)((mi >> 1) * 128 + (mi & 1) * 32) * p.ldc * 2;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        swap32(og[2 * j][0], og[2 * j + 1][0]);
                        swap32(og[2 * j][1], og[2 * j + 1][1]);
                        const u32x4 v = {og[2 * j][0], og[2 * j][1], og[2 * j + 1][0], og[2 * j + 1][1]};
                        *reinterpret_cast<u32x4*>(Cr + (size_t)vCd + j * 32) = v;
                    }
                }
#else
                    *reinterpret_cast<V4*>(ep + l31 * 64 + ((((c >> 3) ^ (l31 >> 2)) & 3) << 4) + ((c >> 2) & 1) * 8) = o;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 16 + (lane_e >> 2), ch = lane_e & 3;
                    const V8 v = *reinterpret_cast<const V8*>(ep + row * 64 + (((ch ^ (row >> 2)) & 3) << 4));
                    char* const Cr = Cw + (long)((mi >> 1) * 128 + (mi & 1) * 32 + i * 16) * p.ldc * 2;
                    *reinterpret_cast<V8*>(Cr + (size_t)vC) = v;
                }
#endif
            }
        } else if (FUSE != 9 && fast) {
            // two straight-line instances (with / without a residual): one body with `if (Rg)` inside carried the
            // never-written residual registers of the other case around the whole tile loop as spills
            auto plain = [&](auto resc) {
                constexpr bool RES = decltype(resc)::value != 0;
                V8 rv[4];
                // addresses = (uniform 64-bit base of the rows) + zext(32-bit lane offset), as in the GEGLU epilogue above
                char* const Cw = reinterpret_cast<char*>(Cg) + ((long)(cm0 + wr * 64) * p.ldc + nwp) * 2;
                const char* const Rw = RES ? reinterpret_cast<const char*>(Rg) + ((long)(cm0 + wr * 64) * p.ldr + nwp) * 2 : nullptr;
                const unsigned vC = (unsigned)(((lane_e >> 3) * p.ldc + (lane_e & 7) * 8) * 2);
                const unsigned vR = (unsigned)(((lane_e >> 3) * p.ldr + (lane_e & 7) * 8) * 2);
                auto load_res = [&](int mi) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const char* const Rr = Rw + (long)((mi >> 1) * 128 + (mi & 1) * 32 + i * 8) * p.ldr * 2;
                        rv[i] = *reinterpret_cast<const V8*>(Rr + (size_t)vR);
                    }
                };
                if constexpr (RES) load_res(0);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int mb = cm0 + (mi >> 1) * 128 + wr * 64 + (mi & 1) * 32;
                    if constexpr (FUSE == 4) {
                        // Operand-swapped consumer (V^T = Wv LN(X_b)^T): the tile ROWS are the output channels, 640 = 2.5 tiles at
                        // the 640-channel level (round 6).  A 32-row piece is wholly inside M or wholly outside (M % 32 == 0,
                        // wave-uniform); an outside piece stores nothing, the hand-over pads the wave's vector-memory count.
                        if (mb >= p.M) {
                            rows_short = true;
                            continue;
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = ni * 32 + 8 * g + 4 * lhi;
                            V4 o;
                            if constexpr (FUSE == 1) {
                                // y = rstd (acc - mean c) + b': one fma per pair, the f32 pair pinned before it is rounded (the
                                // compiler otherwise folds SOME of these into v_fma_mixlo_f16 - one rounding instead of two - and a
                                // row's bits would depend on the accumulator slot it sits in: profiles/r04_determinism_bisect.txt)
                                const f32x2 r2 = {rs4[mi], rs4[mi]};
#pragma unroll
                                for (int e = 0; e < 4; e += 2) {
                                    f32x2 y = __builtin_elementwise_fma(f32x2{acc[mi][ni][4 * g + e], acc[mi][ni][4 * g + e + 1]}, r2,
                                                                        f32x2{(float)bq[ni * 4 + g][e], (float)bq[ni * 4 + g][e + 1]});
                                    asm("" : "+v"(y));
                                    o[e] = (T)y[0], o[e + 1] = (T)y[1];
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = (T)(acc[mi][ni][4 * g + e] + (float)bq[ni * 4 + g][e]);
                            }
                            *reinterpret_cast<V4*>(ep + l31 * 128 + ((((c >> 3) ^ (l31 >> 1)) & 7) << 4) + ((c >> 2) & 1) * 8) = o;
                        }
                    V8 v[4];
                    float sv = 0.f, qv = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = i * 8 + (lane_e >> 3), ch = lane_e & 7;
                        v[i] = *reinterpret_cast<const V8*>(ep + row * 128 + (((ch ^ (row >> 1)) & 7) << 4));
                        if constexpr (RES) {
                            // f16 + f16, rounded once: v_pk_add_f16, four instructions per 8 values.  Until round 4 this was
                            // (T)((float)a + (float)b) - 24 instructions - which is the SAME number: the f32 sum of two f16 values
                            // is exact unless their exponents are >= 13 apart, and then the small one is far below the large
                            // one's half-ulp: no double rounding (checked for all 4.0e9 pairs of finite f16,
                            // tests/test_f16_add_equivalence.py).  bf16 (VAE) keeps the f32 form.
                            if constexpr (std::is_same<T, half_t>::value) v[i] = v[i] + rv[i];
                            else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[i][e] = (T)((float)v[i][e] + (float)rv[i][e]);
                            }
                        }
                        // Fused LayerNorm (producer): (sum, sum of squares) of the 64 values this wave stores per row.  Row
                        // i * 8 + (lane >> 3) of the piece is spread over the 8 lanes lane & 7: per lane two dot instructions per
                        // f16 pair (v_dot2c_f32_f16: x . (1, 1) and x . x, f32 accumulate; round 4 converted, added and fma'd
                        // value by value: 24 instructions per 8 values instead of 8), then a butterfly over the 8 lanes with DPP
                        // (quad_perm xor 1, xor 2, then row_half_mirror pairs lane j with 7 - j), after which lane & 7 == i
                        // keeps row group i: one 8-byte store per row from 32 lanes, 256 contiguous bytes per piece.
                        if constexpr ((FUSE & 2) != 0) {
                            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                            float s1 = 0.f, q1 = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const h2v pr = {v[i][e], v[i][e + 1]};
                                s1 = __builtin_amdgcn_fdot2(pr, h2v{(_Float16)1.f, (_Float16)1.f}, s1, false);
                                q1 = __builtin_amdgcn_fdot2(pr, pr, q1, false);
                            }
                            s1 += dpp_f32<0xB1>(s1);
                            q1 += dpp_f32<0xB1>(q1);
                            s1 += dpp_f32<0x4E>(s1);
                            q1 += dpp_f32<0x4E>(q1);
                            s1 += dpp_f32<0x141>(s1);
                            q1 += dpp_f32<0x141>(q1);
                            if ((lane_e & 7) == i) sv = s1, qv = q1;
                        }
                    }
                    // the NEXT piece's residual rows are requested before this piece's stores: the wait for them then
                    // leaves the stores in flight (requested behind them, it would wait for their acknowledgement)
                    if constexpr (RES) {
                        if (mi < 3) load_res(mi + 1);
                    }
                    if constexpr ((FUSE & 2) != 0) {
                        if ((lane_e & 7) < 4) {
                            const int row = (lane_e & 7) * 8 + (lane_e >> 3);
                            f32x2 o2 = {sv, qv};
                            *reinterpret_cast<f32x2*>(p.stats_out + 2 * ((long)(nwp >> 6) * p.M + mb + row)) = o2;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        char* const Cr = Cw + (long)((mi >> 1) * 128 + (mi & 1) * 32 + i * 8) * p.ldc * 2;
                        *reinterpret_cast<V8*>(Cr + (size_t)vC) = v[i];
                    }
                }
            };
            // (The same hand-over WITHOUT the LDS transposition - v_permlane32_swap + row-per-lane 16-byte stores, as in the GEGLU
            // epilogue above - was built and measured for these plain tiles too: 128-byte rows leave as four 32-byte pieces from four
            // instructions and the residual rows arrive the same way: +4 % on the plain shapes, +15 % with a residual
            // (profiles/r05_pp_direct_epilogue_ab.txt).  Only the 64-byte rows of the GEGLU tiles go direct.)
            wave_cols_in = nwp < p.N;   // wave-uniform (N % 64 == 0: all of the wave's 64 columns or none)
            if (!wave_cols_in) {
            } else if (__builtin_amdgcn_readfirstlane((int)(Rg != nullptr))) plain(IC<1>{});
            else plain(IC<0>{});
        } else if constexpr (FUSE != 0) {
        } else if (geglu) {
            const int nw = cn0 + (wc >> 1) * 128 + (wc & 1) * 32;  // hidden strip; gates 64 columns further
            const int no = (cn0 >> 1) + wc * 32;                   // first output column of the wave
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int mb = cm0 + (mi >> 1) * 128 + wr * 64 + (mi & 1) * 32;
                const int ml = mb + l31;
                const int grp = p.rowbias ? ((ml < p.M ? ml : p.M - 1) / p.rows_per_group) : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 8 * g + 4 * lhi;  // column within the 32-column strip
                    float hv[4], gv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hv[e] = acc[mi][0][4 * g + e];
                        gv[e] = acc[mi][1][4 * g + e];
                    }
                    if (nw + c < p.N) {
                        if (p.bias) {
                            const V4 b0 = *reinterpret_cast<const V4*>(p.bias + nw + c);
                            const V4 b1 = *reinterpret_cast<const V4*>(p.bias + nw + 64 + c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) hv[e] += (float)b0[e], gv[e] += (float)b1[e];
                        }
                        if (p.rowbias) {
                            const half_t* rb = p.rowbias + (long)grp * p.rowbias_ld + nw + c;
                            const V4 b0 = *reinterpret_cast<const V4*>(rb);
                            const V4 b1 = *reinterpret_cast<const V4*>(rb + 64);
#pragma unroll
                            for (int e = 0; e < 4; ++e) hv[e] += (float)b0[e], gv[e] += (float)b1[e];
                        }
                    }
                    V4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hq = (float)(T)hv[e], gq = (float)(T)gv[e];
                        if constexpr ((DBG & 64) != 0) o[e] = (T)(hq * gq);  // ablation: no erf
                        else o[e] = (T)(hq * (float)(T)ds_gelu_erf(gq));
                    }
                    // 64-byte rows: chunk (c>>3) of row l31 sits at slot chunk ^ ((row>>2)&3)
                    *reinterpret_cast<V4*>(ep + l31 * 64 + ((((c >> 3) ^ (l31 >> 2)) & 3) << 4) + ((c >> 2) & 1) * 8) = o;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 16 + (lane_e >> 2), ch = lane_e & 3;
                    const V8 v = *reinterpret_cast<const V8*>(ep + row * 64 + (((ch ^ (row >> 2)) & 3) << 4));
                    const int m = mb + row, n = no + ch * 8;
                    if (m < p.M && n < (p.N >> 1)) *reinterpret_cast<V8*>(Cg + (long)m * p.ldc + n) = v;
                }
            }
        } else {
            const int nw = cn0 + wc * 64;  // 64 adjacent columns: strip ni at +32*ni
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int mb = cm0 + (mi >> 1) * 128 + wr * 64 + (mi & 1) * 32;
                const int ml = mb + l31;
                const int grp = p.rowbias ? ((ml < p.M ? ml : p.M - 1) / p.rows_per_group) : 0;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = ni * 32 + 8 * g + 4 * lhi;  // column within the wave's 64
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
                        if (nw + c < p.N) {
                            if (p.bias) {
                                const V4 bv = *reinterpret_cast<const V4*>(p.bias + nw + c);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
                            }
                            if (p.rowbias) {
                                const V4 bv = *reinterpret_cast<const V4*>(p.rowbias + (long)grp * p.rowbias_ld + nw + c);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
                            }
                        }
                        V4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
                        // 128-byte rows: chunk (c>>3) of row l31 sits at slot chunk ^ ((row>>1)&7)
                        *reinterpret_cast<V4*>(ep + l31 * 128 + ((((c >> 3) ^ (l31 >> 1)) & 7) << 4) + ((c >> 2) & 1) * 8) = o;
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane_e >> 3), ch = lane_e & 7;
                    V8 v = *reinterpret_cast<const V8*>(ep + row * 128 + (((ch ^ (row >> 1)) & 7) << 4));
                    const int m = mb + row, n = nw + ch * 8;
                    if (m < p.M && n < p.N) {
                        if (p.epi == EPI_GELU) {
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const f32x2 ge = ds_gelu_erf2(f32x2{(float)v[e], (float)v[e + 1]});
                                v[e] = (T)ge[0];
                                v[e + 1] = (T)ge[1];
                            }
                        } else if (p.epi == EPI_QUICK_GELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float f = (float)v[e];
                                v[e] = (T)(f / (1.0f + __expf(-1.702f * f)));
                            }
                        }
                        if (Rg) {
                            const V8 rv = *reinterpret_cast<const V8*>(Rg + (long)m * p.ldr + n);
                            if constexpr (std::is_same<T, half_t>::value) v = v + rv;
                            else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (T)((float)v[e] + (float)rv[e]);
                            }
                        }
                        *reinterpret_cast<V8*>(Cg + (long)m * p.ldc + n) = v;
                    }
                }
            }
        }
        }   // DBG 128
        if (!more) break;
        // ---- hand over to the next tile: its coordinates become the current ones, its bias / LayerNorm pieces go out behind
        // this tile's C stores (the previous contents of `ep` have been consumed), the lane-derived staging and fragment offsets
        // are rebuilt (not kept live across the epilogue: 13 registers the epilogue needs)
        id = nid;
        m0 = nm0, n0 = nn0, bz = nbz;
        gA[0] = nA[0], gA[1] = nA[1], gB[0] = nB[0], gB[1] = nB[1];
        if (p.bias && is_fast(m0, n0)) stage_bias(n0);
        if constexpr ((FUSE & 5) != 0) stage_ln(m0, n0, bz);
        if (!fast || !wave_cols_in || rows_short) pad_tail(IC<EX_TAIL>{});   // generic epilogue: its store count depends on the tile's edges; a wave outside a ragged column: no store
        else if (FUSE == 0 && geglu) pad_tail(IC<(EX_TAIL > 8 ? EX_TAIL - 8 : 0)>{});   // unfused GEGLU tile: 8 stores
        derive_stage();
        derive_frag();
    }
    if constexpr ((DBG & 16) != 0) {
        if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.C);
            __builtin_amdgcn_s_waitcnt(0);
            o[0] = __builtin_readcyclecounter() - probe_c0;
            o[1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
        }
    }
}

int g_pp_blocks = 0;  // persistent grid size: one block per CU
// Persistent grid = ceil(tiles / rounds) blocks (rounded up to a multiple of 8 for the XCD walk) instead of one per CU: the
// same number of rounds, but every round full.  Measured (profiles/r02_pp_even_ab.txt): M = 32768 x N = 1280 = 640 tiles =
// 2.5 rounds of 256 CUs runs 9-10 % FASTER on 216 CUs x 3 full rounds (FF down-projection 449 -> 408 us, 957 -> 1053
// TF/s; the N = K = 1280 projections 137 -> 126 us); shapes with whole rounds are unchanged.  A half-empty last round
// leaves the stragglers' operand panels without the sharers the XCD-chunked walk counts on, and on this power-limited
// part the idle CUs buy nothing.  Knob "gemm_pp_even" 0 restores one block per CU (A/B).
thread_local int g_pp_even = 1;

}  // namespace

void ds_gemm_pp_set_even(int v) { g_pp_even = v; }

// Shapes the kernel takes: K a multiple of 128 (an even number of k-tiles), a single A source, M and N multiples of 16 (a wave's 16
// staged rows are then wholly inside or outside the problem).  ds_launch_gemm decides when it is also the faster choice.
bool ds_gemm_pp_applicable(const GemmParams& p) {
    return p.conv == 0 && p.A2 == nullptr && p.M % 16 == 0 && p.N % 16 == 0 && p.K % 128 == 0 &&
           (p.epi != EPI_GEGLU || p.N % 128 == 0) && p.lda * 15 + 64 < (1L << 30) && p.ldw * 15 + 64 < (1L << 30);
}

int ds_launch_gemm_pp(const GemmParams& p0, int batch, hipStream_t stream) {
    GemmParams p = p0;
    DS_REQUIRE(ds_gemm_pp_applicable(p), "gemm_pp: shape M=%d N=%d K=%d not supported", p.M, p.N, p.K);
    if (p.ln_stats || p.ln_c || p.stats_out) {  // fused LayerNorm: only the branch-free epilogues implement it
        DS_REQUIRE((p.ln_swapped ? p.M % 32 == 0 : p.M % 256 == 0) &&
                       (p.N % 256 == 0 || (p.N % 64 == 0 && p.epi == EPI_NONE && !p.ln_swapped && (p.debug & 4096) == 0)) &&
                       !p.rowbias && (p.epi == EPI_NONE || p.epi == EPI_GEGLU),
                   "gemm_pp: fused LayerNorm needs whole 256-row tiles (operand-swapped: whole 32-row pieces), whole 256-column tiles (plain epilogue: whole 64-column strips) and no row bias (M=%d N=%d epi=%d)", p.M, p.N, p.epi);
        DS_REQUIRE(batch == 1 || (p.ln_swapped && p.ln_stats), "gemm_pp: only the operand-swapped fused consumer is batched");
        DS_REQUIRE((p.ln_stats != nullptr) == (p.ln_c != nullptr), "gemm_pp: ln_stats and ln_c come as a pair");
        DS_REQUIRE(!p.ln_stats || p.ln_swapped || p.bias, "gemm_pp: the fused-LayerNorm consumer takes b' = bias + W beta as its bias");
        DS_REQUIRE(!p.ln_swapped || (p.ln_stats && !p.bias && !p.residual && p.epi == EPI_NONE),
                   "gemm_pp: the operand-swapped fused consumer carries b' in ln_c and takes no bias / residual / activation");
        DS_REQUIRE(!p.stats_out || p.epi == EPI_NONE, "gemm_pp: row statistics are emitted by the plain epilogue only");
        DS_REQUIRE(p.dtype == DS_DTYPE_F16, "gemm_pp: fused LayerNorm is an f16 path");
        DS_REQUIRE(!(p.ln_stats && p.stats_out), "gemm_pp: one launch is either the consumer or the producer of a fused LayerNorm");
    }
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    const size_t lds = 8 * HT + 8 * 4096;
    typedef void (*kern_t)(const GemmParams);
    // production build: the plain kernel per element type.  The ablation builds (gemm_debug switches, tools/ablate_pp.py)
    // are only instantiated with -DDS_ABLATION (python -m diffsensei_amd.build --ablation).
    static const struct { int dbg; kern_t k; } table[] = {
        {0, gemm_pp_kernel<half_t, 0>},
#ifdef DS_ABLATION
        {1, gemm_pp_kernel<half_t, 1>},   {2, gemm_pp_kernel<half_t, 2>},   {3, gemm_pp_kernel<half_t, 3>},
        {4, gemm_pp_kernel<half_t, 4>},   {6, gemm_pp_kernel<half_t, 6>},   {8, gemm_pp_kernel<half_t, 8>},   {16, gemm_pp_kernel<half_t, 16>},
        {17, gemm_pp_kernel<half_t, 17>}, {18, gemm_pp_kernel<half_t, 18>}, {20, gemm_pp_kernel<half_t, 20>}, {22, gemm_pp_kernel<half_t, 22>},
        {24, gemm_pp_kernel<half_t, 24>}, {32, gemm_pp_kernel<half_t, 32>}, {48, gemm_pp_kernel<half_t, 48>}, {49, gemm_pp_kernel<half_t, 49>}, {64, gemm_pp_kernel<half_t, 64>}, {128, gemm_pp_kernel<half_t, 128>},
#endif
        {-2, gemm_pp_kernel<half_t, 0, 1>},  // -2 / -5 / -3: fused LayerNorm, consumer (plain / GEGLU epilogue) / producer
        {-5, gemm_pp_kernel<half_t, 0, 9>},
        {-3, gemm_pp_kernel<half_t, 0, 2>},
        {-4, gemm_pp_kernel<half_t, 0, 4>},  // consumer in the operand-swapped (V^T) form
        {-1, gemm_pp_kernel<bf16_t, 0>}};  // -1: the bf16 build (VAE decoder), no ablation variants
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        for (const auto& e : table)
            DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0, cus = 0;
        DS_HIP(hipGetDevice(&dev));
        DS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        g_pp_blocks = cus > 0 ? cus : 256;
    }
    p.nbatch = batch;
    const int tiles = p.tiles_m * p.tiles_n * batch;
    int nblk = tiles < g_pp_blocks ? tiles : g_pp_blocks;
    if (g_pp_even && tiles > g_pp_blocks) {  // same number of rounds, all of them full, on fewer CUs
        const int rounds = (tiles + g_pp_blocks - 1) / g_pp_blocks;
        nblk = (tiles + rounds - 1) / rounds;
        nblk = (nblk + 7) / 8 * 8;           // keep the XCD-chunked tile walk (grid % 8 == 0)
        if (nblk > g_pp_blocks) nblk = g_pp_blocks;
    }
    dim3 grid(nblk, 1, 1);
    kern_t kern = nullptr;
    for (const auto& e : table)
        if (e.dbg == (p.dtype == DS_DTYPE_BF16 ? -1 : p.ln_stats ? (p.ln_swapped ? -4 : p.epi == EPI_GEGLU ? -5 : -2) : p.stats_out ? -3 : (p.debug & 255))) kern = e.k;
    DS_REQUIRE(kern != nullptr, "gemm_pp: no ablation build for gemm_debug=%d", p.debug);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

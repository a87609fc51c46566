#!/usr/bin/env python
"""Index-algebra check of the experimental 4-wave GEMM (diffsensei_amd/csrc/experimental/gemm_w4.hip) without a GPU.

The kernel could not be run when it was written, so its address arithmetic - staging swizzle, fragment reads, the MFMA
operand / accumulator layout every other kernel of this repository relies on, the epilogue's transposition through LDS and
its store addresses - is replayed here per thread in numpy on one 256 x 256 tile (K = 128: two k-tiles, both LDS stages) and
compared with x @ w.T + bias (+ residual), and with the GEGLU pairing on packed weights.  It proves the formulas consistent
with each other and with the 32x32x16 MFMA layout; it says nothing about timing or about what the compiler does with them.

    python tools/w4_index_model.py
"""
import numpy as np

STAGE, BOFF, EP = 65536, 32768, 131072


def mfma_32x32x16(x_frag, y_frag, acc):
    """x_frag / y_frag: [64 lanes, 8] halfs of the first / second operand; acc: [64, 16] f32.
    D[i][j] += sum_k X[i][k] Y[k][j]; lane L gives X[L & 31][8 (L >> 5) + 0..7] and Y[8 (L >> 5) + 0..7][L & 31];
    lane L receives D[(r & 3) + 8 (r >> 2) + 4 (L >> 5)][L & 31] in register r."""
    X = np.zeros((32, 16), np.float32)
    Y = np.zeros((16, 32), np.float32)
    for L in range(64):
        X[L & 31, 8 * (L >> 5):8 * (L >> 5) + 8] = x_frag[L]
        Y[8 * (L >> 5):8 * (L >> 5) + 8, L & 31] = y_frag[L]
    D = X @ Y
    for L in range(64):
        for r in range(16):
            acc[L, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (L >> 5), L & 31]


def run_tile(A, W, bias, residual, geglu):
    """One block: tile (0, 0) of C = A @ W.T (+ bias, + residual | GEGLU); A [256, K], W [256, K] f16."""
    K = A.shape[1]
    nk = K // 64
    lds = np.zeros(163840 // 2, np.float16)          # indexed in halfs

    def lds_w(byte, v8):
        lds[byte // 2:byte // 2 + 8] = v8

    def lds_r(byte):
        return lds[byte // 2:byte // 2 + 8].copy()

    tids = np.arange(256)
    lrow, chunk = tids >> 3, tids & 7
    w0 = lrow * 128 + ((chunk ^ ((lrow >> 1) & 7)) << 4)

    def stage_ktile(stage, kt):                       # gload + lwrite of every thread
        for t in range(256):
            for j in range(8):
                r = j * 32 + lrow[t]
                lds_w(stage * STAGE + j * 4096 + w0[t], A[r, kt * 64 + chunk[t] * 8:kt * 64 + chunk[t] * 8 + 8])
                lds_w(stage * STAGE + BOFF + j * 4096 + w0[t], W[r, kt * 64 + chunk[t] * 8:kt * 64 + chunk[t] * 8 + 8])

    acc = np.zeros((4, 4, 4, 64, 16), np.float32)     # [wave][mi][ni][lane][r]
    for kt in range(nk):
        stage = kt & 1
        stage_ktile(stage, kt)
        for wave in range(4):
            wr, wc = wave >> 1, wave & 1
            for kk in range(4):
                a = np.zeros((4, 64, 8), np.float16)
                b = np.zeros((4, 64, 8), np.float16)
                for lane in range(64):
                    l31, lhi = lane & 31, lane >> 5
                    o = l31 * 128 + (((kk * 2 + lhi) ^ ((l31 >> 1) & 7)) << 4)
                    for i in range(4):
                        a[i, lane] = lds_r(stage * STAGE + wr * 16384 + o + i * 4096)
                        b[i, lane] = lds_r(stage * STAGE + BOFF + wc * 16384 + o + i * 4096)
                for mi in range(4):
                    for ni in range(4):
                        mfma_32x32x16(b[ni].astype(np.float32), a[mi].astype(np.float32), acc[wave, mi, ni])

    no_blocks = 2 if geglu else 4
    rowb = no_blocks * 64
    C = np.zeros((256, 128 if geglu else 256), np.float16)
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        m0, n0 = wr * 128, wc * 128
        no = n0 >> 1 if geglu else n0
        ep = EP + wave * 8192
        for mi in range(4):
            for lane in range(64):
                l31, lhi = lane & 31, lane >> 5
                for ni in range(no_blocks):
                    for g in range(4):
                        o = np.zeros(4, np.float16)
                        for e in range(4):
                            col = ni * 32 + 8 * g + 4 * lhi + e
                            if geglu:
                                h = np.float16(acc[wave, mi, ni, lane, 4 * g + e] + np.float32(bias[n0 + col]))
                                gt = np.float16(acc[wave, mi, ni + 2, lane, 4 * g + e] + np.float32(bias[n0 + 64 + col]))
                                from math import erf, sqrt
                                gel = np.float16(0.5 * float(gt) * (1.0 + erf(float(gt) / sqrt(2.0))))
                                o[e] = np.float16(np.float32(h) * np.float32(gel))
                            else:
                                o[e] = np.float16(acc[wave, mi, ni, lane, 4 * g + e] + np.float32(bias[n0 + col]))
                        c16 = ni * 4 + g
                        byte = ep + l31 * rowb + ((c16 ^ (l31 & (no_blocks * 4 - 1))) << 4) + lhi * 8
                        lds[byte // 2:byte // 2 + 4] = o
            cpr = no_blocks * 4
            rpi = 64 // cpr
            for i in range(32 // rpi):
                for lane in range(64):
                    row, ch = i * rpi + lane // cpr, lane % cpr
                    v = lds_r(ep + row * rowb + ((ch ^ (row & (cpr - 1))) << 4)).astype(np.float32)
                    m = m0 + mi * 32 + row
                    if residual is not None and not geglu:
                        v = np.float16(v).astype(np.float32) + residual[m, no + ch * 8:no + ch * 8 + 8].astype(np.float32)
                    C[m, no + ch * 8:no + ch * 8 + 8] = v.astype(np.float16)
    return C


def main():
    rng = np.random.default_rng(0)
    K = 128
    A = (rng.standard_normal((256, K)) * 0.5).astype(np.float16)
    W = (rng.standard_normal((256, K)) * K ** -0.5).astype(np.float16)
    bias = rng.standard_normal(256).astype(np.float16)
    res = rng.standard_normal((256, 256)).astype(np.float16)
    y = A.astype(np.float32) @ W.astype(np.float32).T + bias.astype(np.float32)
    got = run_tile(A, W, bias, res, geglu=False).astype(np.float32)
    ref = y.astype(np.float16).astype(np.float32) + res.astype(np.float32)
    err = np.abs(got - ref).max()
    print("plain + bias + residual: max |diff| vs numpy", err)
    assert err <= 2e-2, err
    # GEGLU on packed weights: every 128 columns = 64 hidden | their 64 gates
    from math import erf, sqrt
    t = y.astype(np.float16).astype(np.float32).reshape(256, 2, 2, 64)
    gel = np.vectorize(lambda v: 0.5 * v * (1.0 + erf(v / sqrt(2.0))))(t[:, :, 1]).astype(np.float16).astype(np.float32)
    refg = (t[:, :, 0] * gel).reshape(256, 128)
    gotg = run_tile(A, W, bias, None, geglu=True).astype(np.float32)
    errg = np.abs(gotg - refg).max()
    print("GEGLU: max |diff| vs numpy", errg)
    assert errg <= 2e-2, errg
    print("index algebra consistent")


if __name__ == "__main__":
    main()

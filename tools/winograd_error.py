#!/usr/bin/env python3
"""tools/winograd_error.py -- the numbers behind the F(6,3) decision: fp32 error of the Winograd transforms
F(2,3), F(4,3), F(6,3) on the layer shapes of the path, and their transform-operation counts per matrix-core operation.
CPU only (numpy); the matrices are the standard Cook-Toom ones (Lavin & Gray) and are verified against direct convolution in
float64 before anything is measured.

    python tools/winograd_error.py
"""
import numpy as np

F = np.float64

MATS = {
    2: dict(
        AT=[[1, 1, 1, 0], [0, 1, -1, -1]],
        G=[[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]],
        BT=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]),
    4: dict(
        AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
        G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
        BT=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]),
    6: dict(
        AT=[[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0],
            [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 0], [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]],
        G=[[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
           [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]],
        BT=[[1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
            [0, 1 / 2, 1 / 4, -5 / 2, -5 / 4, 2, 1, 0], [0, -1 / 2, 1 / 4, 5 / 2, -5 / 4, -2, 1, 0], [0, 2, 4, -5 / 2, -5, 1 / 2, 1, 0],
            [0, -2, 4, 5 / 2, -5, -1 / 2, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]]),
}


def direct(x, w):
    """x [H, W, C] (already padded), w [OC, C, 3, 3] -> [H-2, W-2, OC] in the dtype of x"""
    H, W, C = x.shape
    out = np.zeros((H - 2, W - 2, w.shape[0]), x.dtype)
    for ky in range(3):
        for kx in range(3):
            out += x[ky:ky + H - 2, kx:kx + W - 2, :] @ w[:, :, ky, kx].T
    return out


def winograd(x, w, m, dt):
    """F(m x m, 3 x 3) with every product and sum rounded to `dt` (transforms as dense matrix products in dt, the plane GEMMs
    accumulated in dt like a k-ordered fma chain would)."""
    AT, G, BT = (np.array(MATS[m][k], dt) for k in ("AT", "G", "BT"))
    a = m + 2
    H, W, C = x.shape
    oh, ow = H - 2, W - 2
    assert oh % m == 0 and ow % m == 0
    x = x.astype(dt)
    U = np.einsum("ij,ocjk,lk->iloc", G, w.astype(dt), G).astype(dt)           # [a, a, OC, C]
    out = np.zeros((oh, ow, w.shape[0]), dt)
    for ty in range(oh // m):
        for tx in range(ow // m):
            d = x[ty * m:ty * m + a, tx * m:tx * m + a, :]                       # [a, a, C]
            V = np.einsum("ij,jkc,lk->ilc", BT, d, BT).astype(dt)
            M = np.einsum("iloc,ilc->ilo", U, V).astype(dt)                       # 'a*a' plane GEMMs
            out[ty * m:(ty + 1) * m, tx * m:(tx + 1) * m, :] = np.einsum("ij,jko,lk->ilo", AT, M, AT).astype(dt)
    return out


def op_counts(m):
    """VALU operations of the two data transforms per tile and channel, counted from the matrices: one op per non-zero beyond the
    first in a row (adds / fmas; multiplications by +-1 are free), applied to every row or column of the tile twice."""
    AT, BT = np.array(MATS[m]["AT"]), np.array(MATS[m]["BT"])
    a = m + 2
    def row_ops(Mx):
        return int(sum(max(int(np.count_nonzero(r)) - 1, 0) + int(np.count_nonzero((np.abs(r) != 1) & (r != 0)) > 0 and 0) for r in Mx))
    def fma_ops(Mx):   # every non-zero is one multiply-add (or add) except the first of a row
        return int(sum(max(int(np.count_nonzero(r)) - 1, 0) for r in Mx))
    in_ops = fma_ops(BT) * a + fma_ops(BT) * a          # B^T d (a columns), then (.) B (a rows)
    out_ops = fma_ops(AT) * a + fma_ops(AT) * m         # A^T M (a columns), then (.) A (m rows)
    return in_ops, out_ops


def main():
    rng = np.random.default_rng(0)
    print("== correctness of the matrices (float64, 12x12 outputs, 8 channels -> 4)")
    x = rng.random((14, 14, 8)); w = rng.random((4, 8, 3, 3)) - 0.5
    ref = direct(x, w)
    for m in (2, 4, 6):
        e = np.abs(winograd(x, w, m, np.float64) - ref).max() / np.abs(ref).max()
        print("  F(%d,3): max|diff|/max|ref| = %.2e" % (m, e))
        assert e < 1e-12
    print("== fp32 error against float64 direct convolution, inputs U[0,1) (what /255 images and SiLU outputs look like), weights U[-a,a], a = sqrt(3/fan_in)")
    print("   %-22s %12s %12s %12s   (max|diff| / max|ref|; the parity bar of the path is 1e-4)" % ("shape (C -> OC)", "F(2,3)", "F(4,3)", "F(6,3)"))
    for C, OC in ((32, 32), (64, 64), (128, 128), (256, 256), (512, 64)):
        x = rng.random((14, 14, C)); a_ = np.sqrt(3.0 / (9 * C)); w = (rng.random((OC, C, 3, 3)) * 2 - 1) * a_
        ref = direct(x, w)
        errs = [np.abs(winograd(x, w, m, np.float32).astype(F) - ref).max() / np.abs(ref).max() for m in (2, 4, 6)]
        d32 = np.abs(direct(x.astype(np.float32), w.astype(np.float32)).astype(F) - ref).max() / np.abs(ref).max()
        print("   %-22s %12.2e %12.2e %12.2e   direct fp32: %.1e" % ("%d -> %d" % (C, OC), errs[0], errs[1], errs[2], d32))
    print("== transform work per matrix-core work (per tile, per input channel c and output channel o)")
    print("   %-8s %6s %10s %10s %16s %22s" % ("", "planes", "in ops/c", "out ops/o", "MFMA flops/(c,o)", "direct flops / MFMA flops"))
    for m in (2, 4, 6):
        a = m + 2
        i_ops, o_ops = op_counts(m)
        print("   F(%d,3)   %6d %10d %10d %16d %22.2f" % (m, a * a, i_ops, o_ops, 2 * a * a, 9.0 * m * m / (a * a)))
    print("   With the fused kernel's 32-output-channel workgroups every input tile is transformed once per 32 output channels:")
    for m in (2, 4, 6):
        a = m + 2
        i_ops, o_ops = op_counts(m)
        # per tile, per 32 output channels and per input channel c: i_ops VALU lane-ops vs 2*a*a*32 MFMA flops = a*a*32/2 ... in 32x32x2 MFMA
        # lane-cycles: one MFMA (4096 flops per 64 lanes) = 64 lane-flops; count VALU lane-ops per MFMA lane-flop
        mfma_flops = 2 * a * a * 32
        print("   F(%d,3): %d VALU lane-ops of input transform per %d MFMA flops = %.3f per flop  (x%.1f the F(2,3) ratio)"
              % (m, i_ops, mfma_flops, i_ops / mfma_flops, (i_ops / mfma_flops) / (op_counts(2)[0] / (2 * 16 * 32))))


if __name__ == "__main__":
    main()

"""CPU emulation of wconv_kernel's index plumbing (csrc/wconv.hip): staging tasks -> swizzled LDS planes -> A-fragment reads,
the packed Winograd weight order, the MFMA 32x32x16 operand / accumulator layout and the output transform, for one 16x16 tile
(float64 values instead of bf16 hi/lo).  Checks the result against a direct 3x3 convolution and reports the ds_read_b128 bank
conflicts of the lane groups.  Development aid for changes of the kernel's layouts: python benchmarks/emulate_wconv.py"""
import sys

import numpy as np

WROW, WSTEPS = 1024, 24


def emulate(NB=4, CIN=64, seed=0, f32=False):
    """NB = blocks of 4 tile rows per workgroup: 4 (16x16-pixel tile) or 2 (8x16).  Returns (max |emulated - direct|, worst number of
    distinct addresses of one ds_read_b128 lane group on one 16-byte bank slot).
    f32 (round 6): the exact-fp32 instantiation wconv_kernel<..., F32>: the two LDS planes / the two 16-byte pieces of a packed fragment hold
    fp32 quads (channels 8 u + 4 P .. + 3 of logical unit u in plane P) and a 16-channel k-step is 8 MFMAs of depth 2 (v_mfma_f32_32x32x2_f32:
    operand lane = (row, k = lane >> 5)), MFMA (P, e) taking float e of plane P from both operands."""
    if f32:
        return _emulate_f32(NB, CIN, seed)
    TR, NTASK = 4 * NB, (5 if NB == 4 else 3)
    H, W = TR, 16
    COUT = 32
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((H, W, CIN))
    w = rng.standard_normal((COUT, CIN, 3, 3))
    y0 = x0 = 0
    nchunk = CIN // 32

    # packed weights: [nb][chunk][q][lane][8]
    U = np.zeros((COUT // 32, nchunk, WSTEPS, 64, 8))
    for nb in range(COUT // 32):
        for chunk in range(nchunk):
            for q in range(WSTEPS):
                ky, xi, ks = q >> 3, (q >> 1) & 3, q & 1
                for lane in range(64):
                    for e in range(8):
                        n = nb * 32 + (lane & 31)
                        k = chunk * 32 + ks * 16 + (lane >> 5) * 8 + e
                        g = w[n, k, ky, :]
                        U[nb, chunk, q, lane, e] = (g[0], 0.5 * (g[0] + g[1] + g[2]), 0.5 * (g[0] - g[1] + g[2]), g[2])[xi]

    acc = np.zeros((4, NB, 64, 16))  # [xi][block][lane][r]  (wave 0 = channel block 0)
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups[:2]]
    worst = 0
    for chunk in range(nchunk):
        lds = np.full((TR + 2) * WROW, np.nan)
        for tid in range(256):
            wave, c4, sp = tid >> 6, tid & 7, (tid >> 3) & 7
            for j in range(NTASK):
                row = wave + 4 * j
                if row >= TR + 2:
                    row -= 2
                yy = y0 + row - 1
                wbase = row * WROW + sp * 32 + (((c4 >> 1) + row) & 3) * 8 + (c4 & 1) * 4
                d = []
                for k in range(4):
                    xx = x0 + 2 * sp + k - 1
                    ok = 0 <= yy < H and 0 <= xx < W
                    d.append(x[yy, xx, chunk * 32 + c4 * 4: chunk * 32 + c4 * 4 + 4] if ok else np.zeros(4))
                V = (d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3])
                for xi in range(4):
                    lds[wbase + xi * 256: wbase + xi * 256 + 4] = V[xi]
        for q in range(WSTEPS):
            ky, xi, ks = q >> 3, (q >> 1) & 3, q & 1
            for b in range(NB):
                Amat = np.zeros((32, 16))  # pixel pairs x k
                Bmat = np.zeros((32, 16))  # channels x k
                addr = {}
                for lane in range(64):
                    l31, hh = lane & 31, lane >> 5
                    lr, lp = l31 >> 3, l31 & 7
                    fro = lr * WROW + lp * 32 + ((hh + lr + ((2 * ks + ky) & 3)) & 3) * 8
                    o = fro + ky * WROW + xi * 256 + b * 4 * WROW
                    addr[lane] = o * 2  # bytes
                    Amat[l31, hh * 8: hh * 8 + 8] = lds[o: o + 8]
                    Bmat[l31, hh * 8: hh * 8 + 8] = U[0, chunk, q, lane]
                for g in groups:
                    slots = [(addr[l] // 16) % 16 for l in g]
                    worst = max(worst, max(slots.count(s) for s in set(slots)))
                D = Bmat @ Amat.T  # rows = channels, columns = pixel pairs
                for lane in range(64):
                    l31, hh = lane & 31, lane >> 5
                    for r in range(16):
                        acc[xi, b, lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh, l31]
    assert not np.isnan(acc).any(), "a fragment read touched an LDS element no task wrote"

    out = np.zeros((H, W, 32))
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        lr, lp = l31 >> 3, l31 & 7
        for b in range(NB):
            for g in range(4):
                for i in range(4):
                    m = [acc[xi, b, lane, 4 * g + i] for xi in range(4)]
                    col = 8 * g + 4 * hh + i
                    out[y0 + 4 * b + lr, x0 + 2 * lp, col] = m[0] + m[1] + m[2]
                    out[y0 + 4 * b + lr, x0 + 2 * lp + 1, col] = m[1] - m[2] - m[3]

    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    ref = np.zeros((H, W, 32))
    for ky in range(3):
        for kx in range(3):
            ref += xp[ky: ky + H, kx: kx + W, :] @ w[:32, :, ky, kx].T

    return float(np.abs(out - ref).max()), worst


def _emulate_f32(NB, CIN, seed):
    TR, NTASK = 4 * NB, (5 if NB == 4 else 3)
    H, W, COUT = TR, 16, 32
    WPLANE = (TR + 2) * WROW  # bf16 elements per plane; an fp32 value occupies two of them
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((H, W, CIN))
    w = rng.standard_normal((COUT, CIN, 3, 3))
    nchunk = CIN // 32
    # pack_wino_kernel<true>: [nb][chunk][q][plane][lane][4 floats]; element e of the lane's 8 channels -> plane e >> 2, float e & 3
    U = np.zeros((nchunk, WSTEPS, 2, 64, 4))
    for chunk in range(nchunk):
        for q in range(WSTEPS):
            ky, xi, ks = q >> 3, (q >> 1) & 3, q & 1
            for lane in range(64):
                for e in range(8):
                    n, k = lane & 31, chunk * 32 + ks * 16 + (lane >> 5) * 8 + e
                    g = w[n, k, ky, :]
                    U[chunk, q, e >> 2, lane, e & 3] = (g[0], 0.5 * (g[0] + g[1] + g[2]), 0.5 * (g[0] - g[1] + g[2]), g[2])[xi]
    acc = np.zeros((4, NB, 64, 16))
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups[:2]]
    worst = 0
    for chunk in range(nchunk):
        lds = np.full(2 * WPLANE // 2, np.nan)  # floats; float index = bf16 element index / 2
        for tid in range(256):
            wave, c4, sp = tid >> 6, tid & 7, (tid >> 3) & 7
            for j in range(NTASK):
                row = wave + 4 * j
                if row >= TR + 2:
                    row -= 2
                yy = row - 1
                wbase = (c4 & 1) * WPLANE + row * WROW + sp * 32 + (((c4 >> 1) + row) & 3) * 8
                d = []
                for k in range(4):
                    xx = 2 * sp + k - 1
                    ok = 0 <= yy < H and 0 <= xx < W
                    d.append(x[yy, xx, chunk * 32 + c4 * 4: chunk * 32 + c4 * 4 + 4] if ok else np.zeros(4))
                V = (d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3])
                for xi in range(4):
                    o = (wbase + xi * 256) // 2
                    lds[o: o + 4] = V[xi]
        for q in range(WSTEPS):
            ky, xi, ks = q >> 3, (q >> 1) & 3, q & 1
            for b in range(NB):
                frag = np.zeros((2, 64, 4))
                for P in range(2):
                    addr = {}
                    for lane in range(64):
                        l31, hh = lane & 31, lane >> 5
                        lr, lp = l31 >> 3, l31 & 7
                        fro = lr * WROW + lp * 32 + ((hh + lr + ((2 * ks + ky) & 3)) & 3) * 8
                        o = P * WPLANE + fro + ky * WROW + xi * 256 + b * 4 * WROW
                        addr[lane] = o * 2
                        frag[P, lane] = lds[o // 2: o // 2 + 4]
                    for g in groups:
                        slots = [(addr[l] // 16) % 16 for l in g]
                        worst = max(worst, max(slots.count(t) for t in set(slots)))
                D = np.zeros((32, 32))
                for P in range(2):
                    for e in range(4):
                        Am = np.zeros((32, 2))  # operand B of the MFMA: pixel pairs x k (k = lane >> 5)
                        Bm = np.zeros((32, 2))  # operand A: channels x k
                        for lane in range(64):
                            Am[lane & 31, lane >> 5] = frag[P, lane, e]
                            Bm[lane & 31, lane >> 5] = U[chunk, q, P, lane, e]
                        D += Bm @ Am.T
                for lane in range(64):
                    l31, hh = lane & 31, lane >> 5
                    for r in range(16):
                        acc[xi, b, lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh, l31]
    assert not np.isnan(acc).any(), "a fragment read touched an LDS element no task wrote"
    out = np.zeros((H, W, 32))
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        lr, lp = l31 >> 3, l31 & 7
        for b in range(NB):
            for g in range(4):
                for i in range(4):
                    m = [acc[xi, b, lane, 4 * g + i] for xi in range(4)]
                    col = 8 * g + 4 * hh + i
                    out[4 * b + lr, 2 * lp, col] = m[0] + m[1] + m[2]
                    out[4 * b + lr, 2 * lp + 1, col] = m[1] - m[2] - m[3]
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    ref = np.zeros((H, W, 32))
    for ky in range(3):
        for kx in range(3):
            ref += xp[ky: ky + H, kx: kx + W, :] @ w[:32, :, ky, kx].T
    return float(np.abs(out - ref).max()), worst


if __name__ == "__main__":
    for nb in ((int(sys.argv[1]),) if len(sys.argv) > 1 else (4, 2)):
        err, worst = emulate(nb)
        print(f"NB={nb}: max |emulated - direct| = {err:.3e}, worst ds_read_b128 lane-group multiplicity = {worst}")
        assert err < 1e-9 and worst == 1
        err, worst = emulate(nb, f32=True)
        print(f"NB={nb} (fp32 products): max |emulated - direct| = {err:.3e}, worst ds_read_b128 lane-group multiplicity = {worst}")
        assert err < 1e-9 and worst == 1

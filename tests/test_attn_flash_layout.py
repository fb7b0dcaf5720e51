"""CPU emulation of the lane-level data movement of csrc/attn_flash.hip (round 5): staging maps (natural / transposed-with-permutation
LDS layouts), MFMA fragment and accumulator layouts, the accumulator-registers-as-next-operand trick, the per-wavefront key / query
partitions and the LDS merges — everything except the bf16 hi/lo split — replayed in float64 with numpy and compared with plain softmax
attention and its gradients.  The MFMA layout itself (lane (l31, hh) holds X[l31][16 s + 8 hh + e]; D[(r & 3) + 8 (r >> 2) + 4 hh][l31] in
register r) is the one the GPU-validated kernels of hgemm.hip / attn.hip rely on.  A wrong index anywhere in the kernels' maps shows up
here without a GPU."""
import numpy as np
import pytest

NP_, TP_, OP_ = 72, 40, 68


def fa_row(lane, i):
    qg = lane >> 4
    return 16 * (qg >> 1) + 4 * (qg & 1) + (i & 3) + 8 * (i >> 2)


def gload(src, nvalid):
    """src: [rows][64] view starting at the block's first row -> rg[lane][8][4]"""
    rg = np.zeros((64, 8, 4))
    for lane in range(64):
        dq = lane & 15
        for i in range(8):
            r = fa_row(lane, i)
            if r < nvalid:
                rg[lane, i] = src[r, 4 * dq:4 * dq + 4]
    return rg


def store_nat(rg, scale):
    pl = np.full(32 * NP_, np.nan)
    for lane in range(64):
        dq = lane & 15
        for i in range(8):
            off = fa_row(lane, i) * NP_ + 4 * dq
            pl[off:off + 4] = rg[lane, i] * scale
    return pl


def store_tr(rg, scale):
    pl = np.full(64 * TP_, np.nan)
    for lane in range(64):
        dq, qg = lane & 15, lane >> 4
        tpos = 16 * (qg >> 1) + 8 * (qg & 1)
        for c in range(4):
            off = (4 * dq + c) * TP_ + tpos
            pl[off:off + 8] = rg[lane, :, c] * scale
    return pl


def frag_nat(pl, s):
    f = np.zeros((64, 8))
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        o = l31 * NP_ + 16 * s + 8 * hh
        f[lane] = pl[o:o + 8]
    return f


def frag_tr(pl, t, j):
    f = np.zeros((64, 8))
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        o = (32 * t + l31) * TP_ + 16 * j + 8 * hh
        f[lane] = pl[o:o + 8]
    return f


def row_frags(mat, row0, T, scale):
    """fa_row_frags: lane (l31, hh) holds mat[row0 + l31][16 s + 8 hh + e] -> [4][64][8]"""
    f = np.zeros((4, 64, 8))
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        if row0 + l31 < T:
            for s in range(4):
                f[s, lane] = mat[row0 + l31, 16 * s + 8 * hh:16 * s + 8 * hh + 8] * scale
    return f


def mma(acc, xf, yf):
    """acc[lane][16] += D in the 32x32 C/D layout, D[i][j] = sum over (hh, e) X-frag[(i, hh)][e] * Y-frag[(j, hh)][e]"""
    X = np.zeros((32, 16))
    Y = np.zeros((32, 16))
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        X[l31, 8 * hh:8 * hh + 8] = xf[lane]
        Y[l31, 8 * hh:8 * hh + 8] = yf[lane]
    D = X @ Y.T
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        for r in range(16):
            acc[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hh, l31]


def split_acc(p):
    """registers 8 j .. 8 j + 7 -> Y operand of k-step j: [2][64][8]"""
    return np.stack([p[:, 0:8], p[:, 8:16]])


def park(o2):
    """o2[t][lane][16] -> slab[x][d] (pitch OP_)"""
    slab = np.full(32 * OP_, np.nan)
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        for t in range(2):
            for g in range(4):
                off = l31 * OP_ + 32 * t + 8 * g + 4 * hh
                slab[off:off + 4] = o2[t][lane, 4 * g:4 * g + 4]
    return slab.reshape(32, OP_)[:, :64]


def reg_row(r, hh):
    return (r & 3) + 8 * (r >> 2) + 4 * hh


def emu_fwd(q, k, v, T, qb, alpha):
    q0 = qb * 32
    nkb = (T + 31) >> 5
    qf = row_frags(q, q0, T, alpha)
    slabs, ms, ls = [], [], []
    for w in range(4):
        o = [np.zeros((64, 16)), np.zeros((64, 16))]
        m_run = np.full(64, -np.inf)
        l_run = np.zeros(64)
        for b in range(w, nkb, 4):
            Kn = store_nat(gload(k[b * 32:], T - b * 32), 1.0)
            Vt = store_tr(gload(v[b * 32:], T - b * 32), 1.0)
            sacc = np.zeros((64, 16))
            for s in range(4):
                mma(sacc, frag_nat(Kn, s), qf[s])
            p = np.zeros((64, 16))
            for lane in range(64):
                hh = lane >> 5
                for r in range(16):
                    key = b * 32 + reg_row(r, hh)
                    p[lane, r] = sacc[lane, r] if key < T else -np.inf
            bm = p.max(axis=1)
            bm = np.maximum(bm, bm[np.arange(64) ^ 32])
            mn = np.maximum(m_run, bm)
            corr = np.exp(m_run - mn)
            p = np.exp(p - mn[:, None])
            l_run = l_run * corr + p.sum(axis=1)
            m_run = mn
            for t in range(2):
                o[t] *= corr[:, None]
            pf = split_acc(p)
            for t in range(2):
                for j in range(2):
                    mma(o[t], frag_tr(Vt, t, j), pf[j])
        l_run = l_run + l_run[np.arange(64) ^ 32]
        slabs.append(park(o))
        ms.append(m_run[:32].copy())
        ls.append(l_run[:32].copy())
    M = np.max(np.stack(ms), axis=0)
    out = np.zeros((32, 64))
    L = np.zeros(32)
    for w in range(4):
        e = np.exp(ms[w] - M)
        L += e * ls[w]
        out += slabs[w] * e[:, None]
    return out / L[:, None], M + np.log(L)


def emu_dq(q, k, v, do, lse_pad, D_pad, T, qb, alpha):
    q0 = qb * 32
    nkb = (T + 31) >> 5
    qf = row_frags(q, q0, T, alpha)
    gf = row_frags(do, q0, T, 1.0)
    lq = np.array([lse_pad[q0 + (lane & 31)] for lane in range(64)])
    Dq = np.array([D_pad[q0 + (lane & 31)] for lane in range(64)])
    total = np.zeros((32, 64))
    for w in range(4):
        dq = [np.zeros((64, 16)), np.zeros((64, 16))]
        for b in range(w, nkb, 4):
            kr = gload(k[b * 32:], T - b * 32)
            Kn, Kt = store_nat(kr, 1.0), store_tr(kr, 1.0)
            Vn = store_nat(gload(v[b * 32:], T - b * 32), 1.0)
            sacc, dp = np.zeros((64, 16)), np.zeros((64, 16))
            for s in range(4):
                mma(sacc, frag_nat(Kn, s), qf[s])
                mma(dp, frag_nat(Vn, s), gf[s])
            ds = np.zeros((64, 16))
            for lane in range(64):
                hh = lane >> 5
                for r in range(16):
                    key = b * 32 + reg_row(r, hh)
                    p = np.exp(sacc[lane, r] - lq[lane]) if key < T else 0.0
                    ds[lane, r] = p * (dp[lane, r] - Dq[lane])
            df = split_acc(ds)
            for t in range(2):
                for j in range(2):
                    mma(dq[t], frag_tr(Kt, t, j), df[j])
        total += park(dq)
    return total * alpha


def emu_dkv(q, k, v, do, lse_pad, D_pad, T, kb, alpha):
    k0 = kb * 32
    nqb = (T + 31) >> 5
    kf = row_frags(k, k0, T, 1.0)
    vf = row_frags(v, k0, T, 1.0)
    dv_tot, dk_tot = np.zeros((32, 64)), np.zeros((32, 64))
    for w in range(4):
        dv = [np.zeros((64, 16)), np.zeros((64, 16))]
        dk = [np.zeros((64, 16)), np.zeros((64, 16))]
        for b in range(w, nqb, 4):
            qr = gload(q[b * 32:], T - b * 32)
            gr = gload(do[b * 32:], T - b * 32)
            Qn, Qt = store_nat(qr, alpha), store_tr(qr, alpha)
            Gn, Gt = store_nat(gr, 1.0), store_tr(gr, 1.0)
            sacc, dp = np.zeros((64, 16)), np.zeros((64, 16))
            for s in range(4):
                mma(sacc, frag_nat(Qn, s), kf[s])
                mma(dp, frag_nat(Gn, s), vf[s])
            p, ds = np.zeros((64, 16)), np.zeros((64, 16))
            for lane in range(64):
                hh = lane >> 5
                for r in range(16):
                    g = r >> 2
                    row = b * 32 + 8 * g + 4 * hh + (r & 3)  # lr[g][r & 3] of the kernel
                    p[lane, r] = np.exp(sacc[lane, r] - lse_pad[row])
                    ds[lane, r] = p[lane, r] * (dp[lane, r] - D_pad[row])
            pf, df = split_acc(p), split_acc(ds)
            for t in range(2):
                for j in range(2):
                    mma(dv[t], frag_tr(Gt, t, j), pf[j])
                    mma(dk[t], frag_tr(Qt, t, j), df[j])
        dv_tot += park(dv)
        dk_tot += park(dk)
    return dk_tot, dv_tot


def reference(q, k, v, do, alpha):
    s = alpha * q @ k.T
    p = np.exp(s - s.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    o = p @ v
    dp = do @ v.T
    D = (do * o).sum(axis=1)
    ds = p * (dp - D[:, None])
    return o, alpha * ds @ k, alpha * ds.T @ q, p.T @ do, D


@pytest.mark.parametrize("T", [96, 100, 160])
def test_flash_attention_lane_maps_reproduce_softmax_attention_and_its_gradients(T):
    rng = np.random.default_rng(T)
    q, k, v, do = (rng.standard_normal((T, 64)) for _ in range(4))
    alpha = 0.125
    o_ref, dq_ref, dk_ref, dv_ref, D_ref = reference(q, k, v, do, alpha)
    nb = (T + 31) >> 5
    Tq = nb * 32
    lse_pad = np.full(Tq, np.inf)
    D_pad = np.zeros(Tq)
    D_pad[:T] = D_ref
    # padded copies so that block views never run out of rows (the kernels clamp instead)
    pad = lambda a: np.concatenate([a, np.full((Tq + 32 - T, 64), 1e30)])  # noqa: E731  (poison: must never be used)
    qp, kp, vp, dop = pad(q), pad(k), pad(v), pad(do)
    for qb in range(nb):
        o, lse = emu_fwd(qp, kp, vp, T, qb, alpha)
        n = min(32, T - qb * 32)
        np.testing.assert_allclose(o[:n], o_ref[qb * 32:qb * 32 + n], rtol=1e-10, atol=1e-12)
        lse_pad[qb * 32:qb * 32 + n] = lse[:n]
    s = alpha * q @ k.T
    np.testing.assert_allclose(lse_pad[:T], np.log(np.exp(s - s.max(1, keepdims=True)).sum(1)) + s.max(1), rtol=1e-12)
    for qb in range(nb):
        n = min(32, T - qb * 32)
        dq = emu_dq(qp, kp, vp, dop, lse_pad, D_pad, T, qb, alpha)
        np.testing.assert_allclose(dq[:n], dq_ref[qb * 32:qb * 32 + n], rtol=1e-9, atol=1e-11)
    for kb in range(nb):
        n = min(32, T - kb * 32)
        dk, dv = emu_dkv(qp, kp, vp, dop, lse_pad, D_pad, T, kb, alpha)
        np.testing.assert_allclose(dk[:n], dk_ref[kb * 32:kb * 32 + n], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(dv[:n], dv_ref[kb * 32:kb * 32 + n], rtol=1e-9, atol=1e-11)


def emu_bwd_small(q, k, v, do, lse_pad, D_pad, T, alpha):
    """attn_flash_bwd_small_kernel: wavefront (qi, kj) computes both orientations of its score tile; dQ partials over kj, dK / dV over qi."""
    dq_tot, dk_tot, dv_tot = np.zeros((64, 64)), np.zeros((64, 64)), np.zeros((64, 64))
    img = {}
    for blk in range(2):
        nv = T - blk * 32
        z = np.zeros((64, 8, 4))
        qr = gload(q[blk * 32:], nv) if nv > 0 else z
        kr = gload(k[blk * 32:], nv) if nv > 0 else z
        vr = gload(v[blk * 32:], nv) if nv > 0 else z
        gr = gload(do[blk * 32:], nv) if nv > 0 else z
        img[blk] = dict(Qn=store_nat(qr, alpha), Qt=store_tr(qr, alpha), Kn=store_nat(kr, 1.0), Kt=store_tr(kr, 1.0), Vn=store_nat(vr, 1.0),
                        Gn=store_nat(gr, 1.0), Gt=store_tr(gr, 1.0))
    for w in range(4):
        qi, kj = w >> 1, w & 1
        A, Bk = img[qi], img[kj]
        st, dpt = np.zeros((64, 16)), np.zeros((64, 16))
        for s_ in range(4):
            mma(st, frag_nat(Bk["Kn"], s_), frag_nat(A["Qn"], s_))
            mma(dpt, frag_nat(Bk["Vn"], s_), frag_nat(A["Gn"], s_))
        ds = np.zeros((64, 16))
        for lane in range(64):
            l31, hh = lane & 31, lane >> 5
            for r in range(16):
                key = kj * 32 + reg_row(r, hh)
                p = np.exp(st[lane, r] - lse_pad[qi * 32 + l31]) if key < T else 0.0
                ds[lane, r] = p * (dpt[lane, r] - D_pad[qi * 32 + l31])
        dq = [np.zeros((64, 16)), np.zeros((64, 16))]
        df = split_acc(ds)
        for t in range(2):
            for j in range(2):
                mma(dq[t], frag_tr(Bk["Kt"], t, j), df[j])
        sa, dp = np.zeros((64, 16)), np.zeros((64, 16))
        for s_ in range(4):
            mma(sa, frag_nat(A["Qn"], s_), frag_nat(Bk["Kn"], s_))
            mma(dp, frag_nat(A["Gn"], s_), frag_nat(Bk["Vn"], s_))
        p2, ds2 = np.zeros((64, 16)), np.zeros((64, 16))
        for lane in range(64):
            hh = lane >> 5
            for r in range(16):
                row = qi * 32 + reg_row(r, hh)
                p2[lane, r] = np.exp(sa[lane, r] - lse_pad[row])
                ds2[lane, r] = p2[lane, r] * (dp[lane, r] - D_pad[row])
        dv = [np.zeros((64, 16)), np.zeros((64, 16))]
        dk = [np.zeros((64, 16)), np.zeros((64, 16))]
        pf, df2 = split_acc(p2), split_acc(ds2)
        for t in range(2):
            for j in range(2):
                mma(dv[t], frag_tr(A["Gt"], t, j), pf[j])
                mma(dk[t], frag_tr(A["Qt"], t, j), df2[j])
        dq_tot[qi * 32:qi * 32 + 32] += park(dq)
        dv_tot[kj * 32:kj * 32 + 32] += park(dv)
        dk_tot[kj * 32:kj * 32 + 32] += park(dk)
    return dq_tot * alpha, dk_tot, dv_tot


@pytest.mark.parametrize("T", [20, 33, 50, 64])
def test_fused_small_T_backward_lane_maps(T):
    rng = np.random.default_rng(100 + T)
    q, k, v, do = (rng.standard_normal((T, 64)) for _ in range(4))
    alpha = 0.125
    o_ref, dq_ref, dk_ref, dv_ref, D_ref = reference(q, k, v, do, alpha)
    s = alpha * q @ k.T
    lse_pad = np.full(64, np.inf)
    lse_pad[:T] = np.log(np.exp(s - s.max(1, keepdims=True)).sum(1)) + s.max(1)
    D_pad = np.zeros(64)
    D_pad[:T] = D_ref
    pad = lambda a: np.concatenate([a, np.full((96 - T, 64), 1e30)])  # noqa: E731  (poison: must never be used)
    dq, dk, dv = emu_bwd_small(pad(q), pad(k), pad(v), pad(do), lse_pad, D_pad, T, alpha)
    np.testing.assert_allclose(dq[:T], dq_ref, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dk[:T], dk_ref, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dv[:T], dv_ref, rtol=1e-9, atol=1e-11)

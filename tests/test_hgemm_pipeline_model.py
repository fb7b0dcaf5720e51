"""CPU model of hgemm2_kernel's software pipeline (csrc/hgemm.hip) for any (RING, NSET): replays the order in which the prologue, the unrolled
loop body (U chunks) and the tail issue weight-fragment loads, patch loads, patch conversions and MFMAs, and checks that every k-step multiplies
the fragments of ITS k index, that every chunk is converted from the register set that holds it into the LDS buffer the MFMAs then read, and
that no ring slot / register set is overwritten before its consumer ran.  No GPU, no library: the index arithmetic is restated here."""
import math

import pytest


def lcm(a, b):
    return a // math.gcd(a, b) * b


def replay(ring, nset, nchunks):
    dist, ahead = ring - 1, nset + 1
    U = lcm(lcm(2, nset), ring // 4)
    kq_last = nchunks * 4 - 1
    bq = [None] * ring          # ring slot -> k-step index it holds
    sets = [None] * nset        # staging register set -> chunk it holds
    lds = [None, None]          # LDS buffer -> chunk it holds (converted patch)
    mfma = []                   # (chunk, k-step within chunk)

    def b_load(slot, kq):
        bq[slot] = min(kq, kq_last)

    def patch_load(k, ch):
        sets[k] = min(ch, nchunks - 1)

    # prologue
    patch_load(0, 0)
    for q in range(dist):
        b_load(q, q)
    lds[0] = sets[0]            # H2_PATCH_STORE(prs[0], buf0)
    for k in range(nset):
        patch_load(k, 1 + k)

    def chunk(S, cur, nxt, C, k):
        kq = C * 4
        for q in range(4):
            b_load((S + q + dist) % ring, kq + q + dist)
            assert lds[cur] == C, (ring, nset, nchunks, C, "MFMAs read the wrong LDS buffer")
            assert bq[(S + q) % ring] == kq + q, (ring, nset, nchunks, C, q, "ring slot does not hold this k-step")
            mfma.append((C, q))
            if q == 2:          # conversion of the next chunk is complete before the barrier in k-step 3
                if C + 1 < nchunks:
                    assert sets[k] == C + 1, (ring, nset, nchunks, C, "staging set does not hold the next chunk")
                lds[nxt] = sets[k]
                patch_load(k, C + ahead)

    c = 0
    while c + U - 1 < nchunks:
        for u in range(U):
            chunk((4 * u) % ring, u & 1, (u & 1) ^ 1, c + u, u % nset)
        c += U
    for u in range(U - 1):
        if c + u < nchunks:
            chunk((4 * u) % ring, u & 1, (u & 1) ^ 1, c + u, u % nset)
    assert mfma == [(C, q) for C in range(nchunks) for q in range(4)]


@pytest.mark.parametrize("ring,nset", [(8, 1), (8, 2), (8, 3), (8, 4), (12, 2), (12, 3), (12, 4), (16, 2), (16, 3), (16, 4), (20, 4)])
def test_every_kstep_uses_its_own_fragments_for_any_depth(ring, nset):
    for nchunks in range(1, 64):
        replay(ring, nset, nchunks)

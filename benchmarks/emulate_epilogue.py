"""CPU emulation of the LDS-transposed epilogues of wconv_kernel (csrc/wconv.hip) and hgemm2_kernel (csrc/hgemm.hip): a wavefront parks its
output block, held in the MFMA 32x32 C/D layout (lane = column, 4-register quads = rows), in a private LDS slab as [pixel or row][32 channels]
fp32 with the 16-byte units of a line XOR-swizzled, and reads it back with 8 consecutive lanes on one line so that every global access moves
whole 128-byte lines.  Replays both index computations, checks that the read-back finds exactly the element the write put there, and counts
bank conflicts per hardware lane group (ds_write_b128: 8 groups of 8 consecutive lanes over 32 banks; ds_read_b128: 4 groups of 16 lanes over
the 16 sixteen-byte slots of the 256-byte bank row — MI355X_MICROARCH LDS table).  Development aid: python benchmarks/emulate_epilogue.py"""
import numpy as np

_G0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
_G1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
READ_GROUPS = [_G0, _G1, [l + 32 for l in _G0], [l + 32 for l in _G1]]
WRITE_GROUPS = [list(range(8 * k, 8 * k + 8)) for k in range(8)]


def _worst_write(addr):
    worst = 0
    for grp in WRITE_GROUPS:
        banks = {}
        for l in grp:
            for e in range(4):
                banks.setdefault((addr[l] + e) % 32, set()).add(addr[l] + e)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def _worst_read(addr):
    worst = 0
    for grp in READ_GROUPS:
        slots = {}
        for l in grp:
            slots.setdefault((addr[l] // 4) % 16, set()).add(addr[l])
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def wconv(NB):
    """wconv_kernel<GN, NB>: tile of 4 NB rows x 16 pixels; lane (l31 = pixel pair (row lr of block b, pair lp), hh) holds channel quads
    8 g + 4 hh of the pair's even and odd pixel.  Returns (worst write conflict, worst read conflict); 1 = conflict-free."""
    TR = 4 * NB
    slab = np.full(TR * 16 * 32, -1.0)
    ww = wr = 0
    for b in range(NB):
        for g in range(4):
            for odd in range(2):
                addr = {}
                for lane in range(64):
                    l31, hh = lane & 31, lane >> 5
                    lr, lp = l31 >> 3, l31 & 7
                    pix = (4 * b + lr) * 16 + 2 * lp + odd
                    a = pix * 32 + ((2 * g + hh) ^ lp) * 4
                    addr[lane] = a
                    for e in range(4):
                        assert slab[a + e] == -1
                        slab[a + e] = pix * 1000 + 8 * g + 4 * hh + e
                ww = max(ww, _worst_write(addr))
    assert (slab >= 0).all()
    for i in range(2 * TR):  # instruction i: tile row i >> 1, columns 8 (i & 1) + psub
        addr = {}
        for lane in range(64):
            psub, quad = lane >> 3, lane & 7
            base = psub * 32 + ((quad ^ (psub >> 1) ^ (4 if i & 1 else 0)) & 7) * 4  # sl0 / sl1 of the kernel
            a = base + i * 256
            addr[lane] = a
            pix = (i >> 1) * 16 + 8 * (i & 1) + psub
            for e in range(4):
                assert slab[a + e] == pix * 1000 + 4 * quad + e, (NB, i, lane, e)
        wr = max(wr, _worst_read(addr))
    return ww, wr


def hgemm2(TM):
    """hgemm2_kernel<1, TM>: a wavefront owns TM rows x 32 columns; lane (l31 = row of block i, hh) holds column quads 8 g + 4 hh."""
    slab = np.full(TM * 32, -1.0)
    ww = wr = 0
    for i in range(TM // 32):
        for g in range(4):
            addr = {}
            for lane in range(64):
                l31, hh = lane & 31, lane >> 5
                rl = i * 32 + l31
                a = rl * 32 + (((2 * g + hh) ^ rl) & 7) * 4
                addr[lane] = a
                for e in range(4):
                    assert slab[a + e] == -1
                    slab[a + e] = rl * 1000 + 8 * g + 4 * hh + e
            ww = max(ww, _worst_write(addr))
    assert (slab >= 0).all()
    for it in range(TM // 8):
        addr = {}
        for lane in range(64):
            rsub, quad = lane >> 3, lane & 7
            a = rsub * 32 + ((quad ^ rsub) & 7) * 4 + it * 256
            addr[lane] = a
            for e in range(4):
                assert slab[a + e] == (8 * it + rsub) * 1000 + 4 * quad + e, (TM, it, lane, e)
        wr = max(wr, _worst_read(addr))
    return ww, wr


if __name__ == "__main__":
    for nb in (4, 2):
        print("wconv_kernel NB", nb, "worst write / read conflict", wconv(nb))
    for tm in (64, 128):
        print("hgemm2_kernel TM", tm, "worst write / read conflict", hgemm2(tm))

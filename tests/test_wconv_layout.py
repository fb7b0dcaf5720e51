"""CPU tests of the Winograd halo conv kernel's host-visible logic (csrc/wconv.hip): the layouts it relies on (emulated end to end on
the CPU), its staging schedule, and the numerics of the 1-D F(2,3) transform on split-bf16 products.  No GPU."""
import ctypes as C
import math
import os
import sys

import torch as th
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib  # noqa: E402


def test_layouts_emulated_end_to_end_match_a_direct_convolution_and_are_bank_conflict_free():
    """staging tasks -> swizzled V planes -> fragment reads -> MFMA operand / accumulator layout -> output transform, for both tile
    heights: the emulation reproduces a direct 3x3 convolution, every fragment read finds an element some task wrote, and no
    ds_read_b128 lane group puts two different addresses on one 16-byte slot of the bank row."""
    from benchmarks import emulate_wconv
    for nb in (4, 2):
        err, worst = emulate_wconv.emulate(nb, CIN=32)
        assert err < 1e-9 and worst == 1, (nb, err, worst)
        # round 6: the exact-fp32 instantiation (fp32 quads in the two planes, 8 MFMAs of depth 2 per k-step, pack_wino_kernel<true>)
        err, worst = emulate_wconv.emulate(nb, CIN=32, f32=True)
        assert err < 1e-9 and worst == 1, ("f32", nb, err, worst)


def test_lds_transposed_epilogues_round_trip_and_are_bank_conflict_free():
    """wconv_kernel and hgemm2_kernel park their output block in an XOR-swizzled LDS slab and read it back line-wise (whole-line global
    stores): the read-back finds exactly the element each lane wrote, and neither the ds_write_b128 nor the ds_read_b128 lane groups put
    two different addresses on one bank / 16-byte slot."""
    from benchmarks import emulate_epilogue
    for nb in (4, 2):
        assert emulate_epilogue.wconv(nb) == (1, 1)
    for tm in (64, 128):
        assert emulate_epilogue.hgemm2(tm) == (1, 1)


def test_staging_schedule_never_reloads_a_live_register_slot():
    """Task k of the next chunk is loaded into slot k & 1 and transformed in six pieces later in the same chunk (one or two per step): every
    task is loaded and transformed exactly once, in order, with >= 5 steps of load latency covered, and a slot is only reloaded after the
    last piece of its previous task (the load is issued at the top of a step, the pieces after the step's MFMAs)."""
    handle = lib.load()
    for nb, ntask in ((4, 5), (2, 3)):
        load, pieces = {}, {}
        for q in range(24):
            out = (C.c_int * 7)()
            assert handle.cgd_op_wconv_schedule(nb, q, out) == 0
            if out[0] >= 0:
                assert out[0] not in load
                load[out[0]] = q
            for piece in range(6):
                if out[1 + piece] >= 0:
                    assert (out[1 + piece], piece) not in pieces
                    pieces[(out[1 + piece], piece)] = q
        assert sorted(load) == list(range(ntask)) and sorted(pieces) == [(k, p) for k in range(ntask) for p in range(6)]
        for k in range(ntask):
            steps = [pieces[(k, p)] for p in range(6)]
            assert load[k] + 5 <= steps[0] and steps == sorted(steps) and steps[-1] <= 23
            if k >= 2:  # slot k & 1 held task k - 2
                assert load[k] > pieces[(k - 2, 5)]
        assert handle.cgd_op_wconv_schedule(3, 0, (C.c_int * 7)()) == -3 and handle.cgd_op_wconv_schedule(4, 24, (C.c_int * 7)()) == -3


def _split(x):
    hi = x.float().bfloat16().float()
    return hi.double(), (x.float() - hi).bfloat16().float().double()


def test_winograd_f23_on_split_bf16_products_keeps_the_fp32_grade_error():
    """1-D F(2,3) along W with bf16x3 products (V and U rounded to fp32, then hi/lo split, al*bh + ah*bl + ah*bh) against float64: its
    error stays within 1.6x of the direct bf16x3 convolution and far inside the 1e-4 absolute tolerance at unit scale."""
    g = th.Generator().manual_seed(3)
    Cc, K, H = 128, 64, 16
    x = th.randn(1, Cc, H, H, generator=g)
    w = th.randn(K, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xh, xl = _split(x)
    wh, wl = _split(w)
    direct = F.conv2d(xh, wh, padding=1) + F.conv2d(xl, wh, padding=1) + F.conv2d(xh, wl, padding=1)
    Bt = th.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=th.float64)
    G = th.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=th.float64)
    At = th.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=th.float64)
    t = F.pad(x.double(), (1, 1, 1, 1)).unfold(3, 4, 2)                       # N C H+2 pairs 4
    V = th.einsum("ij,nchpj->nchpi", Bt, t).float()
    U = th.einsum("ij,kcyj->kcyi", G, w.double()).float()
    Vh, Vl = _split(V)
    Uh, Ul = _split(U)

    def prod(a, b):
        return th.einsum("nchpiy,kcyi->nkhpi", a.unfold(2, 3, 1), b)

    M = prod(Vh, Uh) + prod(Vl, Uh) + prod(Vh, Ul)
    wino = th.einsum("ji,nkhpi->nkhpj", At, M).reshape(1, K, H, H)
    e_d = (direct - ref).pow(2).mean().sqrt().item()
    e_w = (wino - ref).pow(2).mean().sqrt().item()
    assert ref.pow(2).mean().sqrt().item() > 0.5
    assert e_w < 1.6 * e_d and (wino - ref).abs().max().item() < 5e-5

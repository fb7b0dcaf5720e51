"""Oracle pinned against golden vectors produced by the REAL reference modules (tests/golden/make_golden.py):
cgd/losses.py, cgd/modules.py (MakeCutouts incl. RNG draw order and the H/W naming quirk), script_util naming."""
import json
import os

import numpy as np
import torch as th

from oracle import guidance as og

G = os.path.join(os.path.dirname(__file__), "golden")
OPS = np.load(os.path.join(G, "reference_ops.npz"))
HOST = json.load(open(os.path.join(G, "reference_host.json")))


def t(name):
    return th.from_numpy(OPS[name])


def test_losses_bit_exact():
    v = t("loss_in")
    assert th.equal(og.range_loss(v), t("range_loss"))
    assert th.equal(og.tv_loss(v), t("tv_loss"))
    assert th.equal(og.spherical_dist_loss(t("sph_x"), t("sph_y")), t("spherical"))


def test_spherical_dist_formula():
    # the reference's only numeric pin on the hot path (test.py:121-128)
    x, y = th.rand(1, 3), th.rand(1, 3)
    xn, yn = th.nn.functional.normalize(x, dim=-1), th.nn.functional.normalize(y, dim=-1)
    assert th.equal(og.spherical_dist_loss(x, y), (xn - yn).norm(dim=-1).div(2).arcsin().pow(2).mul(2))


def test_make_cutouts_matches_reference_draws_and_values():
    for i, c in enumerate(HOST["cutout_cases"]):
        img = t(f"cut{i}_in")
        mk = og.MakeCutouts(c["cut"], c["cutn"], c["pow"])
        th.manual_seed(c["seed"])
        out = mk(img)
        assert [list(x) for x in mk.last_coords] == c["coords"], "coordinate draw order / arithmetic differs"
        assert th.equal(out, t(f"cut{i}_out"))
        th.manual_seed(c["seed"])
        mk.cache_coordinates(c["W"], c["H"])
        assert [list(x) for x in mk.cached_coords] == c["cached_wh"]
        assert out.shape == (c["cutn"] * c["B"], 3, c["cut"], c["cut"])  # test.py:246-249


def test_product_coords_match_reference_draws():
    import cgd_amd  # noqa: F401
    from cgd_amd import guidance as dg
    for c in HOST["cutout_cases"]:
        th.manual_seed(c["seed"])
        coords = dg.generate_coords(c["H"], c["W"], c["cutn"], c["cut"], c["pow"])
        assert [list(x) for x in coords] == c["coords"]

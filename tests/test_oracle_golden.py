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


# ---- the reference's own generator + cond_fn closure (tests/golden/make_golden_condfn.py) ------------------------------------------
CONDFN = np.load(os.path.join(G, "reference_condfn.npz"))
CONDFN_META = json.load(open(os.path.join(G, "reference_condfn.json")))


def _loss_line(log):
    """The reference's progress line (cgd.py:234-236): every log key containing 'loss', three decimals, tab-separated."""
    return "\t".join(f"{k}: {v:.3f}" for k, v in log.items() if "loss" in k.lower())


def test_oracle_cond_fn_reproduces_the_reference_generator():
    """Trajectories recorded while the REAL reference `clip_guided_diffusion` generator (its weight handling, MakeCutouts, the cond_fn
    closure with the current_timestep bookkeeping, reduce_clip / progressive_cutout / cached_cutouts, magnitude clamp, saturation term,
    B == P broadcast, H/W argument-order quirk, skip_timesteps offset) drove the oracle networks, against the oracle's restated cond_fn
    on the same networks and the same global-RNG draws.  Same machine: bit-exact; elsewhere fp32 summation order may differ."""
    from tests import condfn_replay as cr
    assert set(CONDFN_META) == set(cr.CASES)
    for name, meta in CONDFN_META.items():
        got = cr.replay_with_oracle(name)
        assert len(got) == meta["steps"]
        ref_s, ref_x0 = th.from_numpy(CONDFN[f"{name}/sample"]), th.from_numpy(CONDFN[f"{name}/pred_xstart"])
        for k, (s, x0, _) in enumerate(got):
            scale = max(1.0, ref_s[k].abs().max().item())
            assert (s - ref_s[k]).abs().max().item() <= 2e-5 * scale, (name, k)
            assert (x0 - ref_x0[k]).abs().max().item() <= 2e-5 * max(1.0, ref_x0[k].abs().max().item()), (name, k)
        # the loss lines the reference printed, one per guided step (skipped steps print nothing and leave the log untouched)
        mine, last = [], None
        for _, _, log in got:
            if log and log is not last and _loss_line(log) != (mine[-1] if mine else None):
                mine.append(_loss_line(log))
            last = log
        ref_lines = meta["loss_lines"]
        assert len(mine) == len(ref_lines), (name, mine, ref_lines)
        for a, b in zip(mine, ref_lines):
            ka, kb = [p.split(": ")[0] for p in a.split("\t")], [p.split(": ")[0] for p in b.split("\t")]
            assert ka == kb, (name, a, b)
            for pa, pb in zip(a.split("\t"), b.split("\t")):
                va, vb = float(pa.split(": ")[1]), float(pb.split(": ")[1])
                assert abs(va - vb) <= 1e-4 * max(1.0, abs(vb)) + 2e-3, (name, pa, pb)


def test_dropin_paths_match_what_the_reference_generator_yielded():
    """(batch_idx, path) order and naming of the real generator (cgd.py:266-270, script_util.py:86-101) vs the drop-in's helpers."""
    import cgd_amd  # noqa: F401
    from cgd import script_util
    for name, meta in CONDFN_META.items():
        kw = meta["kwargs"]
        expect = []
        for step in range(meta["steps"]):
            for b in range(kw["batch_size"]):
                expect.append([b, os.path.join(script_util.clean_and_combine_prompts("out", kw["prompts"], b), f"{step:04}.png")])
        assert meta["yielded"] == expect[:len(meta["yielded"])], name

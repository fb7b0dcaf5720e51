"""The oracle's third-party half (UNet, Gaussian / spaced diffusion steps, CLIP ModifiedResNet, LPIPS-VGG16) against fixtures generated
from the REAL packages by tests/golden/make_golden_3p.py.

`tests/golden/reference_3p.npz` can only be produced where `guided_diffusion` (crowsonkb@fb47224), `clip` and `lpips` 0.1.4 are
installed — not in the offline build container, not on the GPU box — so `test_oracle_matches_third_party_fixtures` SKIPS while the file
is absent (DESIGN.md section 2 then says "parity unpinned" for those modules).  The pipeline itself is exercised here regardless:
`test_generator_pipeline_self_test` runs the generator with the oracle standing in for the packages and checks the comparison passes,
and that a perturbed oracle is caught."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, GOLD)
import make_golden_3p as mg  # noqa: E402

RTOL, ATOL = 1e-4, 1e-5  # fp32 CPU against fp32 CPU: far tighter than north_star's GPU tolerance


def compare(fixture_npz, results):
    bad = []
    data = np.load(fixture_npz)
    assert sorted(data.files) == sorted(results), "fixture keys differ from the cases of make_golden_3p.run_all"
    for k in data.files:
        ref, got = th.from_numpy(data[k]).double(), results[k].double()
        if ref.shape != got.shape:
            bad.append((k, "shape", tuple(ref.shape), tuple(got.shape)))
            continue
        scale = max(1.0, ref.abs().max().item())
        err = (ref - got).abs().max().item()
        exact = "timestep_map" in k or "model_ts" in k
        if err > (0.0 if exact else ATOL * scale + RTOL * ref.abs().max().item()):
            bad.append((k, err, scale))
    return bad


def test_oracle_matches_third_party_fixtures():
    path = os.path.join(GOLD, "reference_3p.npz")
    if not os.path.isfile(path):
        pytest.skip("tests/golden/reference_3p.npz absent: generate it with tests/golden/make_golden_3p.py where guided_diffusion, clip and "
                    "lpips are installed (the [3P] half of the oracle is parity-unpinned until then)")
    with open(os.path.join(GOLD, "reference_3p.json")) as f:
        meta = json.load(f)
    assert meta.get("source") == "packages", "reference_3p.* must come from the real packages, not from --self-test"
    bad = compare(path, mg.run_all(mg.ORACLE))
    assert not bad, bad


def test_generator_pipeline_self_test(tmp_path):
    stem = str(tmp_path / "selftest_3p")
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden_3p.py"), "--self-test", stem], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(stem + ".json") as f:
        assert json.load(f)["source"] == "self-test"
    res = mg.run_all(mg.ORACLE)
    assert not compare(stem + ".npz", res)
    # the comparison bites: a 1e-3 relative perturbation of one array is reported
    k = "unet_mini/grad_x"
    res[k] = res[k] * (1 + 1e-3)
    assert [b[0] for b in compare(stem + ".npz", res)] == [k]


def test_generator_reports_missing_packages_without_writing(tmp_path):
    missing = []
    for mod in ("guided_diffusion", "clip", "lpips"):
        try:
            __import__(mod)
        except ImportError:
            missing.append(mod)
    if not missing:
        pytest.skip("all three packages are importable here: run make_golden_3p.py and commit the fixture")
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden_3p.py")], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 2 and "cannot import" in r.stdout
    assert not os.path.exists(os.path.join(GOLD, "reference_3p.npz"))

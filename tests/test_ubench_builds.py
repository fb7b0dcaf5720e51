"""The timeline micro-benchmarks under benchmarks/ubench/ include kernel sources of the library with their stamp macros defined
(CGD_WCONV_STAMPS / CGD_HGEMM_STAMPS), code the library build never sees: cross-compile them for gfx950 so that the instrumentation
cannot rot unnoticed.  No GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,extra", [("wconv_stamps.hip", []), ("hgemm_stamps.hip", []), ("hgemm_stamps.hip", ["-DCGD_HGEMM_STAMPS=2"]),
                                       ("lgemm_bench.hip", []), ("lgemm_bench.hip", ["-DCGD_LGEMM_EXP=5"])])
def test_timeline_microbenchmarks_compile_for_gfx950(tmp_path, src, extra):
    out = tmp_path / "a.out"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), *extra,
           os.path.join(ROOT, "benchmarks", "ubench", src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and out.exists(), r.stderr[-2000:]

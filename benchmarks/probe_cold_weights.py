"""Probe: the weight-streaming conv kernel of the small maps (kconv_kernel + its split-K reduce) with WARM weights (the launch repeated back to back:
the packed weights sit in the 256 MB memory-side Infinity Cache) and with COLD weights (1 GB of other traffic between launches, the situation inside a step:
profiles/r4_mall_probe.txt).  One event pair per launch, median over the repetitions.  Usage: python benchmarks/probe_cold_weights.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
ctx = lib.Context(0, 1)
big = th.empty(256 << 20, device="cuda")  # 1 GB of floats


def timed(fn, cold):
    ts = []
    for i in range(reps):
        if cold:
            big.fill_(float(i))
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        th.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print(f"knobs: CGD_KCONV={os.environ.get('CGD_KCONV', 'default')} CGD_HGEMM={os.environ.get('CGD_HGEMM', 'default')}")
for (H, Ci, Co) in [(8, 1024, 1024), (16, 1024, 1024), (32, 512, 512), (16, 2048, 1024)]:
    x = th.randn(1, H, H, Ci, device="cuda")
    w = th.randn(Co, Ci, 3, 3, device="cuda") / (9 * Ci) ** 0.5
    wp = ops.pack_conv3x3(w)[0]
    wf = ops.pack_conv3x3_frag(ctx, w, dgrad=False)
    fn = lambda: ops.conv3x3(ctx, x, wp, None, force_tile=0, splitk=1, w_frag=wf)  # noqa: E731  (automatic kernel / split-K selection)
    for _ in range(3):
        fn()
    mb = Co * Ci * 9 * 4 / 1e6
    print(f"conv3x3 {H}x{H} {Ci}->{Co} ({mb:5.1f} MB of packed weights): warm {timed(fn, False):6.1f} us   cold {timed(fn, True):6.1f} us  (incl. the split-K reduce)", flush=True)

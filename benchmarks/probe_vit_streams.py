"""Probe: CLIP ViT-B/32 forward + dgrad over 16 cutouts in one stream vs two halves on two HIP streams (two contexts: each has its own
split-K workspace).  Usage: python benchmarks/probe_vit_streams.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, nets, synthetic  # noqa: E402

dev = "cuda:0"
ctx1, ctx2 = lib.Context(0, "bf16x3"), lib.Context(0, "bf16x3")
tA, tB = nets.ClipImageTower(ctx1, "ViT-B/32"), nets.ClipImageTower(ctx2, "ViT-B/32")
sd = synthetic.synthetic_state_dict(tA, seed=4321, device=dev)
tA.load_state_dict(sd)
tB.load_state_dict(sd)
N, g = 16, 7
patches = th.randn(N * g * g, 3 * 32 * 32, device=dev)
demb = th.randn(N, 512, device=dev)
emb = th.empty(N, 512, device=dev)
dp = th.empty_like(patches)
h = N // 2
s1, s2 = th.cuda.Stream(), th.cuda.Stream()


def single():
    tA.encode_image(patches, layout=1, n=N, out=emb)
    tA.dgrad(demb, dp)


def dual():
    main = th.cuda.current_stream()
    e0 = th.cuda.Event()
    e0.record(main)
    evs = []
    for s, t, lo, hi in ((s1, tA, 0, h), (s2, tB, h, N)):
        with th.cuda.stream(s):
            s.wait_event(e0)
            t.encode_image(patches[lo * g * g:hi * g * g], layout=1, n=hi - lo, out=emb[lo:hi])
            t.dgrad(demb[lo:hi], dp[lo * g * g:hi * g * g])
            e = th.cuda.Event()
            e.record(s)
            evs.append(e)
    for e in evs:
        main.wait_event(e)


for fn, name in ((single, "one stream, N=16"), (dual, "two streams, 2 x N=8")):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    th.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
ref = dp.clone()
single()
th.cuda.synchronize()
print("max |dual - single| on d_patches:", float((ref - dp).abs().max()))

"""Micro-benchmark of the MFMA implicit-GEMM conv kernel on the UNet's layer shapes, per tile variant.
Usage: python benchmarks/bench_ops.py [tag] [precision]   -> gpurun_out/ops_<tag>.json + table on stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

SHAPES_ALL = [  # (H=W, Cin, Cout, launches per step in config 2 (fwd+dgrad, approximate))
    (256, 256, 256, 18), (256, 512, 256, 3), (256, 256, 512, 3),
    (128, 256, 256, 22), (128, 512, 256, 3), (128, 256, 512, 3),
    (64, 256, 512, 2), (64, 512, 512, 14), (64, 1024, 512, 3), (64, 512, 1024, 3), (64, 768, 512, 1),
    (32, 512, 512, 14), (32, 1024, 512, 4), (32, 512, 1024, 4),
    (16, 512, 1024, 2), (16, 1024, 1024, 14), (16, 2048, 1024, 3), (16, 1024, 2048, 3), (16, 1536, 1024, 1),
    (8, 1024, 1024, 18), (8, 2048, 1024, 3), (8, 1024, 2048, 3),
]
SHAPES = SHAPES_ALL
TILES = [int(t) for t in os.environ["CGD_BENCH_TILES"].split(",")] if os.environ.get("CGD_BENCH_TILES") else [128, 5121, 5120]  # igemm 128x128 | hconv2 16x16 tiles | hconv2 8x16 tiles (2 workgroups / CU)  # 512x = halo-staged conv kernel, tile variant x


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "run"
    prec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = lib.Context(0, prec)
    os.makedirs("gpurun_out", exist_ok=True)
    res = []
    print(f"{'shape':<22s}" + "".join(f"{t:>14d}" for t in TILES) + "   (us | TFLOP/s algorithmic)")
    for (H, ci, co, cnt) in SHAPES:
        x = th.randn(1, H, H, ci, device="cuda")
        wt = th.randn(co, ci, 3, 3, device="cuda") * 0.02
        w = ops.pack_conv3x3(wt)[0]
        wfrag = ops.pack_conv3x3_frag(ctx, wt)
        b = th.randn(co, device="cuda")
        flop = 2.0 * H * H * co * 9 * ci
        row = {"H": H, "cin": ci, "cout": co, "count": cnt, "flop": flop, "us": {}}
        line = f"{H:>3d}^2 {ci:>4d}->{co:<4d} x{cnt:<3d}"
        for t in TILES:
            if t == 516:
                if H * H > 4096:
                    line += f"{'-':>14s}"
                    continue
            elif ((t % 1000) >= 128 and (H * H < 128 or co < 128)) or (t >= 5120 and H * H < 256):
                line += f"{'-':>14s}"
                continue
            tcode = t
            if t >= 5120:
                ctx.check(ctx.lib.cgd_set_hconv(ctx.h, 1 + 16 * (t - 5120), 256))
                tcode, t = 512, t
            try:
                for _ in range(3):
                    ops.conv3x3(ctx, x, w, b, force_tile=tcode, w_frag=wfrag)
                th.cuda.synchronize()
                e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                n = 10
                e0.record()
                for _ in range(n):
                    ops.conv3x3(ctx, x, w, b, force_tile=tcode, w_frag=wfrag)
                e1.record()
                th.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
                row["us"][str(t)] = us
                line += f"{us:>8.1f}|{flop / us / 1e6:>5.0f}"
            except Exception as e:  # noqa: BLE001
                row["us"][str(t)] = None
                line += f"{'ERR':>14s}"
                print("   ", type(e).__name__, str(e)[:150])
        print(line, flush=True)
        res.append(row)
    with open(f"gpurun_out/ops_{tag}.json", "w") as f:
        json.dump(res, f, indent=1)
    # best-per-shape projection
    if "516" in [str(t) for t in TILES]:
        tk = sum((r["us"].get("516") or 0) * r["count"] for r in res if r["us"].get("516"))
        t0 = sum((r["us"].get("0") or 0) * r["count"] for r in res if r["us"].get("516"))
        print(f"small-map layers: weight-streaming kernel (516) {tk / 1e3:.2f} ms/step vs automatic selection (0) {t0 / 1e3:.2f} ms/step")
    tot_best = sum(min([v for v in r["us"].values() if v] or [0]) * r["count"] for r in res)
    tot_cur = sum((r["us"].get("1256") or r["us"].get("128") or r["us"].get("64") or 0) * r["count"] for r in res)
    for code in [str(t) for t in TILES if t >= 5120]:
        tot_h = sum((r["us"].get(code) or r["us"].get("128") or r["us"].get("64") or 0) * r["count"] for r in res)
        print(f"halo conv kernel {code} wherever supported: {tot_h / 1e3:.2f} ms")
    print(f"projected conv time/step: default-ish {tot_cur / 1e3:.2f} ms, best-per-shape {tot_best / 1e3:.2f} ms")


if __name__ == "__main__":
    main()

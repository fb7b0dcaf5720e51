"""One-shot GPU parity report: runs every check of tests/parity_checks.py, keeps going after failures and writes
gpurun_out/diag_<tag>.json + a readable table on stdout.  Usage: python benchmarks/gpu_diag.py [tag] [group ...]"""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from tests import parity_checks as pc  # noqa: E402

GROUPS = {
    "gemm": [lambda: pc.check_gemm(0), lambda: pc.check_gemm(1), lambda: pc.check_gemm(2)],
    "conv": [lambda: pc.check_conv(0), lambda: pc.check_conv(1), pc.check_wconv],
    "norm": [pc.check_norm],
    "elem": [pc.check_elem],
    "attn": [lambda: pc.check_attn(0), lambda: pc.check_attn(1)],
    "guid": [pc.check_cutouts_loss],
    "unet_small": [lambda: pc.check_unet("mini", 0), lambda: pc.check_unet("mini", 1), lambda: pc.check_unet("mini128", 0),
                   lambda: pc.check_unet("mini64", 0), lambda: pc.check_unet("mini", 0, B=2, hw=(32, 48))],
    "vit": [lambda: pc.check_vit("ViT-B/32", 0), lambda: pc.check_vit("ViT-B/32", 1)],
    "resnet": [lambda: pc.check_resnet("tiny", 0, config=(64, 64, (1, 1, 1, 1), 128, 32)),
               lambda: pc.check_resnet("tiny", 1, config=(64, 64, (1, 1, 1, 1), 128, 32)),
               lambda: pc.check_resnet("RN50", 1),
               lambda: pc.check_resnet("x4-tiny", 0, config=(96, 80, (1, 1, 1, 1), 64, 40)),
               lambda: pc.check_resnet("x16-tiny", 1, config=(64, 96, (1, 1, 1, 1), 64, 48)),
               lambda: _sc().check_step("mini", 1, steps=2, B=2, rn_cfg=(64, 64, (1, 1, 1, 1), 128, 32))],
    "dual": [lambda: _sc().check_step("mini", 1, steps=2, B=2, P=2, dual=True),
             lambda: _sc().check_step("mini", 1, steps=2, rn_cfg=(64, 64, (1, 1, 1, 1), 128, 32), dual=True)],
    "lpips": [lambda: pc.check_lpips(0), lambda: pc.check_lpips(1),
              lambda: _sc().check_step("mini", 1, steps=2, B=2, init_scale=1000.0)],
    "vit_other": [lambda: pc.check_vit("ViT-B/16", 1, N=2), lambda: pc.check_vit("ViT-L/14", 1, N=2)],
    "unet64": [lambda: pc.check_unet("cfg64", 0), lambda: pc.check_unet("cfg64", 1)],
    "unet128": [lambda: pc.check_unet("cfg128", 1)],
    "unet256": [lambda: pc.check_unet("cfg256", 1)],
    "unet512": [lambda: pc.check_unet("cfg512", 1, timestep=417.5)],
    "step": [lambda: _sc().check_step("mini", 0, steps=4),
             lambda: _sc().check_step("mini", 1, steps=4),
             lambda: _sc().check_step("mini", 1, ddim=True, steps=4),
             lambda: _sc().check_step("mini", 1, steps=3, B=2, P=2, use_magnitude=True, sat_scale=3.0, counter_quirk=True),
             lambda: _sc().check_step("mini64", 1, respacing="25", schedule="cosine", steps=2, P=3, hw=(32, 48), use_magnitude=True),
             lambda: _sc().check_step("mini", 1, steps=6, cutn=16, reduce_clip=True, progressive_cutout=True, counter_quirk=True, t_first=10)],
    "headline": [lambda: _sc().check_headline_step(1)],
    "cfg5": [lambda: _sc().check_step("cfg256", 1, steps=1, hw=(256, 288), respacing="500", P=3, cutn=4)],
}


def _sc():
    from tests import step_checks
    return step_checks


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "run"
    groups = sys.argv[2:] or ["gemm", "conv", "norm", "elem", "attn", "guid", "unet_small", "vit"]
    os.makedirs("gpurun_out", exist_ok=True)
    print("device:", th.cuda.get_device_name(0), "| torch", th.__version__, flush=True)
    results = []
    for gname in groups:
        for fn in GROUPS[gname]:
            t0 = time.time()
            try:
                recs = fn()
            except Exception as e:  # keep going: one report per GPU call
                recs = [{"name": f"{gname}: EXCEPTION {type(e).__name__}: {e}", "err_abs": float("nan"), "err_rel": float("nan"),
                         "ref_max": 0.0, "ok": False, "trace": traceback.format_exc()[-1500:]}]
            dt = time.time() - t0
            for r in recs:
                r["group"] = gname
                r["secs"] = round(dt, 2)
                results.append(r)
                print(f"{'OK  ' if r['ok'] else 'FAIL'} {r['name']:<86s} abs {r['err_abs']:.3e} rel {r['err_rel']:.3e} ref {r['ref_max']:.3e}"
                      f" {r.get('criterion', '')}{'' if r.get('ok_strict', True) else ' (strict: no)'}{' VACUOUS' if r.get('vacuous') else ''}",
                      flush=True)
                if "trace" in r:
                    print(r["trace"], flush=True)
            with open(f"gpurun_out/diag_{tag}.json", "w") as f:
                json.dump(results, f, indent=1)
    nfail = sum(not r["ok"] for r in results)
    print(f"== {len(results) - nfail} ok, {nfail} failed ==")
    return 0


if __name__ == "__main__":
    sys.exit(main())

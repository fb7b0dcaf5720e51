"""SURVEY.md parity tier T5: short-trajectory drift report at BASELINE configs[0] — the 64x64 cosine-schedule checkpoint shape,
respace 25 (the WHOLE schedule, from t = 24), cutn 4, batch 1, CLIP ViT-B/32, -cgs 5 -tvs 0.00001 (/root/reference/README.md:113),
gradient-magnitude clamp on (automatic at 64x64, cgd.py:72-74) — GPU (C ABI) against the CPU oracle with a replayed RNG tape.

Not a pass/fail test: with clip_guidance_scale feedback the sampler amplifies 1e-6 perturbations, so two fp32 implementations
drift apart along a trajectory (SURVEY.md 7 "trajectory chaos"); per-step parity is tests/test_gpu_step.py.  This prints, per
step, the deviation of x_{t-1}, x0-hat and g relative to the tensor's peak, for the exact-fp32 MFMA mode and for bf16x3.
Usage (GPU box): python benchmarks/drift_report.py > gpurun_out/drift.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from tests import step_checks as sc  # noqa: E402


def report(title, scen):
    t0 = time.time()
    o_out = scen.run_oracle()
    print(f"\n# {title}\n# {scen.tag('-')}  oracle: {time.time() - t0:.1f} s for {scen.steps} steps on {th.get_num_threads()} threads")
    for precision, label in ((0, "f32 (exact MFMA products)"), (1, "bf16x3 (bench mode)")):
        print(f"\n## precision {label}\n step   t | sample: peak      max|d|/peak | x0-hat: peak     max|d|/peak | g: peak         max|d|/peak | CLIP loss (dev / oracle)")
        for k, (out, guid, legs) in enumerate(scen.run_device(precision)):
            o_s, o_x0, o_log, o_legs = o_out[k]
            d_s = (out["sample"].cpu().double() - o_s.double()).abs().max().item() / (o_s.abs().max().item() + 1e-30)
            d_x = (out["pred_xstart"].cpu().double() - o_x0.double()).abs().max().item() / (o_x0.abs().max().item() + 1e-30)
            d_g = (legs["g"].cpu().double() - o_legs["g"].double()).abs().max().item() / (o_legs["g"].abs().max().item() + 1e-30)
            print(f" {k:4d} {scen.t_first - k:3d} | {o_s.abs().max().item():12.4e} {d_s:10.2e} | {o_x0.abs().max().item():12.4e} {d_x:10.2e} | "
                  f"{o_legs['g'].abs().max().item():12.4e} {d_g:10.2e} | {guid.log()['CLIP Loss']:.4f} / {o_log['CLIP Loss']:.4f}", flush=True)


def main():
    th.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    common = dict(vit_name="ViT-B/32", cutn=4, schedule="cosine", steps=25, scales=(5.0, 1e-5, 50.0), use_magnitude=True, head_scale=0.1)
    # (A) the literal configuration: the whole respace-25 schedule from t = 24.  Synthetic weights do not predict epsilon: at
    #     t = T-1 of the cosine schedule x0-hat = 640 (x - eps-hat), so these tensors are huge and saturated (both sides alike, the
    #     magnitude clamp keeps g * factor bounded); the table shows that the implementations still agree to ~1e-6 of the peak
    report("(A) config 1 as written: respace 25, all 25 steps from t = 24 (saturated trajectory, see the script header)",
           sc.Scenario("cfg64", respacing="25", t_first=24, **common))
    # (B) the same shape and scales on a live trajectory: 25 chained steps over the tame half of the schedule (respace 50,
    #     skip_timesteps 25, init image), where x0-hat is O(1) and the CLIP feedback acts on the sample
    report("(B) same shape / scales, 25 chained steps t = 24..0 of respace 50 (skip_timesteps 25 + init image): live CLIP feedback",
           sc.Scenario("cfg64", respacing="50", t_first=24, **common))


if __name__ == "__main__":
    main()

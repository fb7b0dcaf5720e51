import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import bench
import cgd_amd
from cgd_amd import lib
ctx = lib.Context(0, sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
unet, clip, smp, guid = bench.build_device(ctx, 0, 1, "bf16x3")
th.manual_seed(1000)
y0 = th.zeros(1, dtype=th.long, device="cuda")
gen = smp.p_sample_loop_progressive(unet, (1, 3, 256, 256), clip_denoised=False, model_kwargs={"y": y0}, cond_fn=guid, device="cuda",
                                    progress=False, randomize_class=True, cond_fn_with_grad=True)
guid.current_timestep = smp.num_timesteps - 1
for k, out in enumerate(gen):
    guid.current_timestep -= 1
    s = out["sample"]; x0 = out["pred_xstart"]
    g = guid._buf["g"]; sc = guid.scalars.tolist()
    print(k, f"sample max {s.abs().max().item():.3e} x0 max {x0.abs().max().item():.3e} g max {g.abs().max().item():.3e} gunet {guid._buf['gunet'].abs().max().item():.3e} gclip {guid._buf['gclip'].abs().max().item():.3e} emb {guid.emb.abs().max().item():.3e} scal {[round(v,3) for v in sc[:6]]}", flush=True)
    if not th.isfinite(s).all() or k >= 40:
        break

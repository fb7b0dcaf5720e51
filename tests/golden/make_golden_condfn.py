"""Generates tests/golden/reference_condfn.npz|json by running the REAL reference generator — /root/reference/cgd/cgd.py
`clip_guided_diffusion`, i.e. its prompt / weight handling, MakeCutouts, the `cond_fn` closure (cgd.py:151-239), the
`current_timestep` bookkeeping and the reduce_clip / progressive_cutout / cached_cutouts logic, unmodified — in the build container.

What is NOT the reference's here (the packages are not installed): the UNet, the diffusion loops and the CLIP tower are
the oracle's restatements (oracle/unet.py, oracle/diffusion.py, oracle/clip_vit.py) handed to the reference through its own
`load_guided_diffusion` / `load_clip` seams; text embeddings are fixed vectors; torchvision / clip / lpips / wandb are inert stubs
(`Compose([])` identity, `Normalize` as a plain function, `to_pil_image`).  The fixture therefore pins the oracle's restated cond_fn,
cutouts and generator bookkeeping (oracle/guidance.py, tests/condfn_replay.py) against the reference's own code on identical
networks and identical global-RNG draws.

Run:  python tests/golden/make_golden_condfn.py      (from the repo root, in the build container)
"""
import importlib
import json
import os
import sys
import tempfile

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    import make_golden  # the stub installer of the other fixtures
    from oracle import diffusion as od
    from tests import condfn_replay as cr
    make_golden.install_stubs()
    tvt, tvf = sys.modules["torchvision.transforms"], sys.modules["torchvision.transforms.functional"]

    class Normalize:  # torchvision.transforms.Normalize on (N,3,H,W) tensors
        def __init__(self, mean, std):
            self.mean, self.std = th.tensor(mean).view(1, 3, 1, 1), th.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    def to_pil_image(t):
        from PIL import Image
        return Image.fromarray(t.mul(255).byte().permute(1, 2, 0).numpy())

    class ToTensor:  # torchvision.transforms.ToTensor on an RGB PIL image
        def __call__(self, pil):
            return th.from_numpy(np.array(pil)).permute(2, 0, 1).float().div(255)

    tvt.Normalize, tvt.ToTensor, tvf.to_pil_image = Normalize, ToTensor, to_pil_image
    sys.modules["lpips"].LPIPS = lambda net: cr.build_lpips()  # the oracle's LPIPS-VGG16 with seeded synthetic weights
    sys.path.insert(0, make_golden.REF)
    ref = importlib.import_module("cgd.cgd")
    assert ref.__file__.startswith(make_golden.REF), ref.__file__

    lines = []

    class Quiet:  # stands in for tqdm inside the reference module: collects the loss lines
        @staticmethod
        def write(s):
            lines.append(s)

    ref.tqdm = Quiet
    fx, meta = {}, {}
    for name in cr.CASES:
        kw, steps = cr.case_kwargs(name)
        unet, clip = cr.build_models()
        recorded = []

        class Tee:  # the oracle's diffusion object; records what its loops yield to the reference generator
            def __init__(self, d):
                self._d = d

            def __getattr__(self, k):
                return getattr(self._d, k)

            def _tee(self, gen):
                for o in gen:
                    recorded.append((o["sample"].clone(), o["pred_xstart"].clone()))
                    yield o

            def p_sample_loop_progressive(self, *a, **k):
                return self._tee(self._d.p_sample_loop_progressive(*a, **k))

            def ddim_sample_loop_progressive(self, *a, **k):
                return self._tee(self._d.ddim_sample_loop_progressive(*a, **k))

        def load_guided_diffusion(**k):
            return unet, Tee(od.create_gaussian_diffusion(k["diffusion_steps"], k["noise_schedule"], k["timestep_respacing"], False))

        ref.script_util.download_guided_diffusion = lambda **k: "unused"
        ref.script_util.load_guided_diffusion = load_guided_diffusion
        ref.clip_util.load_clip = lambda model_name, device: (clip, cr.VIT[0])
        ref.clip_util.encode_text_prompt = lambda txt, weight, model_name, device: (cr.text_embedding(txt), weight)
        del lines[:]
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)  # the reference writes ./current.png
            if kw.get("init_image"):
                from PIL import Image
                Image.fromarray(cr.init_image_array()).save(os.path.join(d, "init.png"))
                kw = dict(kw, init_image=os.path.join(d, "init.png"))
            try:
                gen = ref.clip_guided_diffusion(prefix_path=os.path.join(d, "out"), checkpoints_dir=os.path.join(d, "ckpt"), **kw)
                yielded = []
                for item in gen:
                    yielded.append([int(item[0]), os.path.relpath(item[1], d)])
                    if len(recorded) >= steps and len(yielded) >= steps * kw["batch_size"]:
                        break
            finally:
                os.chdir(cwd)
        fx[f"{name}/sample"] = th.stack([s for s, _ in recorded[:steps]]).numpy()
        fx[f"{name}/pred_xstart"] = th.stack([x for _, x in recorded[:steps]]).numpy()
        meta[name] = {"kwargs": {k: (v if k != "init_image" else "<init.png>") for k, v in kw.items()}, "steps": steps, "yielded": yielded[:steps * kw["batch_size"]],
                      "loss_lines": [ln for ln in lines if "CLIP Loss" in ln]}
        # self-check: the oracle replay of the same case, here and now
        mine = cr.replay_with_oracle(name)
        err = max((th.from_numpy(fx[f"{name}/sample"][k]) - mine[k][0]).abs().max().item() for k in range(steps))
        print(f"{name}: {steps} steps, {len(meta[name]['loss_lines'])} guided; oracle replay max |d sample| = {err:.3e}")
    np.savez_compressed(os.path.join(OUT, "reference_condfn.npz"), **fx)
    with open(os.path.join(OUT, "reference_condfn.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote reference_condfn.npz / .json")


if __name__ == "__main__":
    main()

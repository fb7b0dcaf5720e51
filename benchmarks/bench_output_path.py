"""End-to-end rate of the drop-in generator WITH per-step PNG output (the reference CLI default --save_frequency 1), 256x256,
respace 250, cutn 16, ViT-B/32, synthetic weights: frames per second between the 5th and the last yielded item.
Not the headline benchmark (bench.py excludes PNG writes); it documents the pipelined output path of cgd/cgd.py.
Usage: CGD_SYNTHETIC_WEIGHTS=1 python benchmarks/bench_output_path.py [items]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CGD_SYNTHETIC_WEIGHTS", "1")
import cgd_amd  # noqa: E402,F401
from cgd.cgd import clip_guided_diffusion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with tempfile.TemporaryDirectory() as d:
    os.chdir(d)
    gen = clip_guided_diffusion(prompts=["a painting"], image_size=256, num_cutouts=16, timestep_respacing="250", prefix_path=os.path.join(d, "out"),
                                checkpoints_dir=os.path.join(d, "ckpt"), save_frequency=1, progress=False, device="cuda")
    t0 = None
    for k, (b, path) in enumerate(gen):
        if k == 4:
            t0 = time.perf_counter()
        if k == n - 1:
            break
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "generator items/s with a PNG pair written per step (save_frequency 1)", "items_per_sec": round((n - 5) / dt, 2),
                      "ms_per_item": round(dt / (n - 5) * 1e3, 2), "items": n - 5}))

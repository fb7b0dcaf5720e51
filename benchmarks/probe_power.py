"""Clock / power while the halo conv kernel runs back to back (is the kernel power-limited?).
Samples `rocm-smi --showclocks --showpower` from a thread while one UNet layer shape is launched in a loop.
Usage: python benchmarks/probe_power.py [seconds per shape]"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib, ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out = f"ERR {e}"
        sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", out)
        pw = re.findall(r"Power \(W\):\s*([\d.]+)", out)
        samples.append((time.time(), sclk[:1], pw[:1]))
        time.sleep(0.2)


def run(ctx, H, ci, co, var):
    ctx.check(ctx.lib.cgd_set_hconv(ctx.h, 1 + 16 * var, 256))
    x = th.randn(1, H, H, ci, device="cuda")
    wt = th.randn(co, ci, 3, 3, device="cuda") * 0.02
    w = ops.pack_conv3x3(wt)[0]
    wfrag = ops.pack_conv3x3_frag(ctx, wt)
    b = th.randn(co, device="cuda")
    th.cuda.synchronize()
    t0, n = time.time(), 0
    del samples[:]
    while time.time() - t0 < SECS:
        for _ in range(200):
            ops.conv3x3(ctx, x, w, b, force_tile=512, w_frag=wfrag)
        th.cuda.synchronize()
        n += 200
    dt = time.time() - t0
    mid = [s for s in samples if s[0] > t0 + 0.5]
    print(f"{H}^2 {ci}->{co} var{var}: {dt / n * 1e6:.1f} us/launch, {2.0 * H * H * co * 9 * ci * n / dt / 1e12:.0f} TFLOP/s algorithmic | "
          f"sclk {[s[1] for s in mid][:12]} | W {[s[2] for s in mid][:12]}", flush=True)


if __name__ == "__main__":
    ctx = lib.Context(0, 1)
    th_ = threading.Thread(target=sampler, daemon=True)
    th_.start()
    time.sleep(1.0)
    print("idle:", samples[-3:], flush=True)
    for var in (0, 4):
        for (H, ci, co) in [(256, 256, 256), (128, 256, 256), (64, 512, 512)]:
            run(ctx, H, ci, co, var)
    stop.set()
    print(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-1500:])

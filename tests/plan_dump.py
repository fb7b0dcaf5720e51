"""Launch plan of the contraction kernels for one guided step, derived on the CPU: the oracle UNet / CLIP tower run on the meta device
with hooks that record every conv3x3 / 1x1 conv / linear shape, and `cgd_op_plan` (the launcher's own selection code, host-only)
says which kernel, tile and split-K each forward and backward-to-input launch gets.  No GPU needed.
Usage: python tests/plan_dump.py [--csv]"""
import ctypes as C
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import cgd_amd  # noqa: E402,F401
from cgd_amd import lib  # noqa: E402

KERNELS = {0: "igemm", 1: "hconv2", 2: "hgemm", 3: "gemv", 4: "kgemm"}


def record_shapes(net, *args, call=None):
    """[(kind, M, N, K | (H, W, Cin))] for every Conv2d / Conv1d / Linear executed, in call order (meta device: nothing is computed)."""
    recs = []

    def hook(mod, inp, out):
        x = inp[0]
        if isinstance(mod, th.nn.Conv2d) and mod.kernel_size == (3, 3):
            recs.append(("conv3x3", x.shape[0] * out.shape[2] * out.shape[3], mod.out_channels, (out.shape[2], out.shape[3], mod.in_channels)))
        elif isinstance(mod, (th.nn.Conv2d, th.nn.Conv1d)):
            recs.append(("conv1x1", out.numel() // mod.out_channels, mod.out_channels, mod.in_channels * mod.kernel_size[0] * mod.kernel_size[-1]))
        elif isinstance(mod, th.nn.MultiheadAttention):  # in_proj / out_proj run through F.linear inside the module: no Linear hook fires
            rows, e = x.shape[0] * x.shape[1], mod.embed_dim
            recs.append(("linear", rows, 3 * e, e))
            recs.append(("linear", rows, e, e))
        elif isinstance(mod, th.nn.Linear):
            recs.append(("linear", x.numel() // mod.in_features, mod.out_features, mod.in_features))

    kinds = (th.nn.Conv2d, th.nn.Conv1d, th.nn.Linear, th.nn.MultiheadAttention)
    hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, kinds)]
    with th.no_grad():
        (call or (lambda n, *a: n(*a)))(net, *args)
    for h in hs:
        h.remove()
    return recs


def plan(handle, kind, M, N, K, precision=1, num_cu=256):
    out = (C.c_int * 4)()
    if kind == "conv3x3":
        H, W, Cin = K
        rc = handle.cgd_op_plan(1, M, N, 0, H, W, Cin, 1, precision, num_cu, out)
    else:
        rc = handle.cgd_op_plan(0, M, N, K, 0, 0, 0, 1, precision, num_cu, out)
    return rc, tuple(out)


def step_plan(cutn=16):
    """Rows (net, kind, direction, M, N, K, kernel, tile, splitk, workgroups, GFLOP) of BASELINE config 2."""
    import bench
    from oracle import clip_vit as ocv
    from oracle import unet as ou
    handle = lib.load()
    rows = []
    with th.device("meta"):
        unet = ou.UNetModel(**bench.U256).eval()
        u = record_shapes(unet, th.empty(1, 3, 256, 256), th.zeros(1), th.zeros(1, dtype=th.long))
        clip = ocv.ClipImageModel("ViT-B/32").eval()
        v = record_shapes(clip, th.empty(cutn, 3, 224, 224), call=lambda n, x: n.encode_image(x))
    for net, recs in (("unet", u), ("vit", v)):
        for kind, M, N, K in recs:
            kk = 9 * K[2] if kind == "conv3x3" else K
            for direction in ("fwd", "dgrad"):
                if kind == "conv3x3":
                    n_, k_ = (N, K) if direction == "fwd" else (K[2], (K[0], K[1], N))
                else:
                    n_, k_ = (N, K) if direction == "fwd" else (K, N)
                rc, (kern, tile, sk, wg) = plan(handle, kind, M, n_, k_)
                # 515: the Winograd halo kernel, 516: the weight-streaming halo kernel of the small maps
                name = ("wconv" if tile == 515 else "kconv" if tile == 516 else KERNELS[kern]) if rc == 0 else f"n/a({rc})" 
                rows.append((net, kind, direction, M, n_, k_, name, tile, sk, wg, 2.0 * M * N * kk / 1e9))
    return rows


if __name__ == "__main__":
    rows = step_plan()
    if "--csv" in sys.argv:
        print("net,kind,direction,M,N,K,kernel,tile,splitk,workgroups,gflop")
        for r in rows:
            print(",".join(str(x).replace(",", "x").replace(" ", "") for x in r))
    else:
        agg = Counter()
        flop = Counter()
        split = Counter()
        for r in rows:
            agg[r[6]] += 1
            flop[r[6]] += r[10]
            split[r[6]] += r[8] > 1
        print(f"{'kernel':<10s}{'launches':>10s}{'with split-K':>14s}{'GFLOP':>12s}")
        for k in sorted(agg):
            print(f"{k:<10s}{agg[k]:>10d}{split[k]:>14d}{flop[k]:>12.1f}")
        print(f"{'total':<10s}{sum(agg.values()):>10d}{sum(split.values()):>14d}{sum(flop.values()):>12.1f}")

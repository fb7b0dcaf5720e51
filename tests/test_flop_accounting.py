"""The algorithmic work that bench.py (`--config 2..5`) prices their rates with (FLOP = 2 * (MAC_fwd + MAC_dgrad),
MAC_dgrad = MAC_fwd + the attention matmuls once more, no weight gradients, no elementwise work) re-derived from the oracle networks
with torch's FlopCounterMode on the meta device (no arithmetic is executed)."""
import os
import sys

import pytest
import torch as th
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def macs(build, *shapes, call=None):
    """(total GMAC, attention-matmul GMAC) of one forward of the module returned by build(), on the meta device."""
    with th.device("meta"):
        net = build().eval()
        args = [th.empty(*s) if isinstance(s, tuple) else s for s in shapes]
        with FlopCounterMode(display=False) as fc, th.no_grad():
            (call or (lambda n, *a: n(*a)))(net, *args)
    glob = {str(k): v for k, v in fc.get_flop_counts()["Global"].items()}
    bmm = sum(v for k, v in glob.items() if "bmm" in k or "baddbmm" in k)
    return fc.get_total_flops() / 2e9, bmm / 2e9


def unet_macs(cfg, H, W):
    from oracle import unet as ou
    with th.device("meta"):
        t, y = th.zeros(1), th.zeros(1, dtype=th.long)
    return macs(lambda: ou.UNetModel(**cfg), (1, 3, H, W), t, y if cfg.get("num_classes") else None)


def vit_macs(name):
    from oracle import clip_vit as ocv
    return macs(lambda: ocv.ClipImageModel(name), (1, 3, 224, 224), call=lambda n, x: n.encode_image(x))


def test_headline_flop_per_step_matches_the_oracle_networks():
    import bench
    u_fwd, u_att = unet_macs(bench.U256, 256, 256)
    assert u_fwd == pytest.approx(1119.8, abs=0.1) and u_att == pytest.approx(6.09, abs=0.01)
    v_fwd, v_att = vit_macs("ViT-B/32")
    assert v_fwd == pytest.approx(4.409, abs=0.001)
    u_dgrad, v_dgrad = u_fwd + u_att, v_fwd + v_att  # attention backward: 4 matmuls instead of 2, everything else 1:1
    assert u_dgrad == pytest.approx(1125.9, abs=0.1) and v_dgrad == pytest.approx(4.455, abs=0.002)
    flop = 2e9 * (u_fwd + u_dgrad) + 16 * 2e9 * (v_fwd + v_dgrad)
    assert flop == pytest.approx(bench.CONFIGS[2]["tflop"] * 1e12, rel=1e-3)  # 4.775 TFLOP per guided step (BASELINE config 2)


def test_flop_per_step_of_the_other_configs():
    from oracle import clip_resnet as ocr
    from oracle import lpips_vgg as olp
    import bench as bc
    u256 = sum(unet_macs(bc.U256, 256, 256)) + unet_macs(bc.U256, 256, 256)[0]          # fwd + dgrad
    # config 3: 32 cutouts through ViT-B/16
    v16 = vit_macs("ViT-B/16")
    c3 = 2e9 * (u256 + 32 * (2 * v16[0] + v16[1])) / 1e12
    assert c3 == pytest.approx(bc.CONFIGS[3]["tflop"], rel=0.01)
    # config 4: 512^2 UNet, 64 cutouts through ViT-B/32, LPIPS-VGG16 forward on x_in (the init image's features are cached) + dgrad
    u512 = unet_macs(bc.U512, 512, 512)
    v32 = vit_macs("ViT-B/32")
    vgg = macs(lambda: olp.LpipsVGG(), (1, 3, 512, 512), call=lambda n, x: n.features(x))[0]
    c4 = 2e9 * (2 * u512[0] + u512[1] + 64 * (2 * v32[0] + v32[1]) + 2 * vgg) / 1e12
    assert c4 == pytest.approx(bc.CONFIGS[4]["tflop"], rel=0.03)
    # config 5: 256x288 UNet, 16 cutouts through RN50 and ViT-L/14
    u288 = unet_macs(bc.U256, 256, 288)
    l14 = vit_macs("ViT-L/14")
    rn = macs(lambda: ocr.ClipResNetImageModel("RN50"), (1, 3, 224, 224), call=lambda n, x: n.encode_image(x))
    c5 = 2e9 * (2 * u288[0] + u288[1] + 16 * (2 * l14[0] + l14[1]) + 16 * (2 * rn[0] + rn[1])) / 1e12
    assert c5 == pytest.approx(bc.CONFIGS[5]["tflop"], rel=0.03)


def test_launch_plan_agrees_with_the_committed_bench_line():
    """tests/plan_dump.py (oracle shapes + the launcher's own selection code, both on the CPU) against what the GPU run recorded in
    profiles/r1_bench_1gpu.json (round 1: 136 halo-kernel launches per guided step carrying 4.18 of the 4.775 TFLOP; since round 3 the
    8x8 convs run on a halo kernel too: 168 launches, 4.23 TFLOP)."""
    import json
    from tests import plan_dump
    rows = plan_dump.step_plan()
    wconv = [r for r in rows if r[6] == "wconv"]    # Winograd halo kernel: the >= 128x128-pixel levels
    hconv = [r for r in rows if r[6] == "hconv2"]   # direct halo kernel: the 64x64 level
    kconv = [r for r in rows if r[6] == "kconv"]    # weight-streaming halo kernel: 32x32 .. 8x8 (round 3)
    assert len(wconv) == 52 and all(r[1] == "conv3x3" and r[3] >= 16384 and r[8] == 1 for r in wconv)
    assert len(hconv) == 28 and all(r[1] == "conv3x3" and r[3] == 4096 for r in hconv)
    assert len(kconv) == 88 and all(r[1] == "conv3x3" and 64 <= r[3] <= 1024 for r in kconv)
    gflop = sum(r[10] for r in wconv + hconv + kconv)
    assert gflop == pytest.approx(4225.4, abs=0.5)   # 4.23 of the 4.775 TFLOP of a step run on the three halo kernels
    assert sum(r[10] for r in wconv) == pytest.approx(3247.0, abs=0.5)
    assert sum(r[10] for r in kconv) == pytest.approx(418.0, abs=0.5)
    # split-K over channel chunks; (round 5) kconv on 8 x 8-pixel tiles: twice the pixel tiles, so 58 of its 88 launches split (rounds 3-4: 81)
    assert sum(1 for r in hconv if r[8] > 1) == 26 and sum(1 for r in kconv if r[8] > 1) == 58
    assert max(r[8] for r in kconv) == 8  # K is split inside the workgroup first: at most 8 slices (hconv2 / igemm needed up to 32)
    # 16-row tiles while they fill the chip (256x256: 512 / 1024 workgroups), 8-row tiles on the 128x128 level (256 / 384)
    assert {(r[3], r[9]) for r in wconv} == {(65536, 512), (65536, 1024), (16384, 256), (16384, 384)}
    import bench
    algo_w = sum(4 * r[3] * (r[5][2] + r[4]) + 4 * 12 * r[5][2] * r[4] for r in wconv) / len(wconv)
    algo_h = sum(4 * r[3] * (r[5][2] + r[4]) + 4 * 9 * r[5][2] * r[4] for r in hconv) / len(hconv)
    algo_k = sum(4 * r[3] * (r[5][2] + r[4]) + 4 * 9 * r[5][2] * r[4] for r in kconv) / len(kconv)
    assert algo_w == pytest.approx(bench.WCONV_ALGO_BYTES_PER_LAUNCH, rel=1e-3)
    assert algo_h == pytest.approx(bench.HCONV_ALGO_BYTES_PER_LAUNCH, rel=1e-3)
    assert algo_k == pytest.approx(bench.KCONV_ALGO_BYTES_PER_LAUNCH, rel=1e-3)
    with open(os.path.join(ROOT, "profiles", "r1_bench_1gpu.json")) as f:
        roof = json.loads(f.read().strip().splitlines()[-1])["roofline"]
    assert roof["launches_per_step"] == pytest.approx(136.0)  # round 1: 136 launches (16x16 .. 256x256), all on hconv2_kernel
    r1 = sum(r[10] for r in wconv + hconv + kconv if r[3] >= 256)
    assert roof["flop_per_launch"] * roof["launches_per_step"] == pytest.approx(r1 * 1e9, rel=1e-3)
    # round 4 line: the same 52 wconv launches (now counted apart: plain / with the GroupNorm-backward epilogue), library launch counters
    with open(os.path.join(ROOT, "profiles", "r4_bench_1gpu.json")) as f:
        r4 = json.loads(f.read().strip().splitlines()[-1])
    assert r4["roofline"]["launches_per_step"] == pytest.approx(len(wconv))
    cls = r4["roofline"]["launch_classes"]
    assert cls["plain"]["launches_per_step"] + cls["with_groupnorm_backward_epilogue"]["launches_per_step"] == pytest.approx(len(wconv))
    # dgrad convs of plain ResBlocks (both convs) and of the resampling blocks (conv2 only) on wconv: 26 dgrad launches, 3 of them conv1 of up / down blocks
    assert cls["with_groupnorm_backward_epilogue"]["launches_per_step"] == pytest.approx(23.0)
    assert r4["config"]["launches_per_step"] < 1065 and r4["config"]["splitk_reduce_per_step"] == pytest.approx(181.0)
    assert r4["roofline"]["traffic_source"].startswith("profiles/pmc_traffic.json")
    # every ViT linear (16 cutouts = 800 token rows) goes to the weight GEMM kernel; the 8x8-pixel convs and M = 1 embeddings do not
    vit = [r for r in rows if r[0] == "vit" and r[1] == "linear" and r[3] == 800]
    assert len(vit) == 96 and all(r[6] == "hgemm" for r in vit)
    assert all(r[6] == "kconv" for r in rows if r[1] == "conv3x3" and r[3] == 64 and r[7] != 0)  # the 8x8 convs left igemm in round 3

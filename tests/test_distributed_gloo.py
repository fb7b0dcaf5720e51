"""N>1 path on CPU: world_size 2 over gloo.  One weight broadcast, per-rank slices of the global RNG tape, no other
collective."""
import os
import sys

import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cgd_amd  # noqa: F401
    from cgd_amd import shard, synthetic
    specs = [("a.weight", 1000), ("a.bias", 10), ("ln_1.weight", 64)]
    numel = sum(n for _, n in specs)
    calls = []

    def make():
        calls.append(1)
        g = th.Generator().manual_seed(7)
        return th.randn(numel, generator=g)

    flat = shard.broadcast_flat(make, numel, "cpu")
    sd = synthetic.flat_unpack(flat, specs)
    # global tape for a batch of 2 (one sample per rank)
    g = th.Generator().manual_seed(0)
    tape = {"x_T": th.randn(2, 3, 8, 8, generator=g), "noise": [th.randn(2, 3, 8, 8, generator=g) for _ in range(3)],
            "y": [th.randint(0, 1000, (2,), generator=g) for _ in range(3)], "coords": [[(0, 0, 8)]] * 3}
    idx = shard.rank_samples(2)
    mine = shard.slice_tape(tape, idx)
    imgs = shard.gather_images(mine["x_T"])
    q.put((rank, len(calls), float(flat.double().sum()), {k: tuple(v.shape) for k, v in sd.items()}, idx,
           bool(th.equal(mine["x_T"], tape["x_T"][idx])), bool(th.equal(mine["noise"][2], tape["noise"][2][idx])),
           None if imgs is None else [float(i.sum()) for i in imgs], float(tape["x_T"][0].sum()), float(tape["x_T"][1].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_tape_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, c0, s0, shp0, i0, ok0a, ok0b, g0, t0, t1), (r1, c1, s1, shp1, i1, ok1a, ok1b, g1, _, _) = res
    assert (c0, c1) == (1, 0)            # only the source rank materialises the weights
    assert s0 == s1 and shp0 == shp1     # identical parameters everywhere after ONE broadcast
    assert i0 == [0] and i1 == [1]       # one sample per rank
    assert ok0a and ok0b and ok1a and ok1b
    assert g1 is None and abs(g0[0] - t0) < 1e-6 and abs(g0[1] - t1) < 1e-6


def test_bench_py_gpus_n_spawns_n_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start 2 ranks itself (one process per GPU), rendezvous, broadcast the
    weights once and report n_gpus 2 with the max over ranks.  No GPU here: CGD_BENCH_DRYRUN runs the launcher / collective
    plumbing over gloo without device work (the GPU flow of the same code path is tests/test_gpu_step.py)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CGD_BENCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, out.stdout  # exactly one JSON line: rank 0's
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["config"]["world_size_checked"] == 2 and res["config"]["ranks_reporting"] == 2
    assert res["config"]["weights_checksum"] == 499500.0 and res["config"]["max_over_ranks_s"] >= 0.02
    # a launcher that disagrees with --gpus is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_rank_samples_partition():
    import cgd_amd  # noqa: F401
    from cgd_amd import shard
    for B, n in [(8, 8), (4, 8), (7, 3), (1, 1)]:
        parts = [shard.rank_samples(B, r, n) for r in range(n)]
        assert sorted(sum(parts, [])) == list(range(B))


def test_one_sample_per_rank_reproduces_the_batched_run():
    """samples never interact (per-sample GroupNorm, losses summed over the batch, the same cutout boxes for every
    batch element), so rank b running sample b of the global RNG tape alone must reproduce slice b of the batched trajectory.  Oracle
    networks on the CPU; the magnitude clamp and the saturation mean reduce over the whole batch and are the documented exceptions."""
    sys.path.insert(0, ROOT)
    import cgd_amd  # noqa: F401
    from cgd_amd import shard
    from oracle import diffusion as od
    from oracle import guidance as og
    from tests import condfn_replay as cr
    from tests import step_checks as sc
    unet, clip = cr.build_models()
    B, H, W, steps, cutn = 2, 32, 48, 3, 3
    tape = sc.make_tape(B, H, W, steps, cr.UNET["num_classes"], cutn, cr.VIT[0])
    targets = th.randn(1, cr.VIT[5], generator=th.Generator().manual_seed(5))

    def run(tp, nb, **kw):
        diff = od.create_gaussian_diffusion(1000, "linear", "25", False)
        cond, state = og.make_cond_fn(diffusion=diff, clip_model=clip, make_cutouts=og.MakeCutouts(cr.VIT[0], cutn), target_embeds=targets,
                                      weights=th.tensor([1.0]), num_cutouts=cutn, coords_tape=tp["coords"], **kw)
        gen = diff.p_sample_loop_progressive(unet, (nb, 3, H, W), clip_denoised=False, cond_fn=cond, model_kwargs={"y": th.zeros(nb, dtype=th.long)},
                                             device="cpu", skip_timesteps=diff.num_timesteps - steps, randomize_class=True,
                                             cond_fn_with_grad=True, tape=tp)
        state["current_timestep"] = diff.num_timesteps - 1
        outs = []
        for o in gen:
            state["current_timestep"] -= 1
            outs.append(o["sample"].clone())
        return outs

    full = run(tape, B)
    for rank in range(B):
        idx = shard.rank_samples(B, rank, B)
        alone = run(shard.slice_tape(tape, idx), len(idx))
        for k in range(steps):
            # batch-1 and batch-2 convolutions round differently on the CPU (1e-6 .. 2e-5 of the peak after the guidance feedback)
            diff, peak = (alone[k] - full[k][idx]).abs().max().item(), full[k][idx].abs().max().item()
            assert diff <= 1e-4 * max(1.0, peak), (rank, k, diff, peak)
    # the documented exception: the magnitude clamp couples the samples of a batch (cgd.py:229-232)
    coupled = run(tape, B, use_magnitude=True)
    alone0 = run(shard.slice_tape(tape, [0]), 1, use_magnitude=True)
    assert (alone0[-1] - coupled[-1][[0]]).abs().max().item() > 1e-3 * coupled[-1][[0]].abs().max().item()


def test_sampler_sharded_draws_are_rows_of_the_global_draws(monkeypatch):
    """cgd_amd.sampler with `shard = (rows, global_batch)`: x_T, the per-step noise and the class ids of a rank are the rank's rows
    of the tensors a single-process batched run draws from the same seed (CPU tensors, the step itself stubbed out)."""
    sys.path.insert(0, ROOT)
    import types
    import cgd_amd  # noqa: F401
    from cgd_amd import diffusion as dd
    from cgd_amd import sampler
    tables = dd.create_gaussian_diffusion(1000, "linear", "10")

    def run(shard, nb):
        smp = sampler.GuidedSampler(types.SimpleNamespace(device=0), tables)
        smp.shard = shard
        seen = []

        def fake_step(model, x, i, cond_fn, model_kwargs, noise, mode, bufs):
            seen.append((x.clone(), smp._draw_like(x), model_kwargs["y"].clone()))
            return {"sample": x * 0.5, "pred_xstart": x}

        smp._step = fake_step
        th.manual_seed(11)
        model = types.SimpleNamespace(num_classes=1000)
        gen = smp.p_sample_loop_progressive(model, (nb, 3, 4, 4), clip_denoised=False, cond_fn=None, model_kwargs={"y": th.zeros(nb, dtype=th.long)},
                                            device="cpu", randomize_class=True)
        for _ in zip(range(3), gen):
            pass
        return seen

    full = run(None, 4)
    for rows in ([0], [2, 3]):
        part = run((rows, 4), len(rows))
        assert th.equal(part[0][0], full[0][0][rows])                       # x_T
        for k in range(3):
            assert th.equal(part[k][1], full[k][1][rows]), k                  # per-step noise
            assert th.equal(part[k][2], full[k][2][rows]), k                  # class ids


def test_prompt_weight_rows_follow_the_global_batch():
    """B == P > 1 scores sample b against prompt b only (cgd.py:196-200): a rank that owns sample 2 of a global batch of 4 must use
    row 2 of the 4 x 4 weight matrix although its local batch is 1."""
    sys.path.insert(0, ROOT)
    import cgd_amd  # noqa: F401
    from cgd_amd import guidance as dg
    w = th.tensor([0.4, 0.3, 0.2, 0.1])
    full = dg.prompt_weight_matrix(w, 4, "cpu")
    assert th.equal(full, th.eye(4) * w.sum())
    assert th.equal(full[[2]], th.tensor([[0.0, 0.0, 1.0, 0.0]]) * w.sum())
    assert th.equal(dg.prompt_weight_matrix(w, 1, "cpu"), w.view(1, 4))     # what a naive per-rank B = 1 would have used instead


def _download_worker(rank, world, port, cache, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import requests
    import cgd_amd  # noqa: F401
    from cgd import script_util

    class Resp:
        headers = {"Content-Length": str(4 << 20)}

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def raise_for_status(self):
            pass

        def iter_content(self, chunk_size=0):
            for i in range(64):  # 4 MiB in 64 slow chunks: a concurrent writer on the same file would interleave or truncate
                time.sleep(0.005)
                yield bytes([i]) * (64 << 10)

    fetched = []

    def fake_get(url, **kw):
        fetched.append(url)
        return Resp()

    requests.get = fake_get
    if rank == 1:
        time.sleep(0.3)  # a rank that arrives late must still take part (and not re-download)
    path = script_util.download("http://example.invalid/ckpt.pt", "ckpt.pt", root=cache)
    data = open(path, "rb").read()
    ok = len(data) == 4 << 20 and all(data[i * (64 << 10)] == i and data[(i + 1) * (64 << 10) - 1] == i for i in range(64))
    # second call: cache hit decided by rank 0 for everybody, no fetch anywhere
    path2 = script_util.download("http://example.invalid/ckpt.pt", "ckpt.pt", root=cache)
    # failure on rank 0 surfaces on every rank instead of leaving the others in a collective
    def boom(url, **kw):
        raise requests.exceptions.ConnectionError("no route")
    requests.get = boom
    os.environ["CGD_DOWNLOAD_BACKOFF"] = "0"
    try:
        script_util.download("http://example.invalid/other.pt", "other.pt", root=cache)
        failed = None
    except RuntimeError as e:
        failed = str(e)
    q.put((rank, len(fetched), ok, path == path2, failed, sorted(os.listdir(cache))))
    dist.barrier()
    dist.destroy_process_group()


def test_checkpoint_download_runs_on_rank0_only_world2(tmp_path):
    """under the multi-GPU launcher every rank used to stream into the same `<target>.tmp`.  Now rank 0 downloads
    (per-process temporary name, atomic os.replace), the other ranks wait and read the finished file; a failure on rank 0 is raised
    on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    cache = str(tmp_path / "cache")
    procs = [ctx.Process(target=_download_worker, args=(r, 2, port, cache, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, n0, ok0, same0, f0, ls0), (r1, n1, ok1, same1, f1, ls1) = res
    assert (n0, n1) == (1, 0)            # one fetch in total, on rank 0
    assert ok0 and ok1 and same0 and same1
    assert f0 and "Download failed" in f0 and f1 and "rank 0 failed" in f1
    assert ls0 == ["ckpt.pt"] == ls1     # no temporary files left behind


def test_forced_collectives_on_a_one_rank_group():
    """CGD_FORCE_COLLECTIVES=1 (the knob behind the -m gpu RCCL test): an initialised group of ONE rank still takes the broadcast path
    of shard.broadcast_flat / on_rank0 / load_broadcast — here over gloo."""
    import subprocess
    code = (
        "import os, sys, torch as th, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import cgd_amd\n"
        "from cgd_amd import shard\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        "calls = []\n"
        "orig = dist.broadcast\n"
        "dist.broadcast = lambda t, src=0, **kw: (calls.append(t.numel()), orig(t, src=src, **kw))[1]\n"
        "flat = shard.broadcast_flat(lambda: th.arange(10.), 10, 'cpu')\n"
        "assert shard.on_rank0(lambda: 7) == 7\n"
        "class Net:\n"
        "    def param_specs(self): return [('a', 4), ('b', 6)]\n"
        "    def load_state_dict(self, sd, prefix=''): self.sd = sd; return self\n"
        "n = shard.load_broadcast(Net(), lambda: {'a': th.ones(4), 'b': th.zeros(2, 3)}, 'cpu')\n"
        "assert n.sd['a'].sum() == 4 and n.sd['b'].numel() == 6\n"
        "print('broadcasts', calls)\n"
        "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for force, expect in (("1", "broadcasts [10, 10]"), ("0", "broadcasts []")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(env, CGD_FORCE_COLLECTIVES=force), capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert expect in out.stdout, out.stdout

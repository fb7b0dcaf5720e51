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


def test_rank_samples_partition():
    import cgd_amd  # noqa: F401
    from cgd_amd import shard
    for B, n in [(8, 8), (4, 8), (7, 3), (1, 1)]:
        parts = [shard.rank_samples(B, r, n) for r in range(n)]
        assert sorted(sum(parts, [])) == list(range(B))

"""Multi-GPU launcher of the drop-in generator: one process per GPU, samples sharded one (or a contiguous block) per rank.

    python -m cgd_amd.launch --gpus 8 -- --prompts "a photon" --image_size 256 --batch_size 8 ...   (the `cgd` CLI flags)

or, from Python, `cgd_amd.launch.run(n_gpus, **clip_guided_diffusion_kwargs)` -> [(batch_idx, png_path), ...].

The reference has only the batch dimension (/root/reference/cgd/cgd.py:250-252); here `batch_size = B` on a node with N GPUs
runs ceil(B/N) samples per GPU: every rank draws the GLOBAL random tensors (x_T, per-step noise, class ids,
cutout boxes) from the same seed and keeps its rows, so the frames are those of the single-process batched run; weights are
read once by rank 0 and broadcast over RCCL/xGMI (`shard.load_broadcast`); there is NO per-step collective; each rank writes
its own PNGs under `prefix/<prompts>/<batch_idx:02>/` (/root/reference/cgd/script_util.py:87-101).  Documented deviation: the
magnitude clamp and the saturation mean reduce over the samples of ONE rank, not over the global batch (cgd.py:215,230).

CGD_LAUNCH_DEVICE / CGD_LAUNCH_BACKEND are test knobs (every rank on one device over gloo, for 1-GPU boxes).
"""
import json
import os
import socket
import subprocess
import sys


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(n_gpus, worker_args, env_extra=None):
    """Start `n_gpus` ranks of this module's worker (RANK = LOCAL_RANK = 0..n-1, rendezvous on 127.0.0.1) and wait for them."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n_gpus), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the repo root holds the `cgd_amd` / `cgd` import shims
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, "-m", "cgd_amd.launch", "--worker"] + list(worker_args),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n_gpus)]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise RuntimeError(f"cgd_amd.launch: ranks exited with {rcs}")


def _init_rank():
    import torch as th
    import torch.distributed as dist
    local = int(os.environ.get("CGD_LAUNCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    th.cuda.set_device(local)
    backend = os.environ.get("CGD_LAUNCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=th.device(f"cuda:{local}"))
    else:
        dist.init_process_group(backend)
    return local, dist


def _worker(argv):
    """One rank: the unmodified `cgd` CLI (or a kwargs file from `run`) on its own GPU; sharding happens inside the generator."""
    local, dist = _init_rank()
    from cgd import cgd as entry
    try:
        if argv and argv[0] == "--kwargs-json":
            with open(argv[1]) as f:
                spec = json.load(f)
            kw = dict(spec["kwargs"], device=f"cuda:{local}")
            items = [(b, p) for b, p in entry.clip_guided_diffusion(**kw)]
            with open(f"{spec['out']}.{dist.get_rank()}", "w") as f:
                json.dump(items, f)
        else:
            sys.argv = ["cgd"] + [a for a in argv] + ["--device", f"cuda:{local}"]
            entry.main()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run(n_gpus, **kwargs):
    """`clip_guided_diffusion(**kwargs)` sharded over `n_gpus` processes; returns every rank's `(batch_idx, png_path)` items in
    the single-process order (step-major, then batch index).  Frames are on disk when this returns."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        spec = os.path.join(d, "spec.json")
        out = os.path.join(d, "items.json")
        with open(spec, "w") as f:
            json.dump({"kwargs": {k: (str(v) if isinstance(v, os.PathLike) else v) for k, v in kwargs.items()}, "out": out}, f)
        spawn(n_gpus, ["--kwargs-json", spec])
        items = []
        for r in range(n_gpus):
            with open(f"{out}.{r}") as f:
                items += [tuple(x) for x in json.load(f)]
    return sorted(items, key=lambda it: (os.path.basename(it[1]), it[0]))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "--worker":
        return _worker(argv[1:])
    if len(argv) < 2 or argv[0] != "--gpus":
        raise SystemExit("usage: python -m cgd_amd.launch --gpus N -- <cgd command-line flags>")
    n = int(argv[1])
    rest = argv[2:]
    if rest and rest[0] == "--":
        rest = rest[1:]
    spawn(n, rest)


if __name__ == "__main__":
    main()

"""Sample-parallel sharding: independent samples, one per GPU, no per-step collective.

Shared across ranks: packed weights (ONE broadcast over RCCL/xGMI at start-up), prompt embeddings, cutout coordinates
(the reference uses the same crop boxes for every batch element, /root/reference/cgd/modules.py:60-61), class-randomisation
draws and schedule tables.  Per rank: slice `b` of x_T and of each step's noise (the reference draws one (B,3,H,W) tensor;
rank b takes [b]).  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import os

import torch as th
import torch.distributed as dist


def force_collectives():
    """Test knob (tests/test_gpu_step.py::test_rccl_single_rank_collectives): with CGD_FORCE_COLLECTIVES=1 an initialised process group
    of ONE rank still takes the N > 1 code path — the flat-vector broadcast, the object broadcast — so that RCCL executes it on a
    1-GPU box."""
    return os.environ.get("CGD_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_flat(make_flat, numel, device, src=0, dtype=th.float32):
    """Rank `src` materialises the flat parameter vector (make_flat()); everyone else receives it in one broadcast."""
    rank, n = world()
    flat = make_flat().to(device=device, dtype=dtype).contiguous() if rank == src else th.empty(numel, device=device, dtype=dtype)
    if flat.numel() != numel:
        raise ValueError(f"flat parameter vector has {flat.numel()} elements, expected {numel}")
    if n > 1 or force_collectives():
        dist.broadcast(flat, src=src)
    return flat


def on_rank0(fn):
    """Run `fn()` on rank 0 only and hand its (picklable, small) result to every rank; the other ranks wait, so whatever rank 0 wrote to a
    shared cache directory is complete when they carry on.  An exception on rank 0 is raised on EVERY rank (no rank is left waiting in a
    collective).  Single-process: a plain call.  Used for checkpoint downloads and for probing a checkpoint's architecture: N ranks
    streaming into one cache file, or N ranks each un-pickling a multi-GB archive, is the failure mode this removes."""
    rank, n = world()
    if n == 1 and not force_collectives():
        return fn()
    result, err = None, None
    if rank == 0:
        try:
            result = fn()
        except Exception as e:  # noqa: BLE001 - re-raised below, on every rank
            err = e
    box = [(result, None if err is None else f"{type(err).__name__}: {err}")]
    dist.broadcast_object_list(box, src=0)
    result, msg = box[0]
    if msg is not None:
        if err is not None:
            raise err
        raise RuntimeError(f"rank 0 failed: {msg}")
    return result


def flat_pack(sd, names):
    return th.cat([sd[n].reshape(-1).float() for n in names])


def flat_unpack(flat, specs):
    out, off = {}, 0
    for name, numel in specs:
        out[name] = flat[off:off + numel]
        off += numel
    return out


def load_broadcast(net, make_state_dict, device, prefix=""):
    """Load a network handle (nets._Net) on every rank from ONE materialisation of its state dict: rank 0 calls
    `make_state_dict()` (reads the checkpoint / draws the synthetic weights), packs the parameters the handle asks for into one
    flat fp32 vector and broadcasts it over RCCL/xGMI; the other ranks never touch the disk.  Single-process: a plain load."""
    rank, n = world()
    if n == 1 and not force_collectives():
        return net.load_state_dict(make_state_dict(), prefix)
    specs = net.param_specs()
    names = [prefix + name for name, _ in specs]
    flat = broadcast_flat(lambda: flat_pack(make_state_dict(), names), sum(k for _, k in specs), device)
    return net.load_state_dict(flat_unpack(flat, [(prefix + name, k) for name, k in specs]), prefix)


def rank_samples(global_batch, rank=None, nranks=None):
    """Indices of the global batch owned by a rank (contiguous blocks; 1 sample per GPU when global_batch == world)."""
    r, n = world()
    rank = r if rank is None else rank
    nranks = n if nranks is None else nranks
    per, extra = divmod(global_batch, nranks)
    start = rank * per + min(rank, extra)
    return list(range(start, start + per + (1 if rank < extra else 0)))


def slice_tape(tape, idx):
    """Per-rank view of a global RNG tape {'x_T': (B,..), 'noise': [(B,..)], 'y': [(B,)], 'coords': [...]}."""
    out = {"x_T": tape["x_T"][idx], "noise": [n[idx] for n in tape["noise"]], "y": [y[idx] for y in tape["y"]]}
    if "coords" in tape:
        out["coords"] = tape["coords"]  # shared: same crop boxes for every sample
    return out


def gather_images(img, dst=0):
    """Optional end-of-run gather of the finished (b,3,H,W) images to rank `dst` (each rank may also write its own PNGs)."""
    rank, n = world()
    if n == 1:
        return [img]
    bufs = [th.empty_like(img) for _ in range(n)] if rank == dst else None
    dist.gather(img, bufs, dst=dst)
    return bufs

"""Plugin surface `cgd.script_util` (reference: /root/reference/cgd/script_util.py).

The hot-path boundary here is `load_guided_diffusion` (reference :281-324): it returns the MI355X UNet handle and
the guided sampler instead of guided_diffusion's (UNetModel, SpacedDiffusion).  The I/O helpers the generator needs
(prompt parsing, image logging, cached download) keep the reference's behaviour; ffmpeg post-processing and the
network retry machinery are outside the hot-path scope (control plane) and kept minimal.
"""
import io
import os
import time
import re
from functools import lru_cache
from pathlib import Path

import torch as th

from cgd_amd import diffusion as _diffusion
from cgd_amd import lib as _lib
from cgd_amd import nets as _nets
from cgd_amd import shard as _shard
from cgd_amd import sampler as _sampler
from cgd_amd import synthetic as _synthetic

from .model_flags import DIFFUSION_LOOKUP, MODEL_AND_DIFFUSION_DEFAULTS

CACHE_PATH = os.path.expanduser("~/.cache/clip-guided-diffusion")
TIMESTEP_RESPACINGS = tuple(p + n for p in ("", "ddim") for n in ("25", "50", "100", "250", "500", "1000"))
DIFFUSION_SCHEDULES = (25, 50, 100, 250, 500, 1000)
IMAGE_SIZES = (64, 128, 256, 512)


def synthetic_weights_enabled():
    """CGD_SYNTHETIC_WEIGHTS=1: seeded random weights instead of checkpoints (bench / CI boxes have no network)."""
    return os.environ.get("CGD_SYNTHETIC_WEIGHTS", "0") not in ("", "0")


def parse_prompt(prompt):
    """'<text or url>[:<weight>]' -> (text, weight); URLs keep their scheme colon."""
    is_url = prompt.startswith(("http://", "https://"))
    parts = prompt.rsplit(":", 2 if is_url else 1)
    if is_url:
        parts = [parts[0] + ":" + parts[1]] + parts[2:]
    text = parts[0]
    weight = float(parts[1]) if len(parts) > 1 else 1.0
    return text, weight


def fetch(url_or_path):
    if str(url_or_path).startswith(("http://", "https://")):
        import requests
        r = requests.get(url_or_path)
        r.raise_for_status()
        return io.BytesIO(r.content)
    return open(url_or_path, "rb")


def alphanumeric_filter(s: str) -> str:
    return re.sub(r"[^\w\s]", "", s).replace(" ", "_")


def clean_and_combine_prompts(base_path, txts, batch_idx, max_length=255) -> str:
    stem = "_".join(alphanumeric_filter(t) for t in txts)[:max_length]
    return os.path.join(base_path, stem, f"{batch_idx:02}")


def to_uint8_hwc(image: th.Tensor) -> th.Tensor:
    """(...,3,H,W) in [-1,1] -> (...,H,W,3) uint8, on the tensor's device (reference: TF.to_pil_image(image.add(1).div(2).clamp(0, 1)))."""
    return image.detach().float().add(1).div(2).clamp(0, 1).mul(255).byte().movedim(-3, -1).contiguous()


def stage_images(pred_xstart: th.Tensor):
    """Starts the uint8 conversion and an asynchronous device->host copy of a (B,3,H,W) batch; `.get()` of the returned
    handle yields the (B,H,W,3) uint8 host tensor without waiting for GPU work enqueued after this call."""
    from cgd_amd.hostcopy import HostCopy
    return HostCopy(to_uint8_hwc(pred_xstart))


def write_image(arr, base_path: str, txts: list, current_step: int, batch_idx: int) -> str:
    """(H,W,3) uint8 array -> '<base>/<prompts>/<batch:02>/<step:04>.png' (+ ./current.png), returns the path."""
    from PIL import Image
    dirname = clean_and_combine_prompts(base_path, txts, batch_idx)
    os.makedirs(dirname, exist_ok=True)
    filename = os.path.join(dirname, f"{current_step:04}.png")
    pil_image = Image.fromarray(arr)
    pil_image.save(os.path.join(os.getcwd(), "current.png"))
    pil_image.save(filename)
    return str(filename)


def log_image(image: th.Tensor, base_path: str, txts: list, current_step: int, batch_idx: int) -> str:
    """(3,H,W) in [-1,1] -> PNG as above (reference signature, /root/reference/cgd/script_util.py:93-101)."""
    return write_image(to_uint8_hwc(image).cpu().numpy(), base_path, txts, current_step, batch_idx)


def download(url: str, filename: str, root: str = CACHE_PATH, max_retries: int = 3) -> str:
    """Cached download: returns the target immediately when it exists; otherwise streams to a per-process temporary file and moves it
    into place atomically.  In a multi-GPU run (cgd_amd.launch: one process per GPU) only rank 0 downloads; the other ranks wait for it
    and then find the finished file in the cache."""
    return _shard.on_rank0(lambda: _download(url, filename, root, max_retries))


def _download(url: str, filename: str, root: str, max_retries: int) -> str:
    os.makedirs(root, exist_ok=True)
    target = Path(root) / filename
    if target.exists() and not target.is_file():
        raise RuntimeError(f"{target} exists and is not a regular file")
    if target.is_file():
        return str(target)
    import requests
    tmp = target.with_name(f"{target.name}.tmp.{os.getpid()}")  # never shared between processes (two launches on one cache directory)
    last = None
    for attempt in range(max_retries):
        try:
            with requests.get(url, stream=True, timeout=(30, 120)) as resp:
                resp.raise_for_status()
                expected = int(resp.headers.get("Content-Length", 0) or 0)
                with open(tmp, "wb") as out:
                    for chunk in resp.iter_content(chunk_size=1 << 16):
                        out.write(chunk)
                    out.flush()
                    os.fsync(out.fileno())
            got = tmp.stat().st_size
            if expected and got != expected:  # a connection that closed early without raising must not reach the cache
                raise OSError(f"download incomplete: expected {expected} bytes, got {got}")
            os.replace(tmp, target)  # atomic: a reader sees the old state or the complete file
            return str(target)
        except (requests.exceptions.RequestException, OSError) as e:
            last = e
            if tmp.exists():
                tmp.unlink()
            if attempt < max_retries - 1:
                time.sleep(float(os.environ.get("CGD_DOWNLOAD_BACKOFF", "1")) * 2 ** attempt)
    raise RuntimeError(f"Download failed after {max_retries} attempts: {last}") from last


def download_guided_diffusion(image_size: int, class_cond: bool, checkpoints_dir: str = CACHE_PATH, overwrite: bool = False) -> str:
    info = DIFFUSION_LOOKUP["cond" if class_cond else "uncond"][image_size]
    target = Path(checkpoints_dir) / info["filename"]
    if synthetic_weights_enabled():
        return str(target)
    # the cache check is part of what rank 0 decides for everybody: a rank that arrived late and found the freshly downloaded file must
    # not skip the collective the others are waiting in
    return _shard.on_rank0(lambda: str(target) if (not overwrite and target.exists()) else
                           _download(info["url"], info["filename"], checkpoints_dir, 3))


_CTX = {}


def get_context(device):
    """One library context per GPU ('cuda', 'cuda:N')."""
    dev = th.device(device)
    if dev.type != "cuda":
        raise ValueError(f"the MI355X path needs a GPU device string ('cuda[:N]'), got {device!r}; there is no CPU fallback")
    idx = dev.index if dev.index is not None else th.cuda.current_device()
    if idx not in _CTX:
        _CTX[idx] = _lib.Context(idx, os.environ.get("CGD_PRECISION", "bf16x3"))
    return _CTX[idx]


def model_config(image_size, class_cond, diffusion_steps=None, timestep_respacing=None, use_fp16=True, noise_schedule="linear",
                 dropout=0.0):
    """Per-checkpoint flags over upstream defaults, then the user-level overrides (reference :305-315)."""
    cfg = dict(MODEL_AND_DIFFUSION_DEFAULTS)
    cfg.update(DIFFUSION_LOOKUP["cond" if class_cond else "uncond"][image_size]["model_flags"])
    cfg.update(diffusion_steps=diffusion_steps, timestep_respacing=timestep_respacing, use_fp16=use_fp16, noise_schedule=noise_schedule,
               dropout=dropout)
    return cfg


def unet_kwargs(cfg):
    """guided_diffusion's create_model flags -> the device UNet's constructor arguments (NUM_CLASSES = 1000 upstream; the channel
    multipliers default per image size like upstream's create_model)."""
    return dict(image_size=cfg["image_size"], model_channels=cfg["num_channels"], num_res_blocks=cfg["num_res_blocks"],
                attention_resolutions=cfg["attention_resolutions"].replace(" ", ""), channel_mult=None,
                num_classes=1000 if cfg["class_cond"] else None, num_heads=cfg["num_heads"], num_head_channels=cfg["num_head_channels"],
                use_new_attention_order=cfg["use_new_attention_order"], out_channels=6 if cfg["learn_sigma"] else 3)


@lru_cache(maxsize=1)
def load_guided_diffusion(checkpoint_path: str, image_size: int, class_cond: bool, diffusion_steps: int = None,
                          timestep_respacing: str = None, use_fp16: bool = True, device: str = "", noise_schedule: str = "linear",
                          dropout: float = 0.0):
    """-> (model, diffusion): the device UNet handle (`model(x, ts, y)`, `.dgrad`, `.num_classes`) and the guided sampler
    (`.num_timesteps`, `.sqrt_one_minus_alphas_cumprod`, `.p_sample_loop_progressive`, `.ddim_sample_loop_progressive`)."""
    if not (len(device) > 0):
        raise ValueError("device must be set")
    if noise_schedule not in ("linear", "cosine"):
        raise ValueError("linear_or_cosine must be set")
    cfg = model_config(image_size, class_cond, diffusion_steps, timestep_respacing, use_fp16, noise_schedule, dropout)
    ctx = get_context(device)
    model = _nets.UNet(ctx, **unet_kwargs(cfg))
    # multi-GPU runs (cgd_amd.launch): rank 0 reads / draws the weights, one RCCL broadcast hands them to the other ranks
    dev = f"cuda:{ctx.device}"
    if os.path.isfile(checkpoint_path):
        _shard.load_broadcast(model, lambda: th.load(checkpoint_path, map_location="cpu"), dev)
    elif synthetic_weights_enabled():
        _shard.load_broadcast(model, lambda: _synthetic.synthetic_state_dict(model, seed=1234, device=dev), dev)
    else:
        raise FileNotFoundError(f"{checkpoint_path} not found (set CGD_SYNTHETIC_WEIGHTS=1 for seeded random weights)")
    steps = cfg["diffusion_steps"]
    tables = _diffusion.create_gaussian_diffusion(steps=steps, noise_schedule=cfg["noise_schedule"],
                                                  timestep_respacing=cfg["timestep_respacing"] or "",
                                                  rescale_timesteps=cfg["rescale_timesteps"])
    return model, _sampler.GuidedSampler(ctx, tables)


def load_lpips(ctx, checkpoints_dir: str = CACHE_PATH, device="cuda"):
    """Device LPIPS-VGG16 (reference: `lpips.LPIPS(net='vgg')`, cgd.py:147-148).  Weights: `<checkpoints_dir>/lpips_vgg16.pt`, a
    state dict with the package's keys (torchvision VGG16 trunk as `net.slice{k}.{idx}.weight|bias`, heads `lin{k}.model.1.weight`);
    with CGD_SYNTHETIC_WEIGHTS=1 seeded synthetic weights are used (no network on the build / bench boxes)."""
    from cgd_amd import nets, synthetic
    net = nets.LpipsVGG(ctx)
    path = os.path.join(checkpoints_dir, "lpips_vgg16.pt")
    if os.path.isfile(path):
        sd = th.load(path, map_location="cpu")
        sd = {k: (v.reshape(-1) if k.startswith("lin") else v) for k, v in sd.items()}
        return net.load_state_dict({k: v.to(device) for k, v in sd.items() if k.startswith(("net.slice", "lin"))})
    if synthetic_weights_enabled():
        return net.load_state_dict(synthetic.lpips_state_dict(device=device))
    raise FileNotFoundError(f"{path} not found: export lpips.LPIPS(net='vgg').state_dict() there (or set CGD_SYNTHETIC_WEIGHTS=1)")

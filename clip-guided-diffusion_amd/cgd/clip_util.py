"""Plugin surface `cgd.clip_util` (reference: /root/reference/cgd/clip_util.py).

Hot path: `CLIP_NORMALIZE` (:45, fused into the cutout kernel on the native path) and `load_clip(...)` whose
`.encode_image` / `.visual.input_resolution` (:59-66) are served by the MI355X image tower.  Prompt encoding
(`encode_text_prompt`, :104-108) is one-off setup that stays in PyTorch through the optional `clip` package
(out of scope for kernels); without it only synthetic-weight runs are possible.
"""
import hashlib
import os
from functools import lru_cache

import torch as th

from cgd_amd import nets as _nets
from cgd_amd import shard as _shard
from cgd_amd import synthetic as _synthetic
from cgd_amd.guidance import CLIP_MEAN, CLIP_STD, MakeCutouts  # noqa: F401

from . import script_util

CLIP_MODEL_NAMES = ("ViT-B/16", "ViT-B/32", "RN50", "RN101", "RN50x4", "RN50x16", "ViT-L/14")
_AZURE = "https://openaipublic.azureedge.net/clip/models/"
CLIP_MODEL_URLS = {
    "RN50": _AZURE + "afeb0e10f9e5a86da6080e35cf09123aca3b358a0c3e3b6c78a7b63bc04b6762/RN50.pt",
    "RN101": _AZURE + "8fa8567bab74a42d41c5915025a8e4538c3bdbe8804a470a72f30b0d94fab599/RN101.pt",
    "RN50x4": _AZURE + "7e526bd135e493cef0776de27d5f42653e6b4c8bf9e0f653bb11773263205fdd/RN50x4.pt",
    "RN50x16": _AZURE + "52378b407f34354e150460fe41077663dd5b39c54cd0bfd2b27167a4a06ec9aa/RN50x16.pt",
    "ViT-B/32": _AZURE + "40d365715913c9da98579312b702a82c18be219cc2a73407c4526f58eba950af/ViT-B-32.pt",
    "ViT-B/16": _AZURE + "5806e77cd80f8b59890b7e101eabd078d9fb84e6937f9e85e4ecb61988df416f/ViT-B-16.pt",
    "ViT-L/14": _AZURE + "b8cca3fd41ae0c99ba7e8951adf17d267cdb84cd88be6f7c2e0eca1737a03836/ViT-L-14.pt",
    "ViT-L/14@336px": _AZURE + "3035c92b350959924f9f00213499208652fc7ea050643e8b385c2dac08641f02/ViT-L-14-336px.pt",
}


class _Normalize:
    """torchvision.transforms.Normalize(mean, std) for (...,3,H,W) tensors."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, x):
        m = th.tensor(self.mean, dtype=x.dtype, device=x.device).view(3, 1, 1)
        s = th.tensor(self.std, dtype=x.dtype, device=x.device).view(3, 1, 1)
        return (x - m) / s


CLIP_NORMALIZE = _Normalize(CLIP_MEAN, CLIP_STD)


def download_clip_model(model_name: str) -> str:
    if model_name not in CLIP_MODEL_URLS:
        raise ValueError(f"Unknown CLIP model: {model_name}. Available: {list(CLIP_MODEL_URLS.keys())}")
    filename = model_name.replace("/", "-") + ".pt"
    cache_dir = os.path.join(script_util.CACHE_PATH, "clip")
    if script_util.synthetic_weights_enabled():
        return os.path.join(cache_dir, filename)
    return script_util.download(CLIP_MODEL_URLS[model_name], filename, root=cache_dir)


class _Visual:
    def __init__(self, tower):
        self.input_resolution = tower.input_resolution
        self.output_dim = tower.out_dim
        self.tower = tower


_EncodeImageFunction = _nets.EncodeImageFunction  # autograd node over cgd_*_forward / cgd_*_dgrad


class ClipModel:
    """The slice of clip.model.CLIP the generator touches: `.visual.input_resolution`, `.encode_image`, and (when the
    `clip` package + checkpoint are present) `.encode_text`."""

    def __init__(self, tower, text_model=None, name="ViT-B/32"):
        self.visual = _Visual(tower)
        self.tower = tower
        self.text_model = text_model
        self.name = name

    def encode_image(self, image):
        """(N,3,res,res) CLIP-normalised -> (N,D).  Differentiable w.r.t. `image` (autograd node over cgd_*_forward / _dgrad) so
        that a user-supplied cond_fn written like the reference's (cgd.py:190-228) works unchanged."""
        if image.requires_grad and th.is_grad_enabled():
            return _EncodeImageFunction.apply(image, self.tower)
        return self.tower.encode_image(image)

    def encode_text(self, tokens):
        if self.text_model is None:
            raise RuntimeError("text encoding needs the `clip` package and a CLIP checkpoint (setup-time, stays in PyTorch)")
        return self.text_model.encode_text(tokens)

    def eval(self):
        return self


def _vit_config_from_state_dict(sd):
    """clip.model.build_model's shape inference for the visual tower."""
    width = sd["visual.conv1.weight"].shape[0]
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    return (patch * grid, patch, width, layers, width // 64, sd["visual.proj"].shape[1])


def _rn_config_from_state_dict(sd):
    """clip.model.build_model's shape inference for a ModifiedResNet tower."""
    counts = [len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{b}")}) for b in (1, 2, 3, 4)]
    width = sd["visual.layer1.0.conv1.weight"].shape[0]
    grid = round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
    out_dim = sd["visual.attnpool.c_proj.weight"].shape[0]
    return (grid * 32, width, tuple(counts), out_dim, width * 32 // 64)


@lru_cache(maxsize=4)  # the reference caches one model; "A+B" multi-CLIP runs keep several
def load_clip(model_name="ViT-B/32", device="cpu"):
    print(f"Loading clip model\t{model_name}\ton device\t{device}.")
    if device == "cpu" or "cuda" not in device:
        raise ValueError("Invalid or unspecified device: {} (the MI355X path needs 'cuda[:N]'; no CPU fallback)".format(device))
    ctx = script_util.get_context(device)
    model_path = download_clip_model(model_name) if model_name in CLIP_MODEL_URLS else model_name
    if os.path.isfile(model_path):
        held = {}

        def probe():  # rank 0 only (multi-GPU runs): un-pickle the archive once, infer the tower like clip.model.build_model
            try:
                sd = th.jit.load(model_path, map_location="cpu").state_dict()
            except RuntimeError:
                sd = th.load(model_path, map_location="cpu")
            held["sd"] = sd
            return ("vit", _vit_config_from_state_dict(sd)) if "visual.proj" in sd else ("rn", _rn_config_from_state_dict(sd))

        kind, cfg = _shard.on_rank0(probe)  # the other ranks receive the configuration, never the file
        tower = _nets.ClipImageTower(ctx, config=cfg) if kind == "vit" else _nets.ClipResNetTower(ctx, config=cfg)
        _shard.load_broadcast(tower, lambda: {k: v.float() for k, v in held["sd"].items() if k.startswith("visual.")},
                              f"cuda:{ctx.device}", prefix="visual.")
        held.clear()
        text_model = None
        try:
            import clip  # optional, setup-time only
            text_model = clip.load(model_path, jit=False, device=device)[0].eval().requires_grad_(False)
        except ImportError:
            pass
        return ClipModel(tower, text_model, model_name), tower.input_resolution
    if script_util.synthetic_weights_enabled():
        if model_name in _nets.VIT_CONFIGS:
            tower = _nets.ClipImageTower(ctx, model_name)
            _shard.load_broadcast(tower, lambda: _synthetic.synthetic_state_dict(tower, seed=4321, device=f"cuda:{ctx.device}"),
                                  f"cuda:{ctx.device}")
        elif model_name in _nets.RN_CONFIGS:
            tower = _nets.ClipResNetTower(ctx, model_name)
            _shard.load_broadcast(tower, lambda: _synthetic.resnet_state_dict(tower, seed=2468, device=f"cuda:{ctx.device}"),
                                  f"cuda:{ctx.device}")
        else:
            raise NotImplementedError(f"{model_name}: supported towers are {sorted(_nets.VIT_CONFIGS) + sorted(_nets.RN_CONFIGS)}")
        return ClipModel(tower, None, model_name), tower.input_resolution
    raise FileNotFoundError(f"{model_path} not found (set CGD_SYNTHETIC_WEIGHTS=1 for seeded random weights)")


def _synthetic_text_embedding(txt, dim, device):
    seed = int.from_bytes(hashlib.sha256(txt.encode()).digest()[:8], "little") % (2 ** 31)
    return th.randn(1, dim, generator=th.Generator().manual_seed(seed)).to(device)


def encode_text_prompt(txt, weight, clip_model_name="ViT-B/32", device="cpu"):
    clip_model, _ = load_clip(clip_model_name, device)
    if clip_model.text_model is None:
        if script_util.synthetic_weights_enabled():
            return _synthetic_text_embedding(txt, clip_model.visual.output_dim, device), weight
        raise RuntimeError("encode_text_prompt needs the `clip` package (tokenizer + text tower)")
    import clip
    tokens = clip.tokenize(txt).to(device)
    return clip_model.encode_text(tokens).float(), weight


def encode_image_prompt(image: str, weight: float, diffusion_size: int, num_cutouts, clip_model_name: str = "ViT-B/32",
                        device: str = "cpu"):
    """Image prompt -> (cutn, D) embeddings with weight/cutn each (reference :90-101).  The reference resizes with the
    vendored ResizeRight lanczos3; here PIL's LANCZOS does the one-off resize (setup-time, outside the hot path)."""
    import numpy as np
    from PIL import Image
    clip_model, clip_size = load_clip(clip_model_name, device)
    make_cutouts = MakeCutouts(cut_size=clip_size, num_cutouts=num_cutouts, ctx=clip_model.tower.ctx)
    pil_img = Image.open(script_util.fetch(image)).convert("RGB")
    smallest = min(diffusion_size, *pil_img.size)
    scale = smallest / min(pil_img.size)
    pil_img = pil_img.resize((max(1, round(pil_img.size[0] * scale)), max(1, round(pil_img.size[1] * scale))), Image.LANCZOS)
    img = th.from_numpy(np.asarray(pil_img)).float().div(255).permute(2, 0, 1).unsqueeze(0).to(device)
    batch = make_cutouts(img)
    # quirk kept: the reference's `tf` is torch.nn.functional, so `tf.normalize(batch)` (clip_util.py:99) L2-normalises along the
    # channel axis instead of applying CLIP_NORMALIZE
    batch_embed = clip_model.encode_image(th.nn.functional.normalize(batch)).float()
    return batch_embed, [weight / make_cutouts.cutn] * make_cutouts.cutn

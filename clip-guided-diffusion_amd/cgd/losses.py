"""Plugin surface `cgd.losses` (reference: /root/reference/cgd/losses.py:5-22).

The native sampler never calls these: csrc/guidance.hip evaluates the three losses together with their gradients.
They are kept, as differentiable PyTorch functions on whatever device the tensors live on, for user-supplied
cond_fns that follow the reference's recipe.
"""
import torch as th
import torch.nn.functional as F


def range_loss(input):
    """mean over (c,h,w) of the squared excursion outside [-1, 1], one value per sample."""
    excess = input - input.clamp(-1, 1)
    return excess.pow(2).mean([1, 2, 3])


def spherical_dist_loss(x: th.Tensor, y: th.Tensor):
    """2 * asin(||x^ - y^|| / 2)^2 on L2-normalised embeddings (broadcasts like the reference)."""
    x, y = F.normalize(x, dim=-1), F.normalize(y, dim=-1)
    half_chord = (x - y).norm(dim=-1).div(2)
    return half_chord.arcsin().pow(2).mul(2)


def tv_loss(input: th.Tensor):
    """L2 total variation with replicate padding on the right/bottom edge, one value per sample."""
    padded = F.pad(input, (0, 1, 0, 1), "replicate")
    core = padded[..., :-1, :-1]
    dx = padded[..., :-1, 1:] - core
    dy = padded[..., 1:, :-1] - core
    return (dx ** 2 + dy ** 2).mean([1, 2, 3])

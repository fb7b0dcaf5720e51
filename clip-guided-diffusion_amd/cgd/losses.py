"""Plugin surface `cgd.losses` (reference: /root/reference/cgd/losses.py:5-22).

The native sampler never calls these: csrc/guidance.hip evaluates the three losses together with their gradients.
They are kept, as differentiable PyTorch functions on whatever device the tensors live on, for user-supplied
cond_fns that follow the reference's recipe (same values and gradients, written in closed form).
"""
import torch as th
import torch.nn.functional as F


def range_loss(input):
    """Mean squared excursion outside [-1, 1] per sample: (x - clamp(x, -1, 1))^2 = relu(|x| - 1)^2."""
    return th.relu(input.abs() - 1).square().mean(dim=(1, 2, 3))


def spherical_dist_loss(x: th.Tensor, y: th.Tensor):
    """Squared great-circle distance of the L2-normalised embeddings, 2 * asin(chord / 2)^2 (broadcasts like the reference)."""
    chord = th.linalg.vector_norm(F.normalize(x, dim=-1) - F.normalize(y, dim=-1), dim=-1)
    return 2 * th.asin(chord / 2) ** 2


def tv_loss(input: th.Tensor):
    """L2 total variation per sample.  With replicate padding on the right / bottom edge the last column of the horizontal
    differences and the last row of the vertical ones are zero, so the mean over (C, H, W) is the two difference energies / (C*H*W)."""
    energy = th.diff(input, dim=-1).square().sum(dim=(1, 2, 3)) + th.diff(input, dim=-2).square().sum(dim=(1, 2, 3))
    return energy / (input.shape[1] * input.shape[2] * input.shape[3])

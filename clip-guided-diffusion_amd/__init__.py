"""cgd_mi355x: MI355X-native (gfx950) CLIP-guided diffusion sampling step behind the reference's
`cgd.cgd.clip_guided_diffusion()` entry point and cond_fn / losses / MakeCutouts plugin surface.

The directory name carries a hyphen (`clip-guided-diffusion_amd`); import it as `cgd_amd` through the
repo-root shim `cgd_amd.py`.
"""
__all__ = ["lib", "ops", "nets", "diffusion", "guidance", "sampler"]

"""UNet hyper-parameters of the published guided-diffusion checkpoints, keyed like the reference's
DIFFUSION_LOOKUP['cond'|'uncond'][image_size] (/root/reference/data/diffusion_model_flags.py:1-120): a spec table —
it fixes every layer shape the HIP kernels must cover.  Built from a shared base plus per-checkpoint deltas."""

_OPENAI = "https://openaipublic.blob.core.windows.net/diffusion/jul-2021/"
_BASE = dict(diffusion_steps=1000, learn_sigma=True, noise_schedule="linear", num_channels=256, num_res_blocks=2,
             resblock_updown=True, use_fp16=True, use_scale_shift_norm=True)


def _entry(url, cond, size, **delta):
    flags = dict(_BASE, class_cond=cond, image_size=size, attention_resolutions="32,16,8")
    flags.update(delta)
    return {"url": url, "filename": url.rsplit("/", 1)[-1], "model_flags": flags}


_BIG = dict(attention_resolutions="32, 16, 8", rescale_timesteps=True, timestep_respacing="1000", num_head_channels=64)

DIFFUSION_LOOKUP = {
    "cond": {
        64: _entry(_OPENAI + "64x64_diffusion.pt", True, 64, dropout=0.1, noise_schedule="cosine", num_channels=192,
                   num_head_channels=64, num_res_blocks=3, use_new_attention_order=True),
        128: _entry(_OPENAI + "128x128_diffusion.pt", True, 128, num_heads=4),
        256: _entry(_OPENAI + "256x256_diffusion.pt", True, 256, num_head_channels=64),
        512: _entry(_OPENAI + "512x512_diffusion.pt", True, 512, **_BIG),
    },
    "uncond": {
        256: _entry(_OPENAI + "256x256_diffusion_uncond.pt", False, 256, num_head_channels=64),
        512: _entry("https://the-eye.eu/public/AI/models/512x512_diffusion_unconditional_ImageNet/"
                    "512x512_diffusion_uncond_finetune_008100.pt", False, 512, **_BIG),
    },
}

# upstream guided_diffusion.script_util.model_and_diffusion_defaults()
MODEL_AND_DIFFUSION_DEFAULTS = dict(
    image_size=64, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1, num_head_channels=-1,
    attention_resolutions="16,8", channel_mult="", dropout=0.0, class_cond=False, use_checkpoint=False, use_scale_shift_norm=True,
    resblock_updown=False, use_fp16=False, use_new_attention_order=False, learn_sigma=False, diffusion_steps=1000,
    noise_schedule="linear", timestep_respacing="", use_kl=False, predict_xstart=False, rescale_timesteps=False,
    rescale_learned_sigmas=False)

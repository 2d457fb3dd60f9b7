"""eo_diffusion_amd: MI355X (gfx950) native hot path of EODiffusion -- UNet forward + DDPM/DDIM samplers.

Public surface mirrors the reference's modules:
    eo_diffusion_amd.backbones.unet_openai   UNetModel, ResBlock, AttentionBlock, ...
    eo_diffusion_amd.diffusion.model         EODiffusion
    eo_diffusion_amd.diffusion.ddim          DDIMSampler
    eo_diffusion_amd.diffusion.util          schedule helpers
Add `eo_diffusion_amd/dropin` to PYTHONPATH to get the reference's top-level module paths
(`from backbones.unet_openai import UNetModel`, `from diffusion.model import EODiffusion`).
"""
__version__ = "0.1.0"

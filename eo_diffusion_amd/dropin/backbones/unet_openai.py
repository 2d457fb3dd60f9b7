"""Drop-in module path of the reference (`backbones.unet_openai`): re-exports eo_diffusion_amd.backbones.unet_openai."""
from eo_diffusion_amd.backbones.unet_openai import *  # noqa: F401,F403
from eo_diffusion_amd import backbones as _pkg  # noqa: F401
import eo_diffusion_amd.backbones.unet_openai as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})

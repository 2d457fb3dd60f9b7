"""Drop-in module path of the reference (`diffusion.ddpm`): re-exports eo_diffusion_amd.diffusion.ddpm."""
from eo_diffusion_amd.diffusion.ddpm import *  # noqa: F401,F403
from eo_diffusion_amd.diffusion.ddpm import DDPM  # noqa: F401

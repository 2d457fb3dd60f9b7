"""Drop-in module path of the reference (`diffusion.ddim`): re-exports eo_diffusion_amd.diffusion.ddim."""
from eo_diffusion_amd.diffusion.ddim import *  # noqa: F401,F403
from eo_diffusion_amd import diffusion as _pkg  # noqa: F401
import eo_diffusion_amd.diffusion.ddim as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})

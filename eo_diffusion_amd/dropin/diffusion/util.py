"""Drop-in module path of the reference (`diffusion.util`): re-exports eo_diffusion_amd.diffusion.util."""
from eo_diffusion_amd.diffusion.util import *  # noqa: F401,F403
from eo_diffusion_amd import diffusion as _pkg  # noqa: F401
import eo_diffusion_amd.diffusion.util as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})

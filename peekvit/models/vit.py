from peekvit_amd.models.vit import *  # noqa: F401,F403
from peekvit_amd.models import vit as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

from peekvit_amd.models.blocks import *  # noqa: F401,F403
from peekvit_amd.models import blocks as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

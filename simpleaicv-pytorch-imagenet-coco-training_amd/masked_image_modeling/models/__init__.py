from .vit_mae import *  # noqa: F401,F403

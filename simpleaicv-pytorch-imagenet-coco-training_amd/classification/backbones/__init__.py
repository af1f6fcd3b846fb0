from .resnet import *  # noqa: F401,F403

from .resnet import *  # noqa: F401,F403
from .resnetforcifar import *  # noqa: F401,F403

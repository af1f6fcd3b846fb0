"""Name -> factory registry, like simpleAICV/detection/models/__init__.py: the reference's
train_config.py does `models.__dict__['resnet50_retinanet'](**kwargs)`."""
from .retinanet import *  # noqa: F401,F403
from .fcos import *  # noqa: F401,F403
from .retinanet import __all__ as _retina_all
from .fcos import __all__ as _fcos_all

__all__ = list(_retina_all) + list(_fcos_all)

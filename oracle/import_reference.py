"""Oracle-side loader for the reference implementation (TEST INFRASTRUCTURE ONLY).

Imports `simpleAICV.detection.{models,losses,decode}` from /root/reference on CPU
with empty stub modules for third-party packages the reference imports at module
top level but never uses on the RetinaNet/FCOS path.

Only `oracle/make_golden.py` uses this, and only in the build container;
/root/reference does not exist on the GPU box and nothing under tests/,
bench.py or the product package may import this file.
"""
import importlib.machinery
import sys
import types

REFERENCE_ROOT = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    """Return (models, losses, decode) modules of the reference, CPU only."""
    sys.dont_write_bytecode = True
    for name in ('cv2', 'traitlets', 'thop', 'torchvision'):
        if name in sys.modules:
            continue
        try:
            __import__(name)
        except Exception:
            if name == 'cv2':
                _stub('cv2')
            elif name == 'traitlets':
                _stub('traitlets', Instance=object)
            elif name == 'thop':
                _stub('thop', profile=None, clever_format=None)
            elif name == 'torchvision':
                tv = _stub('torchvision')
                tv.transforms = _stub('torchvision.transforms',
                                      Compose=lambda ts: ts)
                tv.ops = _stub('torchvision.ops', nms=None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from simpleAICV.detection import models, losses, decode
    return models, losses, decode


def load_reference_classification():
    """Return (backbones.resnet, losses) modules of simpleAICV.classification, CPU only.  The resnet module is
    imported by file name: the package __init__ pulls every backbone family (and their third-party imports)."""
    load_reference()
    import importlib
    resnet = importlib.import_module('simpleAICV.classification.backbones.resnet')
    closses = importlib.import_module('simpleAICV.classification.losses')
    return resnet, closses

"""Importable alias of the product package.

The package directory mandated for this repository
(`simpleaicv-pytorch-imagenet-coco-training_amd/`) is not a valid Python identifier, so this
stub package points its search path there: `import sacv_amd.detection.losses` loads
`simpleaicv-pytorch-imagenet-coco-training_amd/detection/losses.py`.
"""
import os as _os

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
PACKAGE_DIR = _os.path.join(_ROOT, 'simpleaicv-pytorch-imagenet-coco-training_amd')
__path__ = [PACKAGE_DIR]

with open(_os.path.join(PACKAGE_DIR, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(PACKAGE_DIR, '__init__.py'), 'exec'))

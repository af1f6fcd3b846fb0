"""MI355X-native RetinaNet/FCOS detection training step for SimpleAICV (import as `sacv_amd`).

Layout: csrc/ (HIP kernels + C ABI, built into lib/libsacv_hip.so), _lib.py (ctypes binding),
ops.py (tensor-level wrappers), engine.py (forward/backward executor), detection/ (drop-in
mirrors of simpleAICV.detection.{models,losses,decode}), parallel.py / train.py (data-parallel
step driver restating tools/scripts.py::train_detection).
"""
__version__ = '0.1.0'

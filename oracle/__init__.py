"""CPU oracle for the SimpleAICV detection hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy for index/box arithmetic, torch-CPU fp32
for the floating-point layers and losses) of the reference algorithms on the
RetinaNet/FCOS-ResNet50 training + decode path.  Every
function cites the reference file:line it follows.

Pinning: the restatement is checked against outputs of the reference itself,
generated in the build container by `oracle/make_golden.py` (which imports
/root/reference on CPU) and committed as small fixtures under `tests/golden/`.
The reference has no tests/golden vectors of its own for this path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package, and only as the checker / reported CPU baseline.  The product
package never imports it and fails loudly when the HIP library is missing.
"""

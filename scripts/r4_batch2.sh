#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== compile tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "compile" 2>&1 | tail -25
echo "== fcos wino vs direct vs fp64"; timeout -k 10 900 python3 scripts/diag_wino_fcos.py 2>&1 | grep -v pretrained | tail -8

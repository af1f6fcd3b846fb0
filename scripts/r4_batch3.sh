#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== compile tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "compile" 2>&1 | tail -25
echo "== fcos seeds"; timeout -k 10 1000 python3 scripts/diag_wino_fcos2.py 2>&1 | grep -v pretrained | tail -14

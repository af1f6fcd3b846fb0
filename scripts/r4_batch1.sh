#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_batch1
echo "== fcos wino vs direct: tree"; python3 scripts/diag_wino_fcos.py 2>&1 | grep -v pretrained
echo "== fcos wino vs direct: prev lib"; SACV_HIP_LIB=$PWD/gpurun_variants/libsacv_prev.so python3 scripts/diag_wino_fcos.py 2>&1 | grep -v pretrained
echo "== A/B retina"; bash scripts/ab_lib2.sh "" 3 tree libsacv_prev.so
echo "== A/B fcos"; bash scripts/ab_lib2.sh "--model fcos" 2 tree libsacv_prev.so

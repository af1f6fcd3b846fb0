#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== trunk winograd wgrad microbench"; python3 scripts/bench_wino_wgrad_trunk.py 2>&1 | tail -5
echo "== tests (winograd, streams, e2e)"; timeout -k 10 600 python3 -m pytest tests/test_gpu_winograd.py tests/test_gpu_streams.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3
echo "== A/B engine"; bash scripts/ab_engine.sh 3 "" "--engine-set wino_trunk_wgrad=True"

#!/bin/bash
# round 6, j: small-batch decode with the weight-prefetch branch in the graph (SL_DECODE_PREFETCH=1, default) against without (=0), alternating; ids parity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_j; mkdir -p $O
for i in 1 2; do for v in 0 1; do echo "SL_DECODE_PREFETCH=$v"; SL_DECODE_PREFETCH=$v python tools/time_decode_step.py 1 2 4 8 16 26 2>&1 | grep "decode step"; done; done > $O/decode_prefetch_ab.txt
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "llama" 2>&1 | tail -4 > $O/pytest_llama.txt
cat $O/decode_prefetch_ab.txt $O/pytest_llama.txt

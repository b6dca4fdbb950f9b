#!/bin/bash
# HBM traffic of the two kernels that carry a decode step (gate/up streaming GEMM, decode attention): separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -f $O/${1:-r04}_pmc_decode_kernels.json
for B in 512 1024; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${B}_$c -- python3 $R/tools/probe_decode_kernels.py $B > /dev/null 2>&1
  done
  python3 $R/tools/pmc_decode.py $O/pmc_${B}_FETCH_SIZE $O/pmc_${B}_WRITE_SIZE $B $O/${1:-r04}_pmc_decode_kernels.json
  rm -rf $O/pmc_${B}_FETCH_SIZE $O/pmc_${B}_WRITE_SIZE
done

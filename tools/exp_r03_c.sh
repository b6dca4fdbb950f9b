#!/bin/bash
# round-3 experiment batch C: decode GEMM forms at 512 / 1024 rows, KD per-rank regime profile
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for v in "default" "SL_STREAM_WIDE=0" "SL_STREAM_WSPLITS=1" "SL_STREAM_WSPLITS=3"; do
  echo "=== $v" >> $O/r03_c_stream.txt
  if [ "$v" = "default" ]; then python3 $R/tools/tune_stream.py 512,1024 default >> $O/r03_c_stream.txt 2>&1
  else env $v python3 $R/tools/tune_stream.py 512,1024 default >> $O/r03_c_stream.txt 2>&1; fi
done
echo "=== tiled (row-major) kernels" >> $O/r03_c_stream.txt
python3 $R/tools/bench_decode_gemm.py 512 1024 >> $O/r03_c_stream.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kd2 -- python3 $R/bench.py --batch 1 --steps 1 --warmup 0 --max-new-tokens 2 --pipelines 1 --no-cpu-baseline --kd-optimizer-steps 6 --kd-window 2 --kd-local-accum 0 --no-length-mix --no-extra-legs > $O/r03_c_kd2_line.json 2> $O/r03_c_kd2.err
f=$(find $O/prof_kd2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03_c_kd2_kernel_stats.csv; rm -rf $O/prof_kd2

#!/bin/bash
# round 6, aq: per-launch durations of the encoder's conv-stack GEMMs (grouped implicit GEMMs on the non-swapped 256-tile epilogue) and the pos-conv GEMM
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_aq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/trace
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/prof_encoder.py 256 > $O/encoder.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/encoder_launches.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# last pass: from the last conv0 launch to the end
idx = [i for i, e in enumerate(ev) if "conv0_ln_gelu" in e[2]]
w = ev[idx[-1]:]
t0 = w[0][0]
print("last encoder pass: %d kernels, %.2f ms" % (len(w), (w[-1][1] - t0) / 1e6))
for s, e, n in w[:40]:
    print("%9.1f us  +%8.1f us  %s" % ((e - s) / 1e3, (s - t0) / 1e3, n[:110]))
PY
rm -rf $O/trace
cd $R; cat $O/encoder_launches.txt | head -60; tail -2 $O/encoder.log

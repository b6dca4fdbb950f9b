#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/tools/probe_lm_head.py 1024 both > $O/r03_k_lmhead.txt 2>&1
for m in logits fused; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${m}_$c -- python3 $R/tools/probe_lm_head.py 1024 $m > /dev/null 2>&1
  done
done
python3 - <<PY >> $O/r03_k_lmhead.txt
import csv, glob, json, statistics, os
O = "$O"
out = {}
for m in ("logits", "fused"):
    out[m] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = {}
        for f in glob.glob(os.path.join(O, f"pmc_{m}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == c:
                    k = row["Kernel_Name"].split("(")[0][:60]
                    per.setdefault(k, []).append(float(row["Counter_Value"]))
        for k, v in per.items():
            if "gemm_tiled" in k or "greedy" in k:
                out[m].setdefault(k, {})[c + "_KB_median"] = statistics.median(v)
                out[m][k]["launches"] = len(v)
for m in out:
    tot = 0.0
    for k, d in out[m].items():
        d["hbm_bytes_per_launch"] = d.get("FETCH_SIZE_KB_median", 0) * 1024 * 2 + d.get("WRITE_SIZE_KB_median", 0) * 1024
        tot += d["hbm_bytes_per_launch"]
    out[m]["total_bytes_per_step"] = tot
out["method"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace, tools/probe_lm_head.py 1024 <form>; FETCH_SIZE x2 (gfx950), KiB units"
json.dump(out, open(os.path.join(O, "r03_pmc_lm_head.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_*

"""Per-kernel means of the counters of one rocprofv3 --pmc pass.

    python tools/pmc_sq.py <pass dir> <kernel substring> [out.json]
"""
import collections, csv, glob, json, os, sys

d, ksub = sys.argv[1:3]
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if ksub in row.get("Kernel_Name", ""):
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in vals.items()}
res["launches"] = max((len(v) for v in vals.values()), default=0)
wc = res.get("SQ_WAVE_CYCLES")
if wc:
    for k in list(res):
        if k.startswith("SQ_") and k != "SQ_WAVE_CYCLES":
            res[k + "/WAVE_CYCLES"] = round(res[k] / wc, 4)
print(json.dumps(res, indent=1))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        json.dump(res, fh, indent=1)

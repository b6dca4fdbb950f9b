"""Summarise rocprofv3 --pmc passes of tools/probe_gateup.py into profiles/<name>.json.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <kernel substring> <algorithmic bytes> <out.json> [label]

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of 16 B/lane streams
at 64 bytes, so reads are doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, glob, json, os, statistics, sys


def counter_values(d, counter, kernel_sub):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and kernel_sub in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    return vals


fetch_dir, write_dir, ksub, alg, out = sys.argv[1:6]
label = sys.argv[6] if len(sys.argv) > 6 else ksub
fv, wv = counter_values(fetch_dir, "FETCH_SIZE", ksub), counter_values(write_dir, "WRITE_SIZE", ksub)
assert fv and wv, (len(fv), len(wv))
fk, wk = statistics.median(fv), statistics.median(wv)
rd, wr = fk * 1024 * 2, wk * 1024
res = {"kernel": label, "launches": len(fv), "FETCH_SIZE_KB_median": fk, "WRITE_SIZE_KB_median": wk,
       "hbm_read_bytes_per_launch_corrected_x2": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": int(alg), "traffic_over_algorithmic": (rd + wr) / float(alg),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace --output-format csv, "
                 "tools/probe_gateup.py (84 launches cycling 28 weight matrices); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(gfx950 counts 128-B requests at 64 B for 16 B/lane streams); counters in KiB"}
with open(out, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res))

"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/probe_decode_kernels.py  ->  per-launch HBM bytes of the decode kernels.

    python tools/pmc_decode.py <fetch_dir> <write_dir> <B> <out.json>      (merges into an existing out.json under key str(B))

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of 16 B/lane streams at 64 bytes,
so reads are doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).  "attn" = the single-pass kernel (B * n_kv >= 1024, bf16) or
split kernel + merge kernel of one launch pair."""
import csv, glob, json, os, statistics, sys


def med(d, counter, sub):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and sub in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        return None, 0
    return statistics.median(vals), len(vals)


fd, wd, B, out = sys.argv[1:5]
res = {}
detail = {}
for key, subs in (("gemm", ["gemm_stream_kernel", "gemm_stream_wide_kernel", "gemm_tiled256p_kernel"]), ("attn", ["attn_decode_full_kernel", "attn_decode_split_kernel", "attn_decode_combine_kernel"])):
    tot = 0.0
    for sub in subs:   # whichever of the attention forms this batch dispatched to
        fk, n = med(fd, "FETCH_SIZE", sub)
        wk, _ = med(wd, "WRITE_SIZE", sub)
        if fk is None or wk is None:
            continue
        tot += fk * 1024 * 2 + wk * 1024
        detail[sub] = {"launches": n, "FETCH_SIZE_KB_median": fk, "WRITE_SIZE_KB_median": wk}
    res[key] = tot
res["detail"] = detail
res["method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace --output-format csv, "
                 "tools/probe_decode_kernels.py; FETCH_SIZE doubled (gfx950), KiB units")
allres = {}
if os.path.exists(out):
    with open(out) as fh:
        allres = json.load(fh)
allres[str(B)] = res
with open(out, "w") as fh:
    json.dump(allres, fh, indent=1)
print(json.dumps(res))

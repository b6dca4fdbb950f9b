"""Per-kernel matrix-core / VALU utilisation from ONE rocprofv3 --pmc pass (counters: GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT).

    python tools/pmc_kernels.py <pass dir> <out.json>

Per kernel name (template arguments kept): launches, mean duration in cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs),
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / duration, valu_issue = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES,
stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, bf16 MFMA ops, LDS bank conflict cycles."""
import collections, csv, glob, json, os, re, sys

d, out = sys.argv[1:3]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*$", "", row.get("Kernel_Name", "").replace("(anonymous namespace)::", "")).replace("void ", "").strip()
            vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for name, cs in vals.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    dur = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if dur <= 0:
        continue
    n = max(len(v) for v in cs.values())
    r = {"launches": n, "duration_cycles": round(dur), "total_cycles": round(dur * n)}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        r["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / dur, 4)
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        r["valu_issue_per_wave_cycle"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
        r["issue_stalled_per_wave_cycle"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in m:
        r["mfma_mops_bf16"] = round(m["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
    if "SQ_LDS_BANK_CONFLICT" in m:
        r["lds_bank_conflict_cycles"] = round(m["SQ_LDS_BANK_CONFLICT"])
    res[name] = r
tot = sum(r["total_cycles"] for r in res.values())
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_cycles"]))
for r in res.values():
    r["share_of_kernel_time"] = round(r["total_cycles"] / tot, 4)
json.dump(res, open(out, "w"), indent=1)
for k, r in list(res.items())[:14]:
    print(f"{r['share_of_kernel_time']:6.3f} mfma_busy {r.get('mfma_busy', 0):5.3f} valu {r.get('valu_issue_per_wave_cycle', 0):5.3f} stalled {r.get('issue_stalled_per_wave_cycle', 0):5.3f} x{r['launches']:5d} {k[:90]}")

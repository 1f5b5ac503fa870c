"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass (plus an
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE pass) into per-kernel-family MFMA-pipe and LDS-conflict figures.
    python tools/summarize_mfma.py gpurun_out/pmc_mfma gpurun_out/pmc_lds profiles/r01_mfma_lds.json"""
import csv, glob, json, os, re, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import family, one_csv  # noqa: E402  (same kernel-name -> family map, same one-pass-per-dir rule)


def load(d):
    f = one_csv(d)
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = family(re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")))
        if k is None:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        acc[k]["ns:" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        n[(k, r["Counter_Name"])] += 1
    return acc, n


mf, nm = load(sys.argv[1])
ld, nl = load(sys.argv[2])
out = {"note": "rocprofv3 --pmc passes on bench.py rny002_b8 bf16 --no-graph; sums over all launches of a kernel family. "
               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch time x 2.4 GHz x 1024 SIMDs): share of the chip's matrix-pipe "
               "cycles in use while the kernel runs (profiled dispatches run one at a time); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles per active LDS cycle)",
       "kernels": {}}
for k in sorted(set(mf) | set(ld)):
    e = {}
    if mf[k].get("ns:SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_util"] = round(mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mf[k]["ns:SQ_VALU_MFMA_BUSY_CYCLES"] * 2.4 * 1024), 4)
        e["launches_profiled"] = nm[(k, "SQ_VALU_MFMA_BUSY_CYCLES")]
    if ld[k].get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac"] = round(ld[k].get("SQ_LDS_BANK_CONFLICT", 0.0) / ld[k]["SQ_LDS_IDX_ACTIVE"], 4)
    out["kernels"][k] = e
    print(f"{k:16s} {e}")
json.dump(out, open(sys.argv[3], "w"), indent=1)

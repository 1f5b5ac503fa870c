"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.
Units and gfx950 corrections as MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB;
FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 -> doubled; WRITE_SIZE is exact.
    python tools/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_hbm_traffic.json"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def load(d, counter):
    f = (glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv")))[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    return acc

FAMILY = [("gemm_ws_kernel", "gemm_ws"), ("gemm_splitk", "gemm_splitk"), ("gemm_kernel", "gemm"), ("gconv3x3", "gconv3x3"), ("s1_front", "s1_front"),
          ("gsf_", "gate_shift"), ("se_gate", "se_gate"), ("stem_kernel", "stem"), ("mixer_branch", "mixer_branch"),
          ("sgp_branch", "sgp_branch"), ("layernorm", "layernorm"), ("groupnorm", "groupnorm"), ("maxpool", "maxpool"),
          ("avgpool", "avgpool_posenc"), ("heads", "heads"), ("bneck", "bneck")]

def family(n):
    for k, v in FAMILY:
        if k in n:
            return v
    return None

def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    fam = defaultdict(lambda: dict(launches=0, fetch_bytes=0.0, write_bytes=0.0))
    for n, (v, c) in fe.items():
        k = family(n)
        if k:
            fam[k]["launches"] += c
            fam[k]["fetch_bytes"] += v * 1024 * 2          # gfx950: FETCH_SIZE = 1/2 of wide streaming reads
    for n, (v, c) in wr.items():
        k = family(n)
        if k:
            fam[k]["write_bytes"] += v * 1024
    res = {}
    for k, d in fam.items():
        L = max(d["launches"], 1)
        res[k] = dict(launches_profiled=d["launches"], hbm_bytes_per_launch=round((d["fetch_bytes"] + d["write_bytes"]) / L),
                      fetch_bytes_per_launch=round(d["fetch_bytes"] / L), write_bytes_per_launch=round(d["write_bytes"] / L))
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, FETCH doubled (gfx950), "
                        "bench.py rny002_b8 bf16; averages over every launch of the kernel family",
                   kernels=res), open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"]):
        print(f"{k:16s} launches {v['launches_profiled']:5d}  HBM/launch {v['hbm_bytes_per_launch']/1e6:9.2f} MB (rd {v['fetch_bytes_per_launch']/1e6:8.2f} wr {v['write_bytes_per_launch']/1e6:8.2f})")


if __name__ == "__main__":
    main()

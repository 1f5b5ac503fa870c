"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.
Units and gfx950 corrections as MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB;
FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 -> doubled; WRITE_SIZE is exact.

    python tools/summarize_pmc.py <fetch pass dir> <write pass dir> profiles/rNN_hbm_traffic.json [--cmd "..."] [--steps N]

Each pass directory must hold exactly ONE *counter_collection.csv (anywhere below it): a directory that collected
several passes is ambiguous and refused, so a stale pass can never be summarised by accident.  The output records which
files were read (path, mtime, size), the git HEAD and the profiled command."""
import csv
import glob
import json
import os
import re
import subprocess
import sys
import time
from collections import defaultdict

# training-step kernels (trunk_bwd.hip, sgp_bwd.hip, gsf_bwd.hip, train.hip) first: their names contain inference names
TRAIN_FAMILY = [("narrow_conv1_bwd", "narrow_conv_bwd"), ("gemm_rs", "gemm"), ("multi_cast_transpose", "repack"), ("bn_sums_from_parts", "bn_bwd"),
                ("se_bn_", "se_bn_bwd"), ("bn_parts_finalize", "bn_bwd"), ("gsf_add_cols", "gate_shift_bwd"),
                ("bn_bwd_apply", "bn_bwd"), ("colstats", "bn_bwd"), ("wgrad", "wgrad"), ("gconv_dgrad", "gconv_bwd"),
                ("gconv_wgrad", "gconv_bwd"), ("gsf_bwd", "gate_shift_bwd"), ("affine_kernel", "bn_apply"),
                ("bn_apply", "bn_apply"), ("pool_mean", "se_train"), ("scale_rows", "se_train"), ("se_train", "se_train"),
                ("adamw", "adamw"), ("multi_fold", "grad_writeout"), ("gather_cast", "repack"), ("stem_mfma", "stem"),
                ("stem_wgrad", "stem")]
FAMILY = [("sgp_gemm", "sgp_gemm"), ("sgp_fold", "sgp_fold"), ("bneck_kernel", "bneck"), ("c1_gconv", "c1_gconv"), ("gemm_ws_kernel", "gemm_ws"),
          ("gemm_rs", "gemm_ws"),                  # the register-stationary forms: the family bench.py's kernels table counts them in
          ("gemm_splitk", "gemm_splitk"), ("gemm_big", "gemm_big"), ("gemm_kernel", "gemm"),
          ("gconv3x3", "gconv3x3"), ("s1_front", "s1_front"), ("gsf_", "gate_shift"), ("se_gate", "se_gate"),
          ("stem_kernel", "stem"), ("sgp_mlp", "sgp_mlp"), ("sgp_front", "sgp_front"), ("mixer_front", "mixer_front"), ("mixer_branch", "mixer_branch"),
          ("sgp_branch", "sgp_branch"), ("layernorm", "layernorm"), ("groupnorm", "groupnorm"), ("maxpool", "maxpool"),
          ("avgpool", "avgpool_posenc"), ("heads", "heads"), ("bneck", "bneck")]


def one_csv(d):
    fs = sorted(set(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)))
    if len(fs) != 1:
        raise SystemExit(f"{d}: expected exactly one *counter_collection.csv below it, found {len(fs)}: {fs}")
    return fs[0]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    if not acc:
        raise SystemExit(f"{path}: no {counter} rows")
    return acc


def family(n, train=False):
    for k, v in (TRAIN_FAMILY + FAMILY) if train else FAMILY:
        if k in n:
            return v
    return "other" if train else None


def stamp(path):
    st = os.stat(path)
    return dict(path=path, mtime=time.strftime("%Y-%m-%dT%H:%M:%S", time.localtime(st.st_mtime)), bytes=st.st_size)


def main():
    args = [a for a in sys.argv[1:]]
    cmd, steps = None, None
    if "--cmd" in args:
        i = args.index("--cmd")
        cmd = args[i + 1]
        del args[i:i + 2]
    if "--steps" in args:                        # forward passes the profiled command ran (warm-up + timed): per-step totals
        i = args.index("--steps")
        steps = int(args[i + 1])
        del args[i:i + 2]
    workload = None
    if "--train-workload" in args:               # the passes profiled training steps of this bench workload
        i = args.index("--train-workload")
        workload = args[i + 1]
        del args[i:i + 2]
    train = workload is not None
    fd, wd, out = args[:3]
    ff, wf = one_csv(fd), one_csv(wd)
    fe, wr = load(ff, "FETCH_SIZE"), load(wf, "WRITE_SIZE")
    fam = defaultdict(lambda: dict(launches=0, fetch_bytes=0.0, write_bytes=0.0))
    for n, (v, c) in fe.items():
        k = family(n, train)
        if k:
            fam[k]["launches"] += c
            fam[k]["fetch_bytes"] += v * 1024 * 2          # gfx950: FETCH_SIZE = 1/2 of wide streaming reads
    for n, (v, c) in wr.items():
        k = family(n, train)
        if k:
            fam[k]["write_bytes"] += v * 1024
    res = {}
    for k, d in fam.items():
        L = max(d["launches"], 1)
        res[k] = dict(launches_profiled=d["launches"], hbm_bytes_per_launch=round((d["fetch_bytes"] + d["write_bytes"]) / L),
                      fetch_bytes_per_launch=round(d["fetch_bytes"] / L), write_bytes_per_launch=round(d["write_bytes"] / L))
        if steps:
            # a bench "launch" of a family can be several kernels (a gate-shift site is 3): per-forward totals let the bench
            # line divide by ITS launch count
            res[k]["kernel_launches_per_forward"] = round(d["launches"] / steps, 2)
            res[k]["hbm_bytes_per_forward"] = round((d["fetch_bytes"] + d["write_bytes"]) / steps)
            if train:
                res[k]["hbm_bytes_per_step"] = res[k].pop("hbm_bytes_per_forward")
                res[k]["kernel_launches_per_step"] = res[k].pop("kernel_launches_per_forward")
    try:
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                              cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except OSError:
        head = None
    if not head:                                 # the GPU box has the tree without .git: the revision build() recorded
        try:
            bi = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "t-deed_amd", "csrc",
                                             "build_info.json")))
            head = bi.get("git_head") + ("+dirty" if bi.get("dirty") else "")
        except (OSError, ValueError, TypeError):
            head = None
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, FETCH doubled (gfx950); "
                        "averages over every launch of the kernel family",
                   command=cmd, forwards_profiled=steps, git_head=head, fetch_pass=stamp(ff), write_pass=stamp(wf), kernels=res,
                   **(dict(workload=workload, step_hbm_bytes=round(sum((d["fetch_bytes"] + d["write_bytes"]) for d in fam.values())
                                                                  / max(steps or 1, 1))) if train else {})),
              open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"]):
        print(f"{k:16s} launches {v['launches_profiled']:5d}  HBM/launch {v['hbm_bytes_per_launch']/1e6:9.2f} MB "
              f"(rd {v['fetch_bytes_per_launch']/1e6:8.2f} wr {v['write_bytes_per_launch']/1e6:8.2f})")


if __name__ == "__main__":
    main()

"""Kernel trace grouped by (kernel, grid, LDS bytes): per-shape launch counts and durations of a rocprofv3 --kernel-trace
directory -- the per-SITE view of kernels that run at several geometries (gate-shift, bneck, sgp_gemm).
    python tools/trace_by_shape.py <dir> [name filter]"""
import csv, glob, re, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    if flt and flt not in n:
        continue
    key = (n[:60], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_In_Bytes", "")))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0, 1 << 62])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for k, (c, t, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:60s} grid {k[1]:>9s} lds {k[2]:>7s} calls {c:5d} avg {t / c / 1e3:8.1f} us min {mn / 1e3:8.1f}")

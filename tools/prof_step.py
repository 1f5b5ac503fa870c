"""Largest individual launches of the last training step in a rocprofv3 --kernel-trace directory.
    python tools/prof_step.py <dir> [top]"""
import csv, glob, re, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
step = rows[idx[-1]:]
end = [i for i, r in enumerate(step) if "adamw" in r["Kernel_Name"]]
step = step[:end[0] + 1] if end else step
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(dur(r) for r in step)
print(len(step), "launches,", tot / 1e6, "ms kernel time; wall", (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6)
cum = 0
for r in step:
    cum += dur(r)
    if "loss_kernel" in r["Kernel_Name"]:
        print("forward (up to the loss):", cum / 1e6, "ms")
        break
for r in sorted(step, key=lambda r: -dur(r))[:top]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:64]
    print(f"{dur(r)/1e3:8.1f} us  {name:64s} grid {r.get('Grid_Size','')}")

"""Print the top rows of a rocprofv3 --kernel-trace --stats directory, durations per step.
    python tools/prof_summary.py <dir> [steps] [top]"""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total {tot/1e6/steps:.2f} ms per step over {steps:g} steps, {sum(int(r['Calls']) for r in rows)/steps:.0f} launches per step")
for r in rows[:top]:
    print(f"  {r['Name'][:64]:64s} {int(r['Calls'])/steps:7.1f}/step  {float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step  avg {float(r['AverageNs'])/1e3:8.1f} us  max {float(r['MaxNs'])/1e3:8.1f}")

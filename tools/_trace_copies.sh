# which launches surround the runtime's copy kernels in a training step (rocprofv3 kernel + memory-copy trace)
set -eu
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/copies; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o b -- python3 tools/bench_train.py rny002_b8 8 1 > $O/run.txt 2>&1 || true
ls $O/t
python3 - <<'PY'
import csv, glob, collections
k = sorted(glob.glob('gpurun_out/copies/t/*kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(k)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][-40:] for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if 'copyBuffer' in n:
        prev = next((names[j] for j in range(i - 1, -1, -1) if 'copyBuffer' not in names[j]), '-')
        nxt = next((names[j] for j in range(i + 1, len(names)) if 'copyBuffer' not in names[j]), '-')
        ctx[(prev, nxt)] += 1
for (a, b), c in ctx.most_common(25):
    print(c, '| after', a, '| before', b)
m = glob.glob('gpurun_out/copies/t/*memory_copy_trace.csv')
if m:
    rows = list(csv.DictReader(open(m[0])))
    print('memory copies:', len(rows), rows[0].keys() if rows else '')
    c = collections.Counter((r.get('Direction', ''), r.get('Bytes', r.get('Size', ''))) for r in rows)
    for kk, v in c.most_common(15): print(v, kk)
PY

# same-box A/B of environment switches: bash tools/_ab.sh <workload> "<ENV=.. ENV=..>" "<ENV=..>" ...
# (an entry may carry bench flags after "--": "X=0 -- --inflight 4")
mkdir -p gpurun_out/ab; wl=$1; shift
for r in 1 2; do for e in "$@"; do
  envs="${e%%--*}"; flags=""; case "$e" in *--*) flags="${e#*-- }";; esac
  env $envs python bench.py --workload $wl --no-train --no-cpu-baseline --no-feed $flags > gpurun_out/ab/x.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ab/x.json').read().strip().splitlines()[-1]); print('$wl', '[$e]', d['value'], d['ms_per_step'])"
done; done

# same-box A/B of environment switches: bash tools/_ab.sh <workload> "<ENV=.. ENV=..>" "<ENV=..>" ...
mkdir -p gpurun_out/ab; wl=$1; shift
for r in 1 2; do i=0; for e in "$@"; do i=$((i+1))
  env $e python bench.py --workload $wl --no-train --no-cpu-baseline --no-feed > gpurun_out/ab/x.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ab/x.json').read().strip().splitlines()[-1]); print('$wl', '[$e]', d['value'], d['ms_per_step'])"
done; done

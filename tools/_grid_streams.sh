# experiment script: sub-batches per batch (TDEED_SPLIT) x batches in flight, three repeats each
mkdir -p gpurun_out/r03j
run() { n=$1; shift; f=$1; shift
  env "$@" python bench.py --no-train --no-cpu-baseline --no-feed --inflight $f > gpurun_out/r03j/$n.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r03j/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d['latency_ms_inflight1'])"
}
for r in 1 2 3; do
run n2_if2_$r 2 TDEED_SPLIT=2
run n1_if3_$r 3 TDEED_SPLIT=1
run n1_if2_$r 2 TDEED_SPLIT=1
run n2_if3_$r 3 TDEED_SPLIT=2
done

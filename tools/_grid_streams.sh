mkdir -p gpurun_out/r03j
run() { # name env...
  n=$1; shift
  env "$@" python bench.py --no-train --no-cpu-baseline --no-feed $EXTRA > gpurun_out/r03j/$n.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r03j/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d['latency_ms_inflight1'])"
}
EXTRA="--inflight 2"
run s0_q4 TDEED_GRAPH_SPLIT=0 GPU_MAX_HW_QUEUES=4
run s0_q8 TDEED_GRAPH_SPLIT=0 GPU_MAX_HW_QUEUES=8
run s1_q2 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=2
run s1_q4 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=4
run s1_q6 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=6
run s1_q4_n4 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=4 TDEED_SPLIT=4
run s1_q8_n4 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=8 TDEED_SPLIT=4
run s1_q4_n1 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=4 TDEED_SPLIT=1
EXTRA="--inflight 3"
run s1_q4_if3 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=4
run s1_q6_if3 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=6
run s1_q8_n1_if3 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=8 TDEED_SPLIT=1
EXTRA="--inflight 4"
run s1_q4_n1_if4 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=4 TDEED_SPLIT=1
run s1_q8_n1_if4 TDEED_GRAPH_SPLIT=1 GPU_MAX_HW_QUEUES=8 TDEED_SPLIT=1

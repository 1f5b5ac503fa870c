# same-box A/B of the host-fed measurement's knobs (copy streams the batch is cut over, batches uploaded ahead)
for e in "TDEED_FEED_STREAMS=1" "TDEED_FEED_STREAMS=1 TDEED_FEED_AHEAD=3" "TDEED_FEED_STREAMS=1 TDEED_FEED_AHEAD=4" "TDEED_FEED_STREAMS=1 TDEED_FEED_AHEAD=1" "TDEED_FEED_STREAMS=1 GPU_MAX_HW_QUEUES=4" "TDEED_FEED_STREAMS=1 TDEED_FEED_AHEAD=3 GPU_MAX_HW_QUEUES=4"; do
  env $e python bench.py --no-train --no-cpu-baseline > gpurun_out/ab/f.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ab/f.json').read().strip().splitlines()[-1]); f=d['fed_from_host']; print('[$e]', d['value'], f['value'], f['h2d_GBps'], f['frac_of_resident'])"
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { python bench.py --no-train --no-feed --no-cpu-baseline --repeats 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["value"], d["ms_per_step"], d["latency_ms_inflight1"], {k:(v["ms"],v["launches"]) for k,v in d["kernels"].items() if k in ("gemm","gemm_ws")})'; }
echo "default: $(run)" > gpurun_out/ws_exp.txt
echo "sliced cap96: $(TDEED_WS_SLICED=1 run)" >> gpurun_out/ws_exp.txt
echo "sliced cap150: $(TDEED_WS_SLICED=1 TDEED_WS_CAP_KB=150 run)" >> gpurun_out/ws_exp.txt
echo "sliced cap128: $(TDEED_WS_SLICED=1 TDEED_WS_CAP_KB=128 run)" >> gpurun_out/ws_exp.txt
cat gpurun_out/ws_exp.txt

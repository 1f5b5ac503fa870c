cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6) > gpurun_out/t_gemm.txt
run() { python bench.py --workload $1 --no-train --no-feed --no-cpu-baseline --repeats 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["value"], d["ms_per_step"], d["latency_ms_inflight1"], {k:(v["ms"],v["launches"]) for k,v in d["kernels"].items() if k in ("gemm",)})'; }
for w in rny002_b8 rny008_b16 snb_t250_b4; do
  echo "$w DB=1: $(run $w)"; echo "$w DB=0: $(TDEED_GEMM_DB=0 run $w)"
done > gpurun_out/gemm_db.txt 2>&1
for db in 1 0; do echo "train b16 DB=$db: $(TDEED_GEMM_DB=$db python bench.py --mode train --workload rny008_b16 --no-cpu-baseline --repeats 3 2>/dev/null | cut -c1-230)"; done >> gpurun_out/gemm_db.txt
tail -4 gpurun_out/t_gemm.txt; cat gpurun_out/gemm_db.txt

rm -f gpurun_out/r6_exp13.txt
python -m pytest tests/test_gpu_r6.py -x -q 2>&1 | tail -5 >> gpurun_out/r6_exp13.txt
python tools/bench_bneck.py 800 2>&1 | grep -v "amdgpu\|conv2 of wave\|SE phase" | cut -c1-330 >> gpurun_out/r6_exp13.txt
for rep in 1 2 3; do for f in 1 0; do TDEED_BNECK_QTAIL=$f python bench.py --no-train --no-feed --repeats 5 --full 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('cfg2 qtail=$f', d['value'], d['ms_per_step'], k['bneck']['ms'], k['gate_shift']['ms'], d.get('logit_max_abs_err_bf16'), d.get('logit_rms_err_bf16'))" >> gpurun_out/r6_exp13.txt; done; done

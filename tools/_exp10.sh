rm -f gpurun_out/r6_exp10.txt
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_r2.py tests/test_gpu_r5.py -x -q 2>&1 | tail -3 >> gpurun_out/r6_exp10.txt
python tools/bench_bneck.py 800 2>&1 | grep -v amdgpu | cut -c1-330 >> gpurun_out/r6_exp10.txt
for rep in 1 2 3; do for f in release prevsig; do TDEED_LIB_FLAVOUR=$f python bench.py --no-train --no-feed --repeats 5 --full 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('cfg2 $f', d['value'], d['ms_per_step'], k['bneck']['ms'], k['se_gate']['ms'], d.get('logit_max_abs_err_bf16'), d.get('logit_rms_err_bf16'), d.get('logit_max_abs_err_fp32'))" >> gpurun_out/r6_exp10.txt; done; done

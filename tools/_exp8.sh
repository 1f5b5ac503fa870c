rm -f gpurun_out/r6_exp8.txt
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" >> gpurun_out/r6_exp8.txt
for rep in 1 2 3; do for f in release previdiv; do TDEED_LIB_FLAVOUR=$f python bench.py --no-train --no-feed --no-cpu-baseline --repeats 5 --full 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('cfg2 $f', d['value'], d['ms_per_step'], {n: k[n]['ms'] for n in ('bneck','gate_shift','c1_gconv','gemm_ws','s1_front','sgp_gemm')})" >> gpurun_out/r6_exp8.txt; done; done

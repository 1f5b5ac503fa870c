rm -f gpurun_out/r6_kg.txt
python -m pytest tests/test_gpu_r5.py -x -q 2>&1 | tail -3 > gpurun_out/r6_t5.log
for k1 in 1 2 4; do for k2 in 1 2 3; do echo "== KG mode1=$k1 mode2=$k2" >> gpurun_out/r6_kg.txt; TDEED_SGP_KG_1=$k1 TDEED_SGP_KG_2=$k2 python tools/bench_sgp_gemm.py --stage-only 2>&1 | grep "stage B" >> gpurun_out/r6_kg.txt; done; done
echo "== auto" >> gpurun_out/r6_kg.txt; python tools/bench_sgp_gemm.py --stage-only 2>&1 | grep -v amdgpu >> gpurun_out/r6_kg.txt

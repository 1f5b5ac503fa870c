# same-box A/B: interleaved runs of the SGP stage graph with the library flavours
rm -f gpurun_out/r6_ab.txt
for rep in 1 2 3; do for f in release r5gemm r5all; do echo "== $f" >> gpurun_out/r6_ab.txt; TDEED_LIB_FLAVOUR=$f python tools/bench_sgp_gemm.py --stage-only 2>&1 | grep "stage B" >> gpurun_out/r6_ab.txt; done; done

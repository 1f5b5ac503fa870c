# Round-6 evidence in ONE run at ONE git HEAD (clean tree): kernel traces, HBM-traffic / MFMA / LDS counter passes for the
# headline forward, the 800MF forward and the cfg3 training step, the bench lines -- raw passes under gpurun_out/r06, the
# summaries judged from (profiles/r06_*) written by the summarisers right here, so that every file carries the same HEAD.
#   gpurun --timeout 2400 -- 'bash tools/prof_r06.sh'     then     cp gpurun_out/r06/profiles/* profiles/
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never together with trace domains
# other than the kernel trace; the program itself follows `--`).
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r06; rm -rf $O; mkdir -p $O/profiles
P=$O/profiles
INF="bench.py --no-graph --inflight 1 --steps 4 --warmup 2 --repeats 1 --pmc-pass"
INF8="bench.py --workload rny008_b16 --no-graph --inflight 1 --steps 3 --warmup 1 --repeats 1 --pmc-pass"
TRN="tools/bench_train.py rny008_b16 16 2"
run() { echo "== $*" >&2; "$@"; }
# ---- headline forward (cfg2)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_infer -o b -- python3 bench.py --no-train --no-feed --no-cpu-baseline --repeats 3 > $O/kt_infer.json 2> $O/kt_infer.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python3 $INF > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python3 $INF > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o b -- python3 $INF > $O/pmc_mfma.json 2> $O/pmc_mfma.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds -o b -- python3 $INF > $O/pmc_lds.json 2> $O/pmc_lds.err
python tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write $P/r06_hbm_traffic.json --cmd "python3 $INF" --steps 6 > $O/sum_pmc.txt 2>&1
python tools/summarize_mfma.py $O/pmc_mfma $O/pmc_lds $P/r06_mfma_lds.json > $O/sum_mfma.txt 2>&1
cp "$(ls $O/kt_infer/*/*kernel_stats.csv $O/kt_infer/*kernel_stats.csv 2>/dev/null | head -1)" $P/r06_bench_rny002_b8_kernel_stats.csv
# ---- 800MF forward, B = 16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_infer8 -o b -- python3 bench.py --workload rny008_b16 --no-train --no-feed --no-cpu-baseline --repeats 3 > $O/kt_infer8.json 2> $O/kt_infer8.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch8 -o b -- python3 $INF8 > $O/pmc_fetch8.json 2> $O/pmc_fetch8.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write8 -o b -- python3 $INF8 > $O/pmc_write8.json 2> $O/pmc_write8.err
python tools/summarize_pmc.py $O/pmc_fetch8 $O/pmc_write8 $P/r06_hbm_traffic_800mf_b16.json --cmd "python3 $INF8" --steps 4 > $O/sum_pmc8.txt 2>&1
cp "$(ls $O/kt_infer8/*/*kernel_stats.csv $O/kt_infer8/*kernel_stats.csv 2>/dev/null | head -1)" $P/r06_bench_rny008_b16_kernel_stats.csv
# ---- long clips (BASELINE configs[4] per-GPU share): counter passes for the SGP stage's traffic at T = 250
INFS="bench.py --workload snb_t250_b4 --no-graph --inflight 1 --steps 3 --warmup 1 --repeats 1 --pmc-pass"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetchs -o b -- python3 $INFS > $O/pmc_fetchs.json 2> $O/pmc_fetchs.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_writes -o b -- python3 $INFS > $O/pmc_writes.json 2> $O/pmc_writes.err
python tools/summarize_pmc.py $O/pmc_fetchs $O/pmc_writes $P/r06_hbm_traffic_snb_t250_b4.json --cmd "python3 $INFS" --steps 4 > $O/sum_pmcs.txt 2>&1
# ---- SGP stage alone
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sgp -o b -- python3 tools/bench_sgp_gemm.py --profile > $O/kt_sgp.txt 2>&1
cp "$(ls $O/kt_sgp/*/*kernel_stats.csv $O/kt_sgp/*kernel_stats.csv 2>/dev/null | head -1)" $P/r06_sgp_kernel_stats.csv
# ---- training step (cfg3 and the 200MF geometry)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train_b8 -o b -- python3 tools/bench_train.py rny002_b8 8 3 > $O/kt_train_b8.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train_b16 -o b -- python3 tools/bench_train.py rny008_b16 16 3 > $O/kt_train_b16.txt 2>&1
cp "$(ls $O/kt_train_b8/*/*kernel_stats.csv $O/kt_train_b8/*kernel_stats.csv 2>/dev/null | head -1)" $P/r06_train_rny002_b8_kernel_stats.csv
cp "$(ls $O/kt_train_b16/*/*kernel_stats.csv $O/kt_train_b16/*kernel_stats.csv 2>/dev/null | head -1)" $P/r06_train_rny008_b16_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_train_fetch -o b -- python3 $TRN > $O/pmc_train_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_train_write -o b -- python3 $TRN > $O/pmc_train_write.txt 2>&1
python tools/summarize_pmc.py $O/pmc_train_fetch $O/pmc_train_write $P/r06_train_hbm_traffic.json --cmd "python3 $TRN" --steps 3 --train-workload rny008_b16 > $O/sum_pmc_train.txt 2>&1
PROF_ALL=1 python tools/prof_train_calls.py rny008_b16 > $P/r06_train_calls_b16.txt 2>&1
# ---- bench lines (the counter summaries above are in place: the lines pick their `traffic` from them)
cp $P/r06_hbm_traffic.json $P/r06_hbm_traffic_800mf_b16.json $P/r06_hbm_traffic_snb_t250_b4.json $P/r06_train_hbm_traffic.json profiles/ 2>/dev/null
python bench.py --full > $P/r06_bench_default.json 2> $O/bench_default.err
python bench.py > $P/r06_bench_default_compact.json 2> $O/bench_default_compact.err
python bench.py --mode train --workload rny008_b16 > $P/r06_bench_train_b16.json 2> $O/bench_train_b16.err
python bench.py --mode train --workload rny002_b8 --no-cpu-baseline > $P/r06_bench_train_b8.json 2> $O/bench_train_b8.err
python bench.py --mode train --workload snb_t250_b4 --no-cpu-baseline > $P/r06_bench_train_snb.json 2> $O/bench_train_snb.err
python bench.py --workload rny008_b16 --no-train --no-feed --no-cpu-baseline > $P/r06_bench_infer_b16.json 2> $O/bench_infer_b16.err
python bench.py --workload snb_t250_b4 --no-train --no-feed --no-cpu-baseline > $P/r06_bench_infer_snb.json 2> $O/bench_infer_snb.err
git -C . rev-parse HEAD > $P/r06_git_head.txt 2>/dev/null || python -c "from tdeed_amd import buildinfo; print(buildinfo.head())" > $P/r06_git_head.txt
python tools/bench_bneck.py 800 > $P/r06_bneck_stamps.txt 2>/dev/null
python tools/bench_sgp_front.py > $P/r06_sgp_front_stamps.txt 2>/dev/null
python tools/bench_sgp_gemm.py > $P/r06_sgp_gemm_forms.txt 2>/dev/null
ls -la $P; tail -c 400 $P/r06_bench_default_compact.json; cat $O/sum_pmc_train.txt | head -30

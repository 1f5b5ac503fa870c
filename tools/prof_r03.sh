# round-3 profiling passes on the GPU box (outputs under gpurun_out/r03; summaries are made from them afterwards by
# tools/summarize_pmc.py / summarize_mfma.py and copied to profiles/).  Counter passes are separate runs (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; gpurun refuses --pmc together with trace domains other than the kernel trace).
set -eu
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
INF="bench.py --no-graph --inflight 1 --steps 4 --warmup 2 --repeats 1 --pmc-pass"
TRN="tools/bench_train.py rny008_b16 16 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_infer -o b -- python3 bench.py --no-train --no-feed --no-cpu-baseline --repeats 3 > $O/kt_infer.json 2> $O/kt_infer.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python3 $INF > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python3 $INF > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o b -- python3 $INF > $O/pmc_mfma.json 2> $O/pmc_mfma.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds -o b -- python3 $INF > $O/pmc_lds.json 2> $O/pmc_lds.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sgp -o b -- python3 tools/bench_sgp.py 8 100 368 2 --profile > $O/kt_sgp.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train_b8 -o b -- python3 tools/bench_train.py rny002_b8 8 3 > $O/kt_train_b8.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train_b16 -o b -- python3 tools/bench_train.py rny008_b16 16 3 > $O/kt_train_b16.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_train_fetch -o b -- python3 $TRN > $O/pmc_train_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_train_write -o b -- python3 $TRN > $O/pmc_train_write.txt 2>&1
python tools/prof_train_calls.py rny008_b16 > $O/train_calls_b16.txt 2>&1
python tools/stamp_sgp_mlp2.py 8 100 368 48 > $O/stamp_mlp2_48.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode train --workload rny008_b16 > $O/bench_train_b16.json 2> $O/bench_train_b16.err
python bench.py --mode train --workload rny002_b8 --no-cpu-baseline > $O/bench_train_b8.json 2> $O/bench_train_b8.err
python bench.py --mode train --workload snb_t250_b4 --no-cpu-baseline > $O/bench_train_snb.json 2> $O/bench_train_snb.err
python bench.py --workload rny008_b16 --no-train --no-feed --no-cpu-baseline > $O/bench_infer_b16.json 2> $O/bench_infer_b16.err
python bench.py --workload snb_t250_b4 --no-train --no-feed --no-cpu-baseline > $O/bench_infer_snb.json 2> $O/bench_infer_snb.err
ls -la $O | head -50; tail -c 300 $O/bench_default.json; tail -c 300 $O/bench_train_b16.json

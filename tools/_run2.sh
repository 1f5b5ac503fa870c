cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > gpurun_out/t_all.txt
python bench.py --mode train --workload rny002_b8 --no-cpu-baseline > gpurun_out/train_b8.json 2> gpurun_out/train_b8.err
python bench.py --mode train --workload rny008_b16 --no-cpu-baseline > gpurun_out/train_b16.json 2> gpurun_out/train_b16.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof14_800 -o runc -- python3 tools/bench_train.py rny008_b16 16 3 > gpurun_out/train_eager14_800.txt 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -5 gpurun_out/t_all.txt; tail -1 gpurun_out/train_b8.json | cut -c1-300;  tail -1 gpurun_out/train_b16.json | cut -c1-300; tail -1 gpurun_out/bench_default.json | cut -c1-300

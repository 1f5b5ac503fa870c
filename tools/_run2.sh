cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_bwd.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/t_bn.txt
(timeout 900 python -m pytest tests/test_gpu_r2.py -x -q -m gpu -k "train or cfg3 or bf16" 2>&1 | tail -8) > gpurun_out/t_r2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof9 -o runc -- python3 tools/bench_train.py rny002_b8 8 3 > gpurun_out/train_eager9.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof9_800 -o runc -- python3 tools/bench_train.py rny008_b16 16 3 > gpurun_out/train_eager9_800.txt 2>&1
python tools/prof_ops.py rny002_b8 8 80 > gpurun_out/prof_ops.txt 2>&1
python bench.py --mode train --workload rny002_b8 --no-cpu-baseline > gpurun_out/train_b8.json 2> gpurun_out/train_b8.err
python bench.py --mode train --workload rny008_b16 --no-cpu-baseline > gpurun_out/train_b16.json 2> gpurun_out/train_b16.err
tail -3 gpurun_out/t_bn.txt; tail -3 gpurun_out/t_r2.txt; tail -1 gpurun_out/train_b8.json | cut -c1-600;  tail -1 gpurun_out/train_b16.json | cut -c1-600

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_r2.py tests/test_gpu_model.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25) > gpurun_out/t_all.txt
python bench.py --mode train --workload rny002_b8 --no-cpu-baseline > gpurun_out/train_b8.json 2> gpurun_out/train_b8.err
python bench.py --mode train --workload rny008_b16 --no-cpu-baseline > gpurun_out/train_b16.json 2> gpurun_out/train_b16.err
tail -12 gpurun_out/t_all.txt | cut -c1-300; tail -1 gpurun_out/train_b8.json | cut -c1-250;  tail -1 gpurun_out/train_b16.json | cut -c1-250; tail -3 gpurun_out/train_b8.err

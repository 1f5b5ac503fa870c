cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests/test_gpu_r2.py tests/test_gpu_bwd.py tests/test_gpu_model.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25) > gpurun_out/t_all.txt
tail -12 gpurun_out/t_all.txt | cut -c1-300

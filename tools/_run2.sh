cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_dp.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30) > gpurun_out/t_dp.txt
cat gpurun_out/t_dp.txt | cut -c1-400

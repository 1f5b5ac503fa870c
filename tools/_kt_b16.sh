# kernel trace + stats of the 800MF B=16 inference forward, one batch in flight (GPU box): bash tools/_kt_b16.sh
set -eu
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/kt_b16; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o b -- python3 bench.py --workload rny008_b16 --no-train --no-feed --no-cpu-baseline --repeats 1 --inflight 1 > $O/run.json 2> $O/run.err
head -30 $O/k/b_kernel_stats.csv | cut -c1-200

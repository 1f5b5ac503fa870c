cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --no-train --no-cpu-baseline --repeats 3 2> gpurun_out/bf.err | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["value"], d["ms_per_step"], d["fed_from_host"])' > gpurun_out/bf.txt
cat gpurun_out/bf.txt; tail -3 gpurun_out/bf.err

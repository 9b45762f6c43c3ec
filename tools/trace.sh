# kernel trace of a short bench run -> gap analysis (bash tools/trace.sh [tag] [extra bench flags])
TAG=${1:-t}; EXTRA=${2:-}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-extras $EXTRA > $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.log 2>&1
F=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/gaps.py $F $SEQ
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG

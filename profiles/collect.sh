#!/bin/bash
# Regenerates everything under profiles/ for the current build.  Run on the GPU box from the repo root:
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh'
# then copy gpurun_out/collect/* into profiles/ (see profiles/README.md) and run profiles/pmc_to_traffic.py.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.log 2>&1
python3 bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.log 2>&1
python3 bench.py --workload cfg5 --no-cpu-baseline --steps 10 > $O/bench_cfg5.log 2>&1
python3 bench.py --no-cpu-baseline --train-loop 300 > $O/bench_train.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
find $O -name "*kernel_trace.csv" -delete      # the per-dispatch trace is large; the stats summary is what is kept
echo collect done

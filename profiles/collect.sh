#!/bin/bash
# Regenerates everything under profiles/ for the current build (round 3 naming).  Run on the GPU box from the repo root:
#   gpurun --timeout 1150 -- 'bash profiles/collect.sh'
# then `bash profiles/reduce.sh` here (copies gpurun_out/collect/* into profiles/ and reduces the counter passes).
# Counter passes are their own runs, never combined with traces (gpurun refuses that).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.log 2>&1
python3 bench.py --workload cfg2 --no-cpu-baseline --train-loop 0 > $O/bench_cfg2.log 2>&1
python3 bench.py --workload cfg5 --no-cpu-baseline --steps 10 --train-loop 0 > $O/bench_cfg5.log 2>&1
(cd profiles/valu_microbench && ./valu_microbench 20000 > $O/valu_raw.json 2> $O/valu_raw.err) || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DNDEBUG -S --cuda-device-only -o $O/gsr_render.s \
    structured-gaussian-splatting_amd/csrc/gsr_render.hip > /dev/null 2>&1 || true
cd /tmp && export TMPDIR=/tmp
B3="--no-cpu-baseline --no-secondary --no-4k --train-loop 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $B3 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3n -- python3 $R/bench.py --workload cfg3n --no-cpu-baseline --train-loop 0 > $O/stats_cfg3n.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5n -- python3 $R/bench.py --workload cfg5n --steps 10 --no-cpu-baseline --no-extras --train-loop 0 > $O/stats_cfg5n.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 $B3 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 $B3 > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_cfg5n -- python3 $R/bench.py --workload cfg5n --steps 6 --warmup 2 --no-cpu-baseline --no-extras --train-loop 0 > $O/pmc_fetch_cfg5n.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_cfg5n -- python3 $R/bench.py --workload cfg5n --steps 6 --warmup 2 --no-cpu-baseline --no-extras --train-loop 0 > $O/pmc_write_cfg5n.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU \
    --output-format csv -d $O/pmc_insts -- python3 $R/bench.py --steps 6 --warmup 2 $B3 > $O/pmc_insts.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU \
    --output-format csv -d $O/pmc_insts_cfg3n -- python3 $R/bench.py --workload cfg3n --steps 6 --warmup 2 --no-cpu-baseline --no-extras --train-loop 0 > $O/pmc_insts_cfg3n.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $O/pmc_cycles -- python3 $R/bench.py --steps 6 --warmup 2 $B3 > $O/pmc_cycles.log 2>&1
find $O -name "*kernel_trace.csv" -delete      # the per-dispatch trace is large; the stats summary is what is kept
echo collect done

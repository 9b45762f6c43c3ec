#!/bin/bash
# After `gpurun -- 'bash profiles/collect.sh'`: copy the newest run of every pass from gpurun_out/collect/ into profiles/ and
# reduce the counter passes (run from the repo root, here or on the box).
set -eo pipefail
O=gpurun_out/collect
newest() { ls -t $1 | head -1; }
for w in default:cfg3 cfg2:cfg2 cfg5:cfg5 train:train; do
  grep '^{' $O/bench_${w%%:*}.log | tail -1 > profiles/r02_${w##*:}_bench.json
done
cp "$(newest "$O/stats/*/*kernel_stats.csv")" profiles/r02_cfg3_kernel_stats.csv
cp "$(newest "$O/stats_cfg3n/*/*kernel_stats.csv")" profiles/r02_cfg3n_kernel_stats.csv
cp "$(newest "$O/pmc_fetch/*/*counter_collection.csv")" profiles/r02_pmc_fetch_size.csv
cp "$(newest "$O/pmc_write/*/*counter_collection.csv")" profiles/r02_pmc_write_size.csv
python3 profiles/pmc_to_traffic.py profiles/r02_pmc_fetch_size.csv profiles/r02_pmc_write_size.csv profiles > /dev/null
python3 profiles/sq_summary.py "$(newest "$O/pmc_insts/*/*counter_collection.csv")" "$(newest "$O/pmc_cycles/*/*counter_collection.csv")" profiles/r02_pmc_sq_summary.json > /dev/null
cp $O/valu_raw.json profiles/valu_microbench/r02_valu_raw.json
python3 profiles/valu_mix.py rates profiles/valu_microbench/r02_valu_raw.json profiles/valu_microbench/r02_valu_rates.json > /dev/null
python3 profiles/valu_mix.py mix "$(newest "$O/pmc_insts/*/*counter_collection.csv")" $O/gsr_render.s profiles/r02_valu_mix.json > /dev/null
echo reduced

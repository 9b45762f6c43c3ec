#!/bin/bash
# After `gpurun -- 'bash profiles/collect.sh'`: copy the newest run of every pass from gpurun_out/collect/ into profiles/ and
# reduce the counter passes (run from the repo root, here or on the box).  Round 3 naming.
set -eo pipefail
O=gpurun_out/collect
T=r03
newest() { ls -t $1 | head -1; }
for w in default:cfg3 cfg2:cfg2 cfg5:cfg5; do
  grep '^{' $O/bench_${w%%:*}.log | tail -1 > profiles/${T}_${w##*:}_bench.json
done
cp "$(newest "$O/stats/*/*kernel_stats.csv")" profiles/${T}_cfg3_kernel_stats.csv
cp "$(newest "$O/stats_cfg3n/*/*kernel_stats.csv")" profiles/${T}_cfg3n_kernel_stats.csv
cp "$(newest "$O/stats_cfg5n/*/*kernel_stats.csv")" profiles/${T}_cfg5n_kernel_stats.csv
python3 profiles/pmc_to_traffic.py "$(newest "$O/pmc_fetch/*/*counter_collection.csv")" "$(newest "$O/pmc_write/*/*counter_collection.csv")" profiles > /dev/null
python3 profiles/pmc_to_traffic.py "$(newest "$O/pmc_fetch_cfg5n/*/*counter_collection.csv")" "$(newest "$O/pmc_write_cfg5n/*/*counter_collection.csv")" profiles \
    ${T}_cfg5n_traffic.json ${T}_cfg5n_pmc_traffic_detail.json > /dev/null
python3 profiles/sq_summary.py "$(newest "$O/pmc_insts/*/*counter_collection.csv")" "$(newest "$O/pmc_cycles/*/*counter_collection.csv")" profiles/${T}_pmc_sq_summary.json > /dev/null
cp $O/valu_raw.json profiles/valu_microbench/${T}_valu_raw.json
python3 profiles/valu_mix.py rates profiles/valu_microbench/${T}_valu_raw.json profiles/valu_microbench/${T}_valu_rates.json > /dev/null
python3 profiles/valu_mix.py mix "$(newest "$O/pmc_insts/*/*counter_collection.csv")" $O/gsr_render.s profiles/${T}_valu_mix.json cfg3 > /dev/null
python3 profiles/valu_mix.py mix "$(newest "$O/pmc_insts_cfg3n/*/*counter_collection.csv")" $O/gsr_render.s profiles/${T}_cfg3n_valu_mix.json cfg3n > /dev/null
echo reduced

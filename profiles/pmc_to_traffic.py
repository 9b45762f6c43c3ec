#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md "HBM" prescribes) into profiles/traffic.json = HBM bytes per launch per kernel,
the number bench.py reports as roofline.traffic.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python3 profiles/pmc_to_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> profiles

Units and corrections (guide, "HBM [CDNA4]"): both counters are in KiB-like units of 1024 B as rocprofv3
reports them; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is DOUBLED; WRITE_SIZE is exact for
wide streaming stores.  The first launches of every kernel (warm-up, cold caches) are dropped.
"""
import csv
import json
import os
import sys
from collections import defaultdict

# kernel-name prefix (demangled, as rocprofv3 prints it) -> key used in bench.py's "kernels" table
NAMES = {
    "gsr::k_render_bwd": "render_bwd", "gsr::k_render_fwd": "render_fwd", "void gsr::k_geom_bwd<": "geom_bwd",
    "void gsr::k_geom_bwd_sparse<": "geom_bwd", "void gsr::k_preprocess<": "preprocess",
    "void gsr::k_reduce_rows<": "reduce_rows", "void gsr::k_loss_fwd<": "loss_fwd", "void gsr::k_loss_bwd<": "loss_bwd",
    "void gsr::k_emit_team<": "emit", "void gsr::k_count_team<": "count_open", "void gsr::k_bin_chunk<false>": "count_open",
    "void gsr::k_bin_chunk<true>": "emit", "gsr::k_ranges": "ranges", "gsr::k_tile_gather": "tile_gather",
    "gsr::k_tile_ranges": "tile_ranges", "void gsr::k_sel_hist<": "depth_hist", "gsr::k_part_count": "depth_partition",
    "gsr::k_part_scatter": "depth_partition", "gsr::k_chunk_sort_small": "chunk_sort", "gsr::k_zero_segments": "zero_outputs",
    "gsr::k_act_fwd": "activations_fwd", "gsr::k_act_bwd": "activations_bwd", "void gsr::k_chunk_colors_all<": "chunk_colors",
    "void gsr::k_chunk_colors<": "chunk_colors", "void gsr::k_radix_hist<": "radix", "gsr::k_radix_scan": "radix", "void gsr::k_radix_scatter": "radix",
    "gsr::k_adam_multi": "adam", "void gsr::k_adam<": "adam",
}
SKIP_FIRST = 3


def per_launch(path, counter):
    by = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for prefix, key in NAMES.items():
                if row["Kernel_Name"].startswith(prefix):
                    by[key].append(float(row["Counter_Value"]) * 1024.0)
                    break
    return {k: v[SKIP_FIRST:] for k, v in by.items() if len(v) > SKIP_FIRST}


def main():
    fetch_csv, write_csv, outdir = sys.argv[1:4]
    # optional: <traffic file name> <detail file name> (defaults: the cfg3 files bench.py reads)
    traffic_name = sys.argv[4] if len(sys.argv) > 4 else "traffic.json"
    detail_name = sys.argv[5] if len(sys.argv) > 5 else "r03_pmc_traffic_detail.json"
    fetch, write = per_launch(fetch_csv, "FETCH_SIZE"), per_launch(write_csv, "WRITE_SIZE")
    detail, traffic = {}, {}
    for k in sorted(set(fetch) & set(write)):
        fb = 2.0 * sum(fetch[k]) / len(fetch[k])          # gfx950 correction: x2
        wb = sum(write[k]) / len(write[k])
        detail[k] = dict(fetch_bytes_per_launch=int(fb), write_bytes_per_launch=int(wb), hbm_bytes_per_launch=int(fb + wb),
                         launches_sampled=min(len(fetch[k]), len(write[k])))
        traffic[k] = int(fb + wb)
    with open(os.path.join(outdir, traffic_name), "w") as f:
        json.dump(traffic, f, indent=1)
    with open(os.path.join(outdir, detail_name), "w") as f:
        json.dump(detail, f, indent=1)
    print(json.dumps(detail, indent=1))


if __name__ == "__main__":
    main()

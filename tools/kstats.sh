# per-kernel durations of a short bench run (rocprofv3 --kernel-trace --stats): bash tools/kstats.sh [workload] [filter]
W=${1:-cfg3}; FILT=${2:-.}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kst
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kst -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-extras --workload $W > $GRAFT_REPO_ROOT/gpurun_out/kst.log 2>&1
F=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/kst/*/*kernel_stats.csv | head -1)
python3 - "$F" "$FILT" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if re.search(sys.argv[2], n):
        print("%9.1f us avg  %5s calls  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], n[:100]))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kst

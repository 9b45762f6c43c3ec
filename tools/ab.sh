# A/B of an environment switch on the same box, runs interleaved: bash tools/ab.sh VAR [value when on, default 1] [workload] [rounds]
VAR=$1; VAL=${2:-1}; W=${3:-cfg3}; N=${4:-3}
rm -f gpurun_out/ab_$VAR.log
for i in $(seq $N); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $VAR=$VAL; else unset $VAR; fi
    python bench.py --steps ${STEPS:-40} --warmup 10 --no-secondary --no-cpu-baseline --no-extras --workload $W 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$v', d['ms_per_step'])" >> gpurun_out/ab_$VAR.log
  done
done
python - <<PY
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/ab_$VAR.log"):
    k, v = l.split(); d[k].append(float(v))
for k, v in sorted(d.items()):
    print(k, "median %.4f  min %.4f  all %s" % (statistics.median(v), min(v), v))
PY

# A/B of an environment switch on the same box: bash tools/ab.sh VAR [value when on, default 1] [workload]
VAR=$1; VAL=${2:-1}; W=${3:-cfg3}
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $VAR=$VAL; else unset $VAR; fi
    python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --workload $W 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$v', d['ms_per_step'], d['raster_ms_per_step'], d['profiled_ms_per_step'])" >> gpurun_out/ab_$VAR.log
  done
done
cat gpurun_out/ab_$VAR.log

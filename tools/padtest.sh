# residency sweep of a blend kernel through dynamic LDS padding: bash tools/padtest.sh VAR "pads" [workload]
VAR=$1; W=${3:-cfg3}
for pad in $2; do
  export $VAR=$pad
  for i in 1 2; do
  python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --workload $W 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$pad', d['ms_per_step'], d['raster_ms_per_step'], d['kernels'].get('render_fwd'), d['kernels'].get('render_bwd'))"
  done
done

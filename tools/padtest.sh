for pad in 0 6000 9500 13000 17000; do
  export GSR_BWD_LDS_PAD=$pad
  python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('pad $pad', d['ms_per_step'], d['raster_ms_per_step'])"
done

# build the library on the box with extra -D flags and print the bench's kernel table: bash tools/variant_bench.sh "-DX=1" [workload]
touch structured-gaussian-splatting_amd/csrc/*.hip
make -C structured-gaussian-splatting_amd/csrc -j8 EXTRA="$1" > gpurun_out/variant_build.log 2>&1 || { tail -5 gpurun_out/variant_build.log; exit 1; }
echo "== $1"
python bench.py --no-cpu-baseline --no-secondary --no-4k --train-loop 0 --workload ${2:-cfg3} 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print(d['value'], d['ms_per_step'], 'bwd', k['render_bwd']['ms_per_step'], 'fwd', k['render_fwd']['ms_per_step'])"

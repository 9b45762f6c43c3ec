# build the library on the box with extra -D flags and print the headline step a few times: bash tools/variant_step.sh "-DX=1" [workload] [repeats]
touch structured-gaussian-splatting_amd/csrc/*.hip
make -C structured-gaussian-splatting_amd/csrc -j8 EXTRA="$1" > gpurun_out/variant_build.log 2>&1 || { tail -5 gpurun_out/variant_build.log; exit 1; }
echo "== $1"
for i in $(seq ${3:-3}); do
python bench.py --no-cpu-baseline --no-secondary --no-4k --train-loop 0 --no-extras --workload ${2:-cfg3} 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])"
done

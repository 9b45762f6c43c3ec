# build the library on the box with an extra -D flag and print kernel statistics: bash tools/variant.sh "-DX=1" [workload] [filter]
touch structured-gaussian-splatting_amd/csrc/*.hip
make -C structured-gaussian-splatting_amd/csrc -j8 EXTRA="$1" > gpurun_out/variant_build.log 2>&1 || { tail -5 gpurun_out/variant_build.log; exit 1; }
echo "== $1"
bash tools/kstats.sh ${2:-cfg3} "${3:-.}"

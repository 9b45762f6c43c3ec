# build with extra -D flags on the box and time the loss kernels: bash tools/loss_variants.sh "-DX=1" [H W]
touch structured-gaussian-splatting_amd/csrc/gsr_loss.hip
make -C structured-gaussian-splatting_amd/csrc -j8 EXTRA="$1" > gpurun_out/variant_build.log 2>&1 || { tail -5 gpurun_out/variant_build.log; exit 1; }
echo "== $1"
python tools/loss_bench.py $2 $3 2>&1 | grep -v amdgpu.ids

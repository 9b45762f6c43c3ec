for v in default; do
  if [ $v = default ]; then unset GSR_LIB_PATH; else export GSR_LIB_PATH=$PWD/structured-gaussian-splatting_amd/lib/var/libgsrast_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_x_$v.log 2> gpurun_out/r2_x_$v.err
done
python -m pytest tests/test_gpu_parity.py -x -q -k "fixture or forward_backward or cfg1 or cfg2 or sweep or randomised" > gpurun_out/r2_x_tests.log 2>&1

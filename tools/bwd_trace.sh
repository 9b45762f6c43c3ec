# wave placement / lifetime trace of k_render_bwd: rebuilds the library with -DGSR_BWD_TRACE on the box (the in-tree library
# is untouched: the box's copy is scratch), then runs tools/bwd_trace.py.   bash tools/bwd_trace.sh [workload]
touch structured-gaussian-splatting_amd/csrc/gsr_render.hip
make -C structured-gaussian-splatting_amd/csrc EXTRA=-D${TRACE:-GSR_BWD_TRACE} > gpurun_out/bwd_trace_build.log 2>&1 && python tools/bwd_trace.py ${1:-cfg3}

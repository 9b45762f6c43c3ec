# run-to-run spread of the bench: step time, sum of kernel times (hipEvent pass), the CPU the process ran on
for i in $(seq ${1:-6}); do
python - <<'PY'
import json, os, subprocess, sys
p = subprocess.run([sys.executable, "bench.py", "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True)
d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
ks = sum(v["ms_per_step"] for v in d["kernels"].values())
print(d["ms_per_step"], "kernels %.4f" % ks, "bwd %.4f fwd %.4f" % (d["kernels"]["render_bwd"]["ms_per_step"], d["kernels"]["render_fwd"]["ms_per_step"]), "profiled", d["profiled_ms_per_step"], d.get("host", ""))
PY
done

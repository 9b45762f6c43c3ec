"""GPU timeline of the training step from a rocprofv3 --kernel-trace csv: per step (delimited by k_preprocess launches) the
busy time, the idle gaps and where the large gaps are."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_preprocess" in e[2]]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
steps = steps[len(steps) // 2:][:20]                      # late steps: warm
tot = collections.Counter(); cnt = 0
for a, b in steps:
    seg = ev[a:b + 1]
    span = seg[-1][0] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg[:-1])
    cnt += 1
    tot["span"] += span; tot["busy"] += busy
    for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
        g = s1 - e0
        if g > 3000:
            tot["gap after " + n0.split("(")[0][-40:] + " -> " + n1.split("(")[0][-40:]] += g
print("steps", cnt, "span us %.1f busy us %.1f idle us %.1f" % (tot["span"] / cnt / 1e3, tot["busy"] / cnt / 1e3, (tot["span"] - tot["busy"]) / cnt / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    if k.startswith("gap"):
        print("%8.1f us/step  %s" % (v / cnt / 1e3, k))
if len(sys.argv) > 2 and sys.argv[2] == "--seq":             # one warm step, launch by launch: start, duration, gap before
    a, b = steps[len(steps) // 2]
    t0 = ev[a][0]
    prev_end = ev[a][0]
    for s0, e0, n0 in ev[a:b]:
        print("%8.1f  dur %7.1f  gap %6.1f  %s" % ((s0 - t0) / 1e3, (e0 - s0) / 1e3, (s0 - prev_end) / 1e3, n0.split("(")[0][-70:]))
        prev_end = e0

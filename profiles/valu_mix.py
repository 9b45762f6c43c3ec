#!/usr/bin/env python3
"""VALU-issue roofline inputs for bench.py (`roofline.valu`), both measured on MI355X.

  python3 profiles/valu_mix.py rates profiles/valu_microbench/r02_valu_raw.json profiles/valu_microbench/r02_valu_rates.json
      raw output of profiles/valu_microbench/valu_microbench  ->  issue cycles one wave64 instruction of each class costs
      its SIMD at saturation.  The microbenchmark's wall-clock rate (wave-instructions per second per SIMD, every SIMD of
      the chip loaded with 8 waves) divided into the clock the same waves measured (s_memtime / s_memrealtime): the
      per-wave tick count alone over-states the rate whenever the dispatcher does not keep all W blocks of a CU resident
      together, the wall clock cannot.

  python3 profiles/valu_mix.py mix <counter_collection.csv of the SQ_INSTS pass> <gsr_render.s> profiles/r02_valu_mix.json
      per blend kernel: VALU wave-instructions per launch by class.  rocprofv3's SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F32 /
      INT32 / CVT counters give the classes (a packed instruction counts once, in its class); the disassembly of the
      kernel (hipcc -S of csrc/gsr_render.hip, same flags as the build) gives, per class, the share of packed
      encodings, and among the remaining instructions (compares, selects, min/max, DPP adds, moves) the share of moves.

`roofline.valu.frac` = sum over classes of instructions x issue cycles / (1024 SIMDs x launch duration x clock).
"""
import csv
import json
import re
import sys
from collections import defaultdict

KERNELS = {"gsr::k_render_bwd": "render_bwd", "gsr::k_render_fwd": "render_fwd"}
WORKLOAD = "cfg3"
SKIP_FIRST = 3


def rates(raw_path, out_path):
    raw = json.load(open(raw_path))
    out = {"device": raw["device"], "simds": raw["simds"], "source": raw_path.split("/")[-1], "load": "8 waves per SIMD on every SIMD",
           "formula": "cycles_per_instruction = clock_mhz * 1e-3 / wall_G_per_s_per_simd (w8)", "rates": {}}
    clocks = []
    for name, r in raw["rates"].items():
        w = r["w8"]
        clocks.append(w["clock_mhz"])
        out["rates"][name] = {"cycles_per_instruction": round(w["clock_mhz"] * 1e-3 / w["wall_G_per_s_per_simd"], 3),
                              "clock_mhz": w["clock_mhz"], "wall_G_per_s_per_simd": w["wall_G_per_s_per_simd"]}
    out["clock_hz"] = round(sum(clocks) / len(clocks)) * 1e6
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in out["rates"].items():
        print(f"{k:28s} {v['cycles_per_instruction']:6.2f} cycles")


def isa_counts(path):
    """{kernel key: {mnemonic: count}} over the kernel's whole text (the blend loop dominates the dynamic count)."""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_ZN3gsr\w+):", line)
        if m:
            cur = next((v for k, v in {"k_render_bwd": "render_bwd", "k_render_fwd": "render_fwd"}.items() if k in m.group(1)), None)
            if cur:
                out[cur] = defaultdict(int)
            continue
        if cur and "s_endpgm" in line:
            cur = None
        if cur:
            t = line.strip().split()
            if t and t[0].startswith("v_"):
                mn = t[0]
                if ("row_shr" in line or "row_bcast" in line or "quad_perm" in line or "row_ror" in line or "row_half_mirror" in line or "row_shl" in line) and not mn.endswith("_dpp"):
                    mn += "_dpp"
                out[cur][mn] += 1
    return out


def mix(pmc_csv, isa_path, out_path):
    per = defaultdict(lambda: defaultdict(list))
    with open(pmc_csv, newline="") as f:
        for row in csv.DictReader(f):
            for prefix, key in KERNELS.items():
                if row["Kernel_Name"].startswith(prefix):
                    per[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    isa = isa_counts(isa_path)
    out = {"workload": WORKLOAD, "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 "
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -- python3 bench.py --steps 6 --warmup 2 "
           "--no-cpu-baseline --no-secondary", "kernels": {}}
    for key, ctr in per.items():
        avg = {c: sum(v[SKIP_FIRST:]) / max(len(v[SKIP_FIRST:]), 1) for c, v in ctr.items()}
        n = isa.get(key, {})
        tot = avg.get("SQ_INSTS_VALU", 0.0)
        fma, mul, add = avg.get("SQ_INSTS_VALU_FMA_F32", 0.0), avg.get("SQ_INSTS_VALU_MUL_F32", 0.0), avg.get("SQ_INSTS_VALU_ADD_F32", 0.0)
        trans, i32, cvt = avg.get("SQ_INSTS_VALU_TRANS_F32", 0.0), avg.get("SQ_INSTS_VALU_INT32", 0.0), avg.get("SQ_INSTS_VALU_CVT", 0.0)
        other = max(tot - fma - mul - add - trans - i32 - cvt, 0.0)

        def pk_share(pk, plain):
            a, b = sum(n.get(m, 0) for m in pk), sum(n.get(m, 0) for m in plain)
            return a / (a + b) if a + b else 0.0
        s_fma = pk_share(["v_pk_fma_f32"], ["v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_fma_f32_dpp"])
        s_mul = pk_share(["v_pk_mul_f32"], ["v_mul_f32", "v_mul_f32_e32", "v_mul_f32_e64"])
        plain_add = ["v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_f32_e32", "v_sub_f32_e32", "v_subrev_f32_e32", "v_add_f32_e64", "v_sub_f32_e64"]
        n_dpp = n.get("v_add_f32_dpp", 0)
        n_add_all = n.get("v_pk_add_f32", 0) + sum(n.get(m, 0) for m in plain_add) + n_dpp
        s_add = n.get("v_pk_add_f32", 0) / n_add_all if n_add_all else 0.0
        s_dpp = n_dpp / n_add_all if n_add_all else 0.0       # the wave reduction's DPP adds are counted as ADD_F32 too
        movs = sum(c for m, c in n.items() if m.startswith("v_mov") or m.startswith("v_accvgpr"))
        others_static = sum(c for m, c in n.items() if re.match(r"v_(cmp|cndmask|min|max|med3|mov|readlane|readfirstlane|and|or|lshl|bfe|add_f32_dpp)", m)) or 1
        s_mov = min(movs / others_static, 1.0)
        perm = sum(c for m, c in n.items() if m.startswith("v_permlane32_swap") or m.startswith("v_permlane16_swap"))       # the butterfly reduction's swaps: 8.2 cycles
        s_perm = min(perm / (others_static + perm), 1.0)
        counts = {
            "v_pk_fma_f32": fma * s_fma, "v_fma_f32": fma * (1 - s_fma), "v_pk_mul_f32": mul * s_mul, "v_mul_f32": mul * (1 - s_mul),
            "v_pk_add_f32": add * s_add, "v_add_f32_dpp": add * s_dpp, "v_add_f32": add * (1 - s_add - s_dpp), "v_exp_f32": trans, "v_mov_b32": other * (1 - s_perm) * s_mov + i32 + cvt,
            "v_permlane32_swap_b32": other * s_perm,
            "v_cndmask_b32(sgpr)": other * (1 - s_perm) * (1 - s_mov),       # compares, selects, min/max: 4.2-cycle class
        }
        out["kernels"][key] = {"insts_valu": tot, "launches_sampled": len(ctr.get("SQ_INSTS_VALU", [])) - SKIP_FIRST,
                               "pmc_per_launch": {k: round(v) for k, v in avg.items()},
                               "static_packed_share": {"fma": round(s_fma, 3), "mul": round(s_mul, 3), "add": round(s_add, 3), "dpp_of_add": round(s_dpp, 3), "mov_of_other": round(s_mov, 3), "permlane_of_other": round(s_perm, 3)},
                               "class_share": {k: round(v / tot, 4) if tot else 0.0 for k, v in counts.items()}}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "rates":
        rates(sys.argv[2], sys.argv[3])
    else:
        if len(sys.argv) > 5:
            WORKLOAD = sys.argv[5]
        mix(sys.argv[2], sys.argv[3], sys.argv[4])

#!/usr/bin/env python3
"""Issue-floor model of the layer kernels: launch time predicted from the DYNAMIC instruction classes of a launch
(rocprofv3 --pmc, profiles/pmc_classes.sh) x the issue prices of profiles/microbench/r03_valu_cost.txt / r03_shadow.txt, at the clock
the kernel actually ran at (GRBM_GUI_ACTIVE / 8 XCDs / duration of the same dispatch), against the measured launch time.

    python profiles/issue_floor.py gpurun_out/<tag>/classes [bench.json [traffic.json]]  ->  markdown table on stdout; with a third
    argument the per-kernel clock, predicted and measured launch times and their ratio are also written INTO that hash-stamped file
    (key "issue_floor"), which is where bench.py's roofline.issue_floor_ratio / roofline.clock_GHz come from

Model (profiles/microbench/README.md): on gfx950 the VALU and MFMA instructions of the waves of one SIMD issue one after the
other - a v_mfma_f32_16x16x32_f16 hides nothing (20.8 cycles with or without VALU around it, r03_shadow.txt) - so a SIMD's time is the SUM
of the issue prices of everything its waves issue. Prices in units of one v_fma_f32 (1.396 ns per instruction and SIMD at three waves
per SIMD = 3.35 cycles at the 2.4 GHz the micro-benchmarks ran at):
    plain fp32 / integer ALU 1.0 | packed fp32, v_med3, v_cvt_pk, DPP, v_readlane, 64-bit integer 1.38 | v_exp / v_rcp / v_sqrt 2.55 |
    v_fma_mix{lo,hi}_f16 2.46 | v_mfma_f32_16x16x32_f16 6.2 (20.8 cycles) | v_mfma_f32_16x16x4_f32 10.7 (35.8 cycles)
The counters give CLASSES, not opcodes; the packed share inside the fp32 classes and the composition of the unclassified rest (v_mov,
v_med3, v_cndmask, v_max, DPP / permlane) come from the static listing of the shipped kernels (profiles/dev/isa_by_line.py) and are
stated below - they move the result by a few per cent, not more.
"""
import collections
import csv
import glob
import json
import re
import sys

UNIT_CYCLES = 1.396 * 2.4          # one issue unit = 3.35 cycles (micro-benchmark clock 2.4 GHz)
N_SIMD = 1024
PRICE = {"SQ_INSTS_VALU_TRANS_F32": 2.55, "SQ_INSTS_VALU_FMA_F16": 2.46, "SQ_INSTS_VALU_CVT": 1.36, "SQ_INSTS_VALU_INT64": 1.40,
         "SQ_INSTS_VALU_INT32": 1.05,      # v_add_u32 0.93 ... v_mul_lo / v_mad_u32_u24 1.38 (about a quarter of the class)
         "SQ_INSTS_VALU_ADD_F32": 1.10,    # v_add_f32 / v_pk_add_f32 (a quarter packed)
         "SQ_INSTS_VALU_MUL_F32": 1.15,    # 24 v_mul_f32 + 16 v_pk_mul_f32 per tile
         "SQ_INSTS_VALU_FMA_F32": 1.26,    # ~120 v_pk_fma_f32 + ~58 v_fma / v_fmac per tile
         "other": 1.20,                    # per tile ~64 v_med3 (1.38), ~60 v_mov / v_max / v_cmp (1.0), ~20 DPP / permlane / readlane (1.38), cndmask
         "SQ_INSTS_VALU_MFMA_F16": 20.8 / UNIT_CYCLES, "SQ_INSTS_VALU_MFMA_F32": 35.8 / UNIT_CYCLES}
CLASSES = [c for c in PRICE if c.startswith("SQ_INSTS_VALU_") and "MFMA" not in c]


def load(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    dur = collections.defaultdict(list)
    for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"k_edge<(\d+), (\d+), (?:true|false), \d+, (\d+)()>", row["Kernel_Name"])
            if not m:
                continue
            k = (int(m.group(1)), int(m.group(3)))
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[k].append((float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    res = {k: {c: v / cnt[(k, c)] for c, v in d.items()} for k, d in agg.items()}
    for k in res:
        # clock of the launch: busy cycles summed over the 8 XCDs / 8 / duration of the same dispatch (ns)
        res[k]["clock_GHz"] = sum(g for g, _ in dur[k]) / 8.0 / sum(t for _, t in dur[k])
        res[k]["pmc_pass_us"] = sum(t for _, t in dur[k]) / len(dur[k]) / 1e3
    return res


def main():
    res = load(sys.argv[1])
    meas = {}
    if len(sys.argv) > 2:
        b = json.load(open(sys.argv[2]))
        for nn in (8, 16, 32, 64):
            meas[nn] = b["whole_forward"]["kernels"][f"edge_nn{nn}"]["avg_launch_ms"] * 1e3
    n1 = 24001
    summary = {}
    print("| kernel | tiles | VALU / tile | MFMA f16 + f32 / tile | issue units / tile (VALU + MFMA) | clock under the kernel | predicted launch | measured (HIP events, same call) | measured / predicted |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (nn, ne) in sorted(res):
        d = res[(nn, ne)]
        tiles = n1 * nn / 16.0
        valu_units = sum(d.get(c, 0.0) * PRICE[c] for c in CLASSES)
        other = d["SQ_INSTS_VALU"] - sum(d.get(c, 0.0) for c in CLASSES)
        valu_units += other * PRICE["other"]
        mfma_units = d["SQ_INSTS_VALU_MFMA_F16"] * PRICE["SQ_INSTS_VALU_MFMA_F16"] + d["SQ_INSTS_VALU_MFMA_F32"] * PRICE["SQ_INSTS_VALU_MFMA_F32"]
        units = valu_units + mfma_units
        t_pred = units / N_SIMD * UNIT_CYCLES / (d["clock_GHz"] * 1e3)          # us
        t_meas = meas.get(nn)
        summary[str(nn)] = {"item_waves": ne, "clock_GHz": d["clock_GHz"], "issue_units_per_tile": units / tiles, "predicted_us": t_pred,
                            "measured_us": t_meas, "measured_over_predicted": (t_meas / t_pred) if t_meas else None}
        print(f"| `k_edge<{nn}>` ({ne} item waves) | {tiles:,.0f} | {d['SQ_INSTS_VALU'] / tiles:.0f} | {d['SQ_INSTS_VALU_MFMA_F16'] / tiles:.1f} + {d['SQ_INSTS_VALU_MFMA_F32'] / tiles:.1f} | "
              f"{units / tiles:,.0f} ({valu_units / tiles:,.0f} + {mfma_units / tiles:,.0f}) | {d['clock_GHz']:.2f} GHz | {t_pred:.1f} us | "
              + (f"{t_meas:.1f} us | {t_meas / t_pred:.2f} |" if t_meas else "- | - |"))
    if len(sys.argv) > 3:
        tf = json.load(open(sys.argv[3]))
        tf["issue_floor"] = {"per_nn": summary, "how": "profiles/issue_floor.py: dynamic instruction classes of a launch (rocprofv3 --pmc, profiles/pmc_classes.sh) x "
                             "the micro-benchmark price list, at the clock of the same dispatch (GRBM_GUI_ACTIVE / 8 XCDs / duration), against "
                             "the launch time by HIP events in the same gpurun call"}
        json.dump(tf, open(sys.argv[3], "w"), indent=1)
    print()
    for (nn, ne) in sorted(res):
        d = res[(nn, ne)]
        tiles = n1 * nn / 16.0
        print(f"k_edge<{nn}> per tile: " + " ".join(f"{c.replace('SQ_INSTS_VALU_', '')}={d.get(c, 0.0) / tiles:.1f}" for c in CLASSES + ['SQ_INSTS_VALU_MFMA_F16', 'SQ_INSTS_VALU_MFMA_F32'])
              + f" other={(d['SQ_INSTS_VALU'] - sum(d.get(c, 0.0) for c in CLASSES)) / tiles:.1f}")


if __name__ == "__main__":
    main()

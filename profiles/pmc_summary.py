"""Per-kernel summary of the rocprofv3 --pmc passes profiles/pmc_collect.sh left in a directory: python profiles/pmc_summary.py gpurun_out/pmc_<tag>"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        m = re.search(r"k_edge<([^>]*)>", name)
        k = "k_edge<" + m.group(1).replace(" ", "") + ">" if m else re.sub(r"\(.*", "", name)[-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    if "k_edge" not in k and "k_node" not in k:
        continue
    d = {c: v / cnt[(k, c)] for c, v in agg[k].items()}
    if "SQ_WAVE_CYCLES" not in d:
        continue
    wc = d["SQ_WAVE_CYCLES"]
    g = lambda c: d.get(c, 0.0)
    print(f"{k}   ({cnt[(k, 'SQ_WAVE_CYCLES')]} dispatches)")
    print(f"   per wave-cycle: active {100 * g('SQ_ACTIVE_INST_ANY') / wc:.1f} %  issue-stalled {100 * g('SQ_WAIT_INST_ANY') / wc:.1f} %  waiting (s_waitcnt / barrier) {100 * g('SQ_WAIT_ANY') / wc:.1f} %  "
          f"LDS-issue-stalled {100 * g('SQ_WAIT_INST_LDS') / wc:.1f} %")
    print(f"   instructions: VALU {g('SQ_INSTS_VALU') / 1e6:.2f} M  MFMA {g('SQ_INSTS_MFMA') / 1e6:.2f} M  (VALU / MFMA {g('SQ_INSTS_VALU') / max(g('SQ_INSTS_MFMA'), 1):.2f})  LDS {g('SQ_INSTS_LDS') / 1e6:.2f} M  "
          f"VMEM rd {g('SQ_INSTS_VMEM_RD') / 1e6:.2f} M")
    print(f"   MFMA busy {100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * wc):.1f} % of wave-cycles x 4  LDS bank conflicts {100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.1f} % of LDS-active cycles  "
          f"wave-cycles {wc / 1e6:.1f} M  GRBM_GUI_ACTIVE {g('GRBM_GUI_ACTIVE') / 1e6:.2f} M  FETCH {g('FETCH_SIZE') / 1024:.1f} MiB  WRITE {g('WRITE_SIZE') / 1024:.1f} MiB")

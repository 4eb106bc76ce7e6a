#!/bin/bash
# PMC collection for the bench workload: separate rocprofv3 --pmc passes (no tracing combined), per
# /opt/skills/guides/MI355X_MICROARCH.md (SQ 8 slots, TCC 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2).
# usage: profiles/pmc_collect.sh <tag>   (run on the GPU box from the repo root; writes gpurun_out/pmc_<tag>/)
# gpurun_out/pmc_<tag>/traffic.json is stamped with the sha256 of the kernel sources and the kernel symbols it saw; copy it to
# profiles/traffic_i_v4_1_n3000_b8.json - bench.py quotes roofline.traffic from that file only while the hash matches the tree.
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_$TAG
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency --no-extras --precision f16_split ${PESTO_BENCH_ARGS:-}"      # PESTO_BENCH_ARGS: e.g. "--edge-mode 3"
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE"
do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $SET -d $R/gpurun_out/pmc_$TAG/p$i -o pmc --output-format csv -- $CMD > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1 || echo "pass $i failed"
done
python - "$R/gpurun_out/pmc_$TAG" "$R" <<'PY'
import csv, glob, re, sys, collections
root = sys.argv[1]
sys.path.insert(0, sys.argv[2])
def key(name):      # kernel key: the whole template argument list of k_edge (the tail of the mangled name would cut it off)
    m = re.search(r"k_edge<([^>]*)>", name)
    return "k_edge<" + m.group(1).replace(" ", "") + ">" if m else re.sub(r"\(.*", "", name)[-46:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = key(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS", "TCC_HIT_sum", "FETCH_SIZE", "WRITE_SIZE"):
            cnt[(k, row["Counter_Name"])] += 1
with open(root + "/summary.txt", "w") as out:
    for k in sorted(agg):
        if "k_edge" not in k and "k_node" not in k and "k_layer" not in k:
            continue
        n = max([v for (kk, c), v in cnt.items() if kk == k] + [1])
        out.write(f"{k}  dispatches~{n}\n")
        for c in sorted(agg[k]):
            out.write(f"    {c:28s} {agg[k][c] / n:16.1f} per dispatch\n")
# per-forward HBM-side traffic of the layer kernels (bench default: 32-layer i_v4_1 -> 8 launches of each edge kernel, 33 node launches)
import json
tot_f = tot_w = 0.0
per = {}
for k in agg:
    if "k_edge" in k or "k_node" in k:
        nf = max(cnt.get((k, "FETCH_SIZE"), 1), 1); nw = max(cnt.get((k, "WRITE_SIZE"), 1), 1)
        n_fwd = max([v for (kk, c), v in cnt.items() if "k_embed" in kk and c == "FETCH_SIZE"] + [1])     # one k_embed launch per forward
        calls = round(nf / n_fwd)
        f = agg[k].get("FETCH_SIZE", 0.0) / nf * 1024.0     # rocprofv3 reports KiB
        w = agg[k].get("WRITE_SIZE", 0.0) / nw * 1024.0
        per[k.strip()] = {"fetch_bytes_per_dispatch_raw": f, "write_bytes_per_dispatch": w, "dispatches_per_forward": calls}
        tot_f += f * calls; tot_w += w * calls
from bench import source_hash
out = {"source_hash": source_hash(), "kernel_symbols": sorted(per),
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane reads); WRITE_SIZE uncorrected",
       "fetch_bytes_per_forward_raw": tot_f, "fetch_bytes_per_forward_corrected": 2 * tot_f, "write_bytes_per_forward": tot_w,
       "hbm_bytes_per_forward": 2 * tot_f + tot_w, "kernels": per}
json.dump(out, open(root + "/traffic.json", "w"), indent=1)
print(open(root + "/summary.txt").read())
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
PY

mkdir -p gpurun_out/abl
for rep in 1 2 3; do for t in ser new; do
  PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so python bench.py --batch 8 --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-check --precision f16_split > gpurun_out/abl/${t}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/abl/*.json")):
    d=json.load(open(f)); k=d["whole_forward"]["kernels"]
    print(f.split("/")[-1], "%.1f /s" % d["value"], "batch1 %.4f ms (first10 %.4f)" % (d["batch1"]["steady_state_ms"], d["batch1"]["first_10_calls_after_100ms_idle_ms"]), " ".join("%s %.1f" % (n.replace("edge_",""), v["avg_launch_ms"]*1e3) for n,v in sorted(k.items())))
PY

"""Static instruction counts of ONE kernel of an AMDGPU listing compiled with -gline-tables-only, attributed to source lines:
  hipcc ... -gline-tables-only -S pesto_edge.hip -o /tmp/layer_g.s
  python profiles/dev/isa_by_line.py /tmp/layer_g.s 'k_edgeILi8ELi12ELb1ELi1ELi8EE' [first:last[:name] ...]
Without ranges: one row per source line of file 0 (the .hip file) with >= 4 instructions; with ranges: one row per range.
(.loc gives the innermost inlined location: split8 / elu4s / row_reduce bodies show up under their own lines.)"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
ranges = []
for a in sys.argv[3:]:
    p = a.split(":")
    ranges.append((int(p[0]), int(p[1]), p[2] if len(p) > 2 else f"{p[0]}-{p[1]}"))
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5pesto") and sym in l and l.rstrip().endswith(":") or (l.startswith("_ZN5pesto") and sym in l and ": ;" in l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
cur = (0, 0)
cnt = collections.defaultdict(collections.Counter)
ops = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^([a-z_0-9]+)", s)
    if not m or s.startswith(";") or s.endswith(":") or s.startswith("."):
        continue
    op = m.group(1)
    if op.startswith("v_mfma"): c = "MFMA"
    elif op.startswith("v_"): c = "VALU"
    elif op.startswith("ds_"): c = "LDS"
    elif op.startswith(("buffer_", "global_", "scratch_", "flat_")): c = "VMEM"
    elif op.startswith(("s_waitcnt", "s_nop")): c = "wait"
    elif op.startswith("s_"): c = "SALU"
    else: continue
    cnt[cur][c] += 1
    ops[cur][re.sub(r"_e32$|_e64$", "", op)] += 1
def row(name, keys):
    t = collections.Counter(); o = collections.Counter()
    for k in keys:
        t.update(cnt[k]); o.update(ops[k])
    top = ", ".join(f"{k} {v}" for k, v in o.most_common(7) if not k.startswith("s_"))
    print(f"{name:28s} VALU {t['VALU']:5d} MFMA {t['MFMA']:4d} LDS {t['LDS']:4d} VMEM {t['VMEM']:4d} SALU {t['SALU']:4d} wait {t['wait']:4d} | {top}")
if ranges:
    used = set()
    for a, b, name in ranges:
        keys = [k for k in cnt if k[0] == 0 and a <= k[1] <= b]
        used.update(keys)
        row(name, keys)
    row("other (file 0)", [k for k in cnt if k[0] == 0 and k not in used])
    row("headers (files > 0)", [k for k in cnt if k[0] != 0])
else:
    for k in sorted(cnt):
        if sum(cnt[k].values()) >= 4:
            row(f"{k[0]}:{k[1]}", [k])
row("TOTAL", list(cnt))

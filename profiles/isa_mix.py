"""Static instruction mix of a line range of an AMDGPU .s file: python profiles/isa_mix.py file.s [first_line last_line]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
cat = collections.Counter()
ops = collections.Counter()
for l in lines[a - 1:b]:
    l = l.strip()
    m = re.match(r"^([a-z_0-9]+)", l)
    if not m or l.startswith(";") or l.endswith(":"):
        continue
    op = m.group(1)
    if op.startswith("v_mfma"):
        c = "MFMA"
    elif op.startswith("v_"):
        c = "VALU"
    elif op.startswith("ds_"):
        c = "LDS"
    elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        c = "VMEM"
    elif op.startswith("s_waitcnt"):
        c = "waitcnt"
    elif op.startswith("s_nop"):
        c = "s_nop"
    elif op.startswith("s_"):
        c = "SALU"
    else:
        continue
    cat[c] += 1
    key = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    ops[(c, key)] += 1
print("  ".join(f"{k} {v}" for k, v in cat.most_common()))
for c in ("VALU", "LDS", "VMEM", "MFMA"):
    print(f"{c}:", ", ".join(f"{k} {v}" for (cc, k), v in ops.most_common() if cc == c))

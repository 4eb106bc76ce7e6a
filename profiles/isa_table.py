"""Per-region table of the static instruction mix of an AMDGPU .s listing, VALU split by what the instruction is for:
python profiles/isa_table.py file.s name:first:last [name:first:last ...]   (line ranges of the listing)"""
import collections
import re
import sys

GROUPS = [  # (column, opcode prefixes)
    ("exp / rcp", ("v_exp_f32", "v_rcp_f32", "v_div_")),
    ("ELU med3 + scale", ("v_med3_f32",)),
    ("f16 split (cvt_pk, fma_mix)", ("v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16")),
    ("packed f32 (pk_fma / pk_mul / pk_add)", ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")),
    ("scalar f32 (fma, fmac, add, mul, sub, max)", ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_max_f32", "v_max3_f32", "v_min_f32")),
    ("cross-lane (DPP, readlane, permlane)", ("v_mov_b32_dpp", "v_add_f32_dpp", "v_max_f32_dpp", "v_readlane_b32", "v_readfirstlane_b32", "v_permlane")),
    ("address / integer", ("v_add_u32", "v_mad_i64_i32", "v_lshl", "v_ashr", "v_lshr", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mul_lo", "v_mul_u32", "v_mul_i32",
                           "v_mul_hi", "v_add3_u32", "v_sub_u32", "v_bfe", "v_mad_u32", "v_add_lshl", "v_lshl_add", "v_lshl_or", "v_and_or", "v_bitop3", "v_min_i32", "v_cmp_", "v_cndmask")),
    ("moves", ("v_mov_b32", "v_mov_b64", "v_accvgpr")),
]
lines = open(sys.argv[1]).read().split("\n")
print("| region | VALU | " + " | ".join(g for g, _ in GROUPS) + " | other VALU | MFMA | LDS (bpermute) | VMEM | SALU | waitcnt + nop |")
print("|---|---|" + "---|" * (len(GROUPS) + 6))
for spec in sys.argv[2:]:
    name, a, b = spec.rsplit(":", 2)
    c = collections.Counter()
    other = collections.Counter()
    for l in lines[int(a) - 1:int(b)]:
        l = l.strip()
        m = re.match(r"^([a-z_0-9]+)", l)
        if not m or l.startswith(";") or l.endswith(":"):
            continue
        op = m.group(1)
        full = op + ("_dpp" if " quad_perm" in l or " row_" in l else "")
        if op.startswith("v_mfma"):
            c["MFMA"] += 1
        elif op.startswith("v_"):
            c["VALU"] += 1
            key = re.sub(r"_e32$|_e64$|_sdwa$", "", full)
            for g, pre in GROUPS:
                if key.startswith(pre) and not (g == "moves" and key.endswith("_dpp")):
                    c[g] += 1
                    break
            else:
                c["other"] += 1; other[key] += 1
        elif op.startswith("ds_"):
            c["LDS"] += 1; c["bperm"] += op.startswith("ds_bpermute")
        elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
            c["VMEM"] += 1
        elif op.startswith(("s_waitcnt", "s_nop")):
            c["wait"] += 1
        elif op.startswith("s_"):
            c["SALU"] += 1
    print(f"| {name} | {c['VALU']} | " + " | ".join(str(c[g]) for g, _ in GROUPS) + f" | {c['other']} | {c['MFMA']} | {c['LDS']} ({c['bperm']}) | {c['VMEM']} | {c['SALU']} | {c['wait']} |")

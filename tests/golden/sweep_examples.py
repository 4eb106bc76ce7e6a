#!/usr/bin/env python3
"""Build-container sweep (not a test: it reads /root/reference): the native structure I/O (libpesto_io.so) against the reference's own
Python functions on EVERY input file under the reference's examples/ (30 files: channels, lipid / DNA / RNA / ion / ligand complexes,
multi-model files), and against every *_i0.pdb the reference saved next to them.
  * reader + to_dict -> the reference's clean_structure / tag_hetatm_chains / split_by_chain / filter_non_atomic_subunits /
    remove_duplicate_tagged_subunits / concatenate_chains / encode_structure / encode_features (src/structure.py,
    src/data_encoding.py) == the native preprocess() + encode(123);
  * native read -> preprocess -> save_pdb == the reference's saved output in every column but the b-factor value.
Usage: python tests/golden/sweep_examples.py [glob relative to the reference root]   (one line per file; exit code 1 on any
difference; default examples/*/*.pdb; "md_analysis/pdbs_clusters/*.pdb" = the 567 MD cluster conformations)"""
import glob
import os
import re
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.import_reference("i_v4_0_2021-09-07_11-20")     # installs the gemmi stub
    sys.path = [mg.REF] + [p for p in sys.path if "/model/save/" not in p and p != mg.REF]
    for m in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(m)
    from src.structure import (clean_structure, tag_hetatm_chains, split_by_chain, filter_non_atomic_subunits,
                               remove_duplicate_tagged_subunits, concatenate_chains)
    from src.data_encoding import encode_structure, encode_features
    from pesto_amd.structure_io import Structure
    pattern = sys.argv[1] if len(sys.argv) > 1 else os.path.join("examples", "*", "*.pdb")      # e.g. "md_analysis/pdbs_clusters/*.pdb"
    files = sorted(f for f in glob.glob(os.path.join(mg.REF, pattern)) if not re.search(r"_i\d\.pdb$", f))
    bad = 0
    for f in files:
        rel = os.path.relpath(f, mg.REF)
        try:
            st = Structure.read_pdb(f).to_dict()
            st["resid"] = st["resid"].astype(np.int32)
            s = clean_structure({k: v.copy() for k, v in st.items()})
            s = tag_hetatm_chains(s)
            sub = remove_duplicate_tagged_subunits(filter_non_atomic_subunits(split_by_chain(s)))
            s = concatenate_chains(sub)
            X, M = encode_structure(s)
            q = np.concatenate([t.numpy() for t in encode_features(s)], 1)
            nat = Structure.read_pdb(f).preprocess()
            d = nat.to_dict()
            Xn, qn, roa, R = nat.encode(123)
            ok = (np.array_equal(Xn, X.numpy()) and np.array_equal(qn, q) and R == M.shape[1] and np.array_equal(roa, M.numpy().argmax(1))
                  and all(np.array_equal(d[k], s[k]) for k in ("name", "element", "resname", "het_flag", "chain_name", "resid")))
            msg = f"N={Xn.shape[0]} R={R} chains={len(sub)}"
            out_ref = f[:-4] + "_i0.pdb"
            if os.path.exists(out_ref):
                with tempfile.TemporaryDirectory() as tmp:
                    nat.save_pdb(os.path.join(tmp, "o.pdb"))
                    strip = lambda l: l[:54] + l[66:] if l.startswith(("ATOM", "HETATM")) else l
                    same = [strip(l) for l in open(os.path.join(tmp, "o.pdb")).read().split("\n")] == [strip(l) for l in open(out_ref).read().split("\n")]
                ok = ok and same
                msg += " saved-output " + ("==" if same else "!=")
        except Exception as e:      # noqa: BLE001
            ok, msg = False, f"{type(e).__name__}: {e}"
        bad += not ok
        print(("ok   " if ok else "DIFF ") + rel + "  " + msg, flush=True)
    print(f"{len(files)} files, {bad} differing")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Timing of the native structure I/O (libpesto_io.so) next to the reference's Python functions on the same arrays.
BUILD CONTAINER ONLY: imports /root/reference (with a gemmi stub, as tests/golden/make_golden.py does) and reads its examples/.
Output committed as profiles/r01_io.txt."""
import sys, time, types, tempfile
import numpy as np
sys.path.insert(0, "/root/repo")
g = types.ModuleType("gemmi"); g.cif = types.ModuleType("gemmi.cif"); sys.modules["gemmi"] = g; sys.modules["gemmi.cif"] = g.cif
sys.path.insert(0, "/root/reference")
from src.structure import clean_structure, tag_hetatm_chains, split_by_chain, filter_non_atomic_subunits, remove_duplicate_tagged_subunits, concatenate_chains, encode_bfactor
from src.data_encoding import encode_structure, encode_features
from src.structure_io import save_pdb
from pesto_amd.structure_io import Structure
for rel in ("lipids/7KHT_lipid", "channel/6Y5B", "channel/5JZT"):
    f = f"/root/reference/examples/{rel}.pdb"
    t0 = time.perf_counter(); s = Structure.read_pdb(f); t_read = time.perf_counter() - t0
    st = s.to_dict(); st["resid"] = st["resid"].astype(np.int32)
    t0 = time.perf_counter(); s.preprocess(); t_pre = time.perf_counter() - t0
    t0 = time.perf_counter(); X, q, roa, R = s.encode(30); t_enc = time.perf_counter() - t0
    p = np.random.rand(R).astype(np.float32)
    tmp = tempfile.mkdtemp()
    t0 = time.perf_counter(); s.save_pdb(tmp + "/a.pdb", p); t_save = time.perf_counter() - t0
    # reference python on the same dict
    t0 = time.perf_counter()
    r = clean_structure(st); r = tag_hetatm_chains(r); su = split_by_chain(r); su = filter_non_atomic_subunits(su); su = remove_duplicate_tagged_subunits(su); r = concatenate_chains(su)
    rt_pre = time.perf_counter() - t0
    t0 = time.perf_counter(); Xr, M = encode_structure(r); qr = encode_features(r)[0]; rt_enc = time.perf_counter() - t0
    t0 = time.perf_counter(); r = encode_bfactor(r, p); save_pdb(split_by_chain(r), tmp + "/b.pdb"); rt_save = time.perf_counter() - t0
    assert open(tmp + "/a.pdb").read() == open(tmp + "/b.pdb").read()
    print(f"{rel}: {len(st['xyz'])} atoms read, {len(s)} kept | native read {t_read*1e3:.1f} ms, preprocess {t_pre*1e3:.1f}, encode {t_enc*1e3:.1f}, bfactor+save {t_save*1e3:.1f} | reference python preprocess {rt_pre*1e3:.1f} ms, encode {rt_enc*1e3:.1f}, bfactor+save {rt_save*1e3:.1f}")

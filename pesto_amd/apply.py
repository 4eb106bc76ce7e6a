"""Bulk form of the reference's inference loop (apply_model.ipynb cell 6, interfaceome/apply_model.py:57-82):
PDB files in, per-residue interface probabilities (and b-factor PDB files) out.

The reference runs one structure at a time behind a DataLoader with 8 worker processes. Here the host stages (native read /
clean / encode, native write; both release the GIL inside libpesto_io.so) run in a thread pool, and the GPU stage takes several
structures per launch: collated coordinates -> GPU k-NN (pesto_knn_collate) -> forward -> GPU sigmoid + b-factor expansion.
Structures that fail to parse are reported and skipped, as the reference's try/except does.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .structure_io import PestoIOError, Structure


def _load(path, n0):
    s = Structure.read_pdb(path).preprocess()
    X, q, roa, R = s.encode(n0)
    return s, X, q, roa, R


def _is_h5(path):
    return os.fspath(path).lower().endswith((".h5", ".hdf5", ".hdf"))


def h5_dataset_names(result_keys, keys=None):
    """{result key -> HDF5 dataset name} as save_results uses it: ``keys`` where given, else the key without leading '/'. Two result keys
    with the same dataset name ('/a.pdb' and 'a.pdb') raise ValueError - found here, before any work, not in the middle of a write."""
    out, seen = {}, {}
    for k in result_keys:
        n = (keys or {}).get(k, str(k).lstrip("/"))
        if not n:
            raise ValueError(f"result key {k!r} has an empty HDF5 dataset name")
        if n in seen:
            raise ValueError(f"result keys {seen[n]!r} and {k!r} map to the same HDF5 dataset name {n!r}: pass keys={{key: name}}")
        seen[n] = k
        out[k] = n
    return out


def check_results_path(path, result_keys, keys=None):
    """What can be known about a bulk result file before the work is done: for an HDF5 name, that the HDF5 C library loads here
    (h5store.H5Unavailable otherwise) and that the dataset names are unique (ValueError). apply_model calls it before its first launch:
    a run of hours must not end in an exception that loses the computed results."""
    if path is not None and _is_h5(path):
        from . import h5store
        h5store.load()
        h5_dataset_names(list(result_keys), keys)


def save_results(results, path, keys=None):
    """Bulk result file of the reference's interfaceome driver, ``hf[key] = p.cpu().numpy()`` per structure
    (interfaceome/apply_model.py:53-79). The extension picks the container:

    * ``.h5`` / ``.hdf5``: an HDF5 file like the reference's - one float32 dataset [R, n_out] per key, groups along the ``/`` of a key
      (h5store.H5Store: the HDF5 C library through ctypes, no h5py; raises h5store.H5Unavailable when the machine has no libhdf5 -
      nothing is written in another format behind the caller's back). ``keys``: {result key -> dataset name}; default: the path
      without its leading '/' (a file path is a valid HDF5 name; the reference's keys are its store's ``pdbid/assembly/chain`` names).
    * anything else: one .npz with the per-structure tables stacked along the residue axis: ``keys`` [n] (str), ``offsets`` [n+1]
      (int64), ``p`` [sum R_i, n_out] (float32) - structure i is p[offsets[i]:offsets[i+1]].
    Both are written to a temporary file and renamed; a failed write removes its temporary file. An HDF5 store is read back under its
    DATASET names (load_results: the default name drops the leading '/' of a path key - h5_dataset_names gives the map)."""
    path = os.fspath(path)
    names = list(results)
    tabs = [np.asarray(results[k], dtype=np.float32).reshape(len(results[k]), -1) for k in names]
    if _is_h5(path):
        from . import h5store
        dsn = h5_dataset_names(names, keys)      # (raises on two keys with one dataset name BEFORE anything is written)
        tmp = path + ".tmp"
        try:
            with h5store.H5Store(tmp, "w") as hf:
                for k, t in zip(names, tabs):
                    hf[dsn[k]] = t
            os.replace(tmp, path)
        except BaseException:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise
        return path
    n_out = tabs[0].shape[1] if tabs else 0
    offs = np.zeros(len(names) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([t.shape[0] for t in tabs])
    p = np.concatenate(tabs, 0) if tabs else np.zeros((0, n_out), np.float32)
    tmp = path + ".tmp.npz"
    np.savez_compressed(tmp, keys=np.array(names, dtype=str), offsets=offs, p=p)
    os.replace(tmp, path)
    return path


def load_results(path):
    """{key: p [R, n_out]} from a file written by save_results (or, for .h5, by the reference's own loop). An HDF5 store gives its
    dataset names as keys (h5_dataset_names: a path key comes back without its leading '/'); the .npz form the keys as they were."""
    if _is_h5(path):
        from . import h5store
        with h5store.H5Store(path) as hf:
            return dict(hf.items())
    d = np.load(path)
    offs, p, keys = d["offsets"], d["p"], d["keys"]      # NpzFile decompresses an array on EVERY access: materialise each once
    return {str(k): p[offs[i]:offs[i + 1]] for i, k in enumerate(keys)}


def apply_model(model, pdb_filepaths, write=True, suffix="_i{}.pdb", max_atoms=24576, workers=8, on_error=print, results_path=None,
                report_ties=True):
    """Returns {path: p} with p = sigmoid(z) as numpy [R, n_out] for every structure that could be processed.
    write=True also saves ``path[:-4] + suffix.format(i)`` for each output channel i (apply_model.ipynb:157-166).
    results_path: also write all probability tables into one bulk result file (save_results): ``*.h5`` = the reference's HDF5 store
    (one dataset per structure), otherwise the .npz form.
    ``model``: a pesto_amd.Model on a GPU; max_atoms: atoms per launch (about 24k fills an MI355X).
    report_ties: log (logging "pesto_amd.apply", WARNING) the structures in which two neighbours at exactly the same float32 distance
    straddle a layer's neighbourhood cut-off - the rows on which the reference's torch.topk (src/data_encoding.py:98-99) may have chosen
    the other atom, the one known source of logit differences beyond 1e-4 against the reference (pesto_knn_tie_rows)."""
    import torch
    pdb_filepaths = list(pdb_filepaths)
    check_results_path(results_path, pdb_filepaths)      # no HDF5 library / colliding dataset names: raised now, not after the run
    n0 = model.config["em"]["N0"]
    dev = torch.device("cuda", model._gpu)
    results = {}
    # launches stay asynchronous (pesto_set_async_auto): the post-op of a launch - the next call on the handle - makes its deferred range /
    # input check, so the host packs, slices and writes while the GPU computes
    was_async = model.async_auto
    model.set_async_auto(True)
    try:
        _apply_loop(model, pdb_filepaths, write, suffix, max_atoms, workers, on_error, results, dev, n0, report_ties)
    except BaseException:
        # the loop failed: still drain the handle and restore its mode, but the deferred check of the last launch (model.synchronize() may
        # raise its bad-input / range error) must not replace the exception that is already in flight
        try:
            model.synchronize()
        except Exception:
            pass
        finally:
            model.set_async_auto(was_async)
        raise
    else:
        try:
            model.synchronize()
        finally:
            model.set_async_auto(was_async)
    if results_path is not None:
        try:
            save_results(results, results_path)
        except Exception as e:      # (checked up front; what is left is the file system) - the computed tables are not lost with it
            e.results = results
            raise
    return results


def _apply_loop(model, pdb_filepaths, write, suffix, max_atoms, workers, on_error, results, dev, n0, report_ties):
    import logging
    log = logging.getLogger("pesto_amd.apply")
    used_cuts = sorted({int(l["nn"]) for l in model.config["sum"]})
    cut_mask = sum(1 << {8: 0, 16: 1, 32: 2, 64: 3}[c] for c in used_cuts)
    import torch
    with ThreadPoolExecutor(max_workers=workers) as pool:
        loads = [(p, pool.submit(_load, p, n0)) for p in pdb_filepaths]
        writes = []

        # One launch = pack (host) -> H2D, GPU k-NN, forward (queued, asynchronous) -> fetch (post-op, D2H) -> hand out (slice, b-factor
        # files to the pool). The launch in flight is fetched AFTER the next one has been packed and handed out after the next one
        # has been queued, so that the host's packing, slicing and file writing run while the GPU computes (profiles/r03_bulk_breakdown.txt: the loop is GPU-bound, packing was the part of the
        # host's share that could hide). The post-op is the first consumer of the logits: under precision "auto" it is also where
        # the deferred range / input check of the forward is made, so it belongs to the collecting half.
        in_flight = []      # at most one (group, sizes, r_off, z, roa) whose results are still on the device

        def fetch():
            """post-op + D2H of the launch in flight -> host arrays (None when nothing is in flight)"""
            if not in_flight:
                return None
            group, sizes, r_off, z, roa, ties = in_flight.pop()
            p, bf = model.postprocess(z, roa)
            return group, sizes, r_off, p.cpu().numpy(), bf.cpu().numpy(), (ties.cpu().numpy() if ties is not None else None)

        def hand_out(done):
            """slice a fetched launch per structure, queue its b-factor files (host only: runs while the GPU computes the next launch)"""
            if done is None:
                return
            group, sizes, r_off, p, bf, ties = done
            a_off = np.cumsum([0] + sizes)
            for i, (path, s, *_rest) in enumerate(group):
                results[path] = p[r_off[i]:r_off[i + 1]]
                if ties is not None:
                    n_t = int(np.count_nonzero(ties[a_off[i]:a_off[i + 1]] & cut_mask))
                    if n_t:
                        log.warning("%s: %d atom(s) with an exact distance tie across a neighbourhood cut-off (nn in %s): the reference's "
                                    "torch.topk may pick the other atom there", path, n_t, used_cuts)
                if write:
                    for c in range(bf.shape[0]):
                        out = path[:-4] + suffix.format(c)
                        writes.append(pool.submit(s.save_pdb, out, np.ascontiguousarray(bf[c, a_off[i]:a_off[i + 1]])))

        def flush(group):
            if not group:
                return
            sizes = [len(g[1]) for g in group]
            Xh = np.concatenate([g[2] for g in group])
            qh = np.concatenate([g[3] for g in group])
            r_off = np.cumsum([0] + [g[5] for g in group])
            rh = np.concatenate([g[4] + r_off[i] for i, g in enumerate(group)]).astype(np.int32)
            done = fetch()                                                              # the previous launch (the GPU had the packing time)
            X, q, roa = torch.from_numpy(Xh).to(dev), torch.from_numpy(qh).to(dev), torch.from_numpy(rh).to(dev)
            ids = model.knn_collate(X, sizes)
            ties = model.knn_tie_rows(X, sizes, ids) if report_ties else None
            z = model.forward_segments(X, ids, q, roa, int(r_off[-1]), sizes=sizes)    # one call per structure, semantically
            in_flight.append((group, sizes, r_off, z, roa, ties))
            hand_out(done)

        group, atoms = [], 0
        pad_cols = model.max_nn         # (neighbour slots a layer reads)
        small, small_atoms = [], 0      # structures of fewer atoms than that (zero-padded neighbour slots): launches of their own - precision "auto"
                                        # repeats them on the exact kernels, and that repeat covers the whole launch (sharding.forward_local)
        for path, fut in loads:
            try:
                s, X, q, roa, R = fut.result()
            except (PestoIOError, OSError) as e:
                if on_error:
                    on_error(f"error with {path}: {e}")
                continue
            if len(s) < pad_cols:
                if small and small_atoms + len(s) > max_atoms:
                    flush(small)
                    small, small_atoms = [], 0
                small.append((path, s, X, q, roa, R))
                small_atoms += len(s)
                continue
            if group and atoms + len(s) > max_atoms:
                flush(group)
                group, atoms = [], 0
            group.append((path, s, X, q, roa, R))
            atoms += len(s)
        flush(group)
        flush(small)
        hand_out(fetch())
        for w in writes:
            w.result()

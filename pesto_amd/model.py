"""Host mirror of the reference's ``Model`` (reference model/model.py:6-52) over the HIP C ABI.

Drop-in usage, as in apply_model.ipynb:73-93,155:

    from pesto_amd import Model, config_model
    model = Model(config_model)
    model.load_state_dict(torch.load(".../model_ckpt.pt", map_location="cpu"))
    model = model.eval().to(device)
    z = model(X, ids_topk, q, M.float())          # fp32 [R, N2], same kind/device as the inputs

All arithmetic runs in libpesto_hip.so on the MI355X; there is no PyTorch or CPU fallback. ``device`` only
says where the caller's tensors live: CUDA(ROCm) tensors are consumed in place on torch's current stream,
CPU tensors / numpy arrays are staged through the library's own buffers.

``precision`` (no reference counterpart - torch computes in fp32): "auto" (default) runs the state-update GEMMs as f16 hi/lo
split MFMA and computes every STRUCTURE in which an activation left the f16 range again on the exact fp32 kernels (the other
structures of the launch keep their logits: results do not depend on the grouping); "f16_split" never repeats (such a structure
raises / returns NaN); "fp32" always uses the exact fp32 MFMA kernels (enum pesto_precision).
``async_auto`` (default False): with ROCm tensors and precision "auto", ``model(...)`` returns once the launch's range / input check
has been read back (a 4-byte copy queued in front of the pool kernels; the call waits for that copy, not for the stream): bad inputs
have raised, an overflowed structure has been queued again on the exact kernels, and ``z`` is final in STREAM ORDER on torch's current
stream - what a drop-in caller that goes on with ``torch.sigmoid(z)`` needs (torch keeps the tensors alive and ordered; a caller that
reads ``z`` from another stream or overwrites an input from the host waits for the stream first). True makes the call fully asynchronous: the check is made by the NEXT call on the model (or
``synchronize()``), which may write the fp32 repeat into the same ``z``; until then an overflowed structure holds NaN.
"""
import ctypes

import numpy as np

from . import _lib
from .config import normalise
from .topology import mask_to_segments
from .weights import blob_schema, flatten_state_dict, unflatten_blob


def _is_torch(x):
    return hasattr(x, "detach") and hasattr(x, "device")


class Model:
    def __init__(self, config, device=None, validate=True, precision="auto", async_auto=False):
        self.config = normalise(config)
        self.precision = str(precision).lower()
        self.async_auto = bool(async_auto)
        self._cc = _lib.make_c_config(self.config, self.precision)
        self._debug = (0, 0)        # pesto_debug_select(layer_kernels, knn_brute_force): test hook
        self._edge_mode = 0         # pesto_debug_edge_mode: test hook
        self._auto_bill_logged = False
        self._state_limit = None    # pesto_set_auto_state_limit (None: library default)
        self._pad_trigger = None    # pesto_set_auto_pad_trigger (None: library default = on)
        self._last_device_call = None           # async_auto: tensors of the last device call (kept alive for its deferred check)
        self._blob = None
        self._handle = None
        self._gpu = 0
        self._io_device = None      # torch.device the caller asked for with .to(); informational
        self.validate = validate    # check M (one residue per atom, none empty) on every call
        self.training = False
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ torch.nn.Module-like surface
    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's state_dict (torch tensors or numpy arrays). ``strict=True`` (torch's default): missing AND
        unexpected keys raise KeyError. ``strict=False`` ignores unexpected keys only - every parameter of the architecture must
        still be present, because a handle cannot be built from a partial blob (torch would keep its random initialisation)."""
        self._blob = np.ascontiguousarray(flatten_state_dict(self.config, state_dict, strict=strict), dtype=np.float32)
        self._release()
        return "<All keys matched successfully>"

    def load_blob(self, blob):
        """The weights as the flat float32 blob pesto_create takes (pesto_amd.weights.flatten_state_dict order: the reference's state_dict
        without m_nn / sdk) - what a rank receives when rank 0 broadcasts the model (sharding.broadcast_weights; SURVEY 8e: ncclBroadcast)."""
        from .weights import blob_size
        b = np.ascontiguousarray(blob.detach().cpu().numpy() if _is_torch(blob) else blob, dtype=np.float32).ravel()
        if b.size != blob_size(self.config):
            raise ValueError(f"weight blob has {b.size} values, the configuration needs {blob_size(self.config)}")
        self._blob = b
        self._release()
        return self

    def blob(self):
        """The flat float32 weight blob of the loaded state (see load_blob)."""
        if self._blob is None:
            raise RuntimeError("load_state_dict() / load_blob() must be called first")
        return self._blob

    def set_precision(self, precision):
        """"auto" | "f16_split" | "fp32" (pesto_set_precision); takes effect with the next call."""
        code = _lib.precision_code(precision)
        self.precision = str(precision).lower()
        self._cc.precision = code
        if self._handle is not None:
            _lib.check(_lib.load().pesto_set_precision(self._handle, code))
        return self

    def set_async_auto(self, enabled=True):
        """pesto_set_async_auto: precision "auto" with ROCm tensors returns without synchronising; the launch is checked by the next
        call on the model (see the module docstring). Callers that consume z through postprocess() / a later model call, like
        apply.apply_model, switch this on."""
        self.async_auto = bool(enabled)
        if self._handle is not None:
            _lib.check(_lib.load().pesto_set_async_auto(self._handle, 1 if self.async_auto else 0))
        if not self.async_auto:
            self._last_device_call = None
        return self

    def set_auto_state_limit(self, limit):
        """pesto_set_auto_state_limit: precision "auto" repeats a structure on the exact fp32 kernels once its state magnitude exceeds
        `limit` in any layer (the split kernels' error is relative to the states); <= 0 switches the trigger off. None = library default."""
        self._state_limit = None if limit is None else float(limit)
        if self._handle is not None and self._state_limit is not None:
            _lib.check(_lib.load().pesto_set_auto_state_limit(self._handle, self._state_limit))
        return self

    def set_auto_pad_trigger(self, enabled=True):
        """pesto_set_auto_pad_trigger: precision "auto" repeats structures with zero-padded neighbour slots (fewer than 64 atoms / columns) on
        the exact fp32 kernels - the inputs on which the forward is ill-conditioned for any fp32 evaluation. On by default."""
        self._pad_trigger = bool(enabled)
        if self._handle is not None:
            _lib.check(_lib.load().pesto_set_auto_pad_trigger(self._handle, 1 if self._pad_trigger else 0))
        return self

    def status(self):
        """{"precision", "n_forward", "n_fp32_rerun"}: launch sequences run by this handle and how many "auto" repeated on the
        exact fp32 kernels (pesto_get_status)."""
        p, a, b = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.load().pesto_get_status(self._ensure(), ctypes.byref(p), ctypes.byref(a), ctypes.byref(b)))
        return {"precision": {v: k for k, v in _lib.PRECISIONS.items()}[p.value], "n_forward": a.value, "n_fp32_rerun": b.value}

    def auto_counters(self):
        """{"n_structures", "n_repeated"}: structures forwarded on the split kernels under precision "auto" and how many of them were
        computed again on the exact fp32 kernels (pesto_get_auto_counters)."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.load().pesto_get_auto_counters(self._ensure(), ctypes.byref(a), ctypes.byref(b)))
        return {"n_structures": a.value, "n_repeated": b.value}

    def _tick_auto_bill(self, every=16):
        """called behind a forward: looks at the counters every ``every``-th call until the warning has been given"""
        if self._auto_bill_logged:
            return
        self._auto_calls = getattr(self, "_auto_calls", 0) + 1
        if self._auto_calls % every == 0:
            self._note_auto_bill()

    def _note_auto_bill(self):
        """Once per model: a WARNING (logging "pesto_amd") when "auto" has repeated at least 90 % of >= 16 structures - such a model (the
        reference's trained i_v3_1) pays the split AND the exact kernels on every structure and is ~1.4x faster under precision="fp32".
        Never called where it would synchronise an asynchronous launch (async_auto)."""
        if self._auto_bill_logged or self.precision != "auto" or self.async_auto or self._handle is None:
            return
        c = self.auto_counters()
        if c["n_structures"] >= 16 and c["n_repeated"] >= 0.9 * c["n_structures"]:
            import logging
            logging.getLogger("pesto_amd").warning(
                "precision='auto' has repeated %d of %d structures on the exact fp32 kernels (an activation left the f16 range, or a "
                "neighbour table is zero-padded): this model pays the split AND the exact kernels on every call - construct it with "
                "precision='fp32' (Model(config, precision='fp32') / set_precision('fp32'))", c["n_repeated"], c["n_structures"])
            self._auto_bill_logged = True

    def debug_select(self, layer_kernels=0, knn_brute_force=False):
        """Test hook (pesto_debug_select): 0 = shipped kernels, 1 = fp32 VALU reference-formulation kernel; brute-force k-NN."""
        self._debug = (int(layer_kernels), int(bool(knn_brute_force)))
        if self._handle is not None:
            _lib.check(_lib.load().pesto_debug_select(self._handle, *self._debug))
        return self

    def debug_host_only(self, enabled=True):
        """Measurement hook (pesto_debug_host_only): forward_batch_submit does its host half only; the wait returns zeros."""
        _lib.check(_lib.load().pesto_debug_host_only(self._ensure(), 1 if enabled else 0))
        return self

    @property
    def max_nn(self):
        """The largest neighbourhood a layer gathers from: a structure with fewer atoms (or a neighbour table with fewer columns) has
        zero-padded slots a layer reads - what precision "auto" repeats on the exact kernels (pesto_set_auto_pad_trigger)."""
        return max(int(l["nn"]) for l in self.config["sum"])

    def debug_edge_mode(self, mode=0):
        """Test hook (pesto_debug_edge_mode): 0 = work decomposition of the state-update kernel chosen per launch, 1 = rendezvous
        mode, 2 = node-wave mode. Results must not depend on it."""
        self._edge_mode = int(mode)
        if self._handle is not None:
            _lib.check(_lib.load().pesto_debug_edge_mode(self._handle, self._edge_mode))
        return self

    def state_dict(self):
        if self._blob is None:
            raise RuntimeError("no weights loaded")
        return unflatten_blob(self.config, self._blob)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("pesto_amd.Model is inference-only (the reference's training loop is out of scope)")
        return self

    def to(self, device):
        """``device``: 'cpu', 'cuda', 'cuda:1', torch.device, or an int GPU ordinal. 'cpu' keeps computing on GPU 0."""
        idx = None
        if isinstance(device, int):
            idx = device
        else:
            s = str(device)
            if s.startswith("cuda") and ":" in s:
                idx = int(s.split(":")[1])
            elif s.startswith("cuda"):
                idx = 0
                try:
                    import torch
                    if torch.cuda.is_available():
                        idx = torch.cuda.current_device()
                except Exception:
                    pass
            elif s != "cpu":
                raise ValueError(f"unsupported device {device!r}")
        self._io_device = device
        if idx is not None and idx != self._gpu:
            self._gpu = idx
            self._release()
        return self

    def parameters(self):
        return iter(())

    # ------------------------------------------------------------------ handle management
    def _release(self):
        if self._handle is not None:
            _lib.load().pesto_destroy(self._handle)
            self._handle = None
        self._last_device_call = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self):
        if self._handle is None:
            if self._blob is None:
                raise RuntimeError("load_state_dict() must be called before forward()")
            lib = _lib.load()
            h = ctypes.c_void_p()
            _lib.check(lib.pesto_create(ctypes.byref(self._cc), self._blob.ctypes.data_as(ctypes.c_void_p), self._blob.size,
                                        self._gpu, ctypes.byref(h)))
            self._handle = h
            if self._debug != (0, 0):
                _lib.check(lib.pesto_debug_select(h, *self._debug))
            if self._edge_mode:
                _lib.check(lib.pesto_debug_edge_mode(h, self._edge_mode))
            if self.async_auto:
                _lib.check(lib.pesto_set_async_auto(h, 1))
            if getattr(self, "_pad_trigger", None) is not None:
                _lib.check(lib.pesto_set_auto_pad_trigger(h, 1 if self._pad_trigger else 0))
            if getattr(self, "_state_limit", None) is not None:
                _lib.check(lib.pesto_set_auto_state_limit(h, self._state_limit))
        return self._handle

    @property
    def handle(self):
        return self._ensure()

    # ------------------------------------------------------------------ forward
    def _segments(self, M):
        """(res_of_atom, R) of the dense residue mask M [N,R]. A mask on the GPU is reduced there by k_mask_to_segments
        (pesto_mask_to_segments: one pass over M on torch's current stream, no host round trip, no ATen kernel); rows with != 1 member
        or an empty residue column poison the array, so the forward that consumes it reports PESTO_ERR_INVALID / NaN logits."""
        if _is_torch(M) and M.is_cuda:
            import torch
            if M.device.index != self._gpu:
                raise RuntimeError(f"inputs are on cuda:{M.device.index} but the model is on cuda:{self._gpu} (use .to())")
            Mc = M.detach()
            if Mc.dtype != torch.float32 or not Mc.is_contiguous():
                Mc = Mc.to(torch.float32).contiguous()
            N, R = int(Mc.shape[0]), int(Mc.shape[1])
            roa = torch.empty((N,), dtype=torch.int32, device=M.device)
            stream = torch.cuda.current_stream(M.device).cuda_stream
            _lib.check(_lib.load().pesto_mask_to_segments(self._ensure(), N, R, Mc.data_ptr(), roa.data_ptr(), _lib.PTR_DEVICE, stream))
            return roa, R
        if self.validate:
            return mask_to_segments(M)
        Mn = M.detach().numpy() if _is_torch(M) else np.asarray(M)
        return Mn.argmax(1).astype(np.int32), int(Mn.shape[1])

    def forward(self, X, ids_topk, q0, M):
        """z = forward(X [N,3] f32, ids_topk [N,k] int (1-based, 0 = sink), q0 [N,N0] f32, M [N,R] 0/1) -> [R,N2] f32"""
        roa, R = self._segments(M)
        return self.forward_segments(X, ids_topk, q0, roa, R)

    __call__ = forward

    def forward_segments(self, X, ids_topk, q0, res_of_atom, R, sizes=None):
        """Same as forward() with the residue mask given as res_of_atom [N] int32 (column index per atom).
        ``sizes``: atom counts of the structures the collated batch consists of - every structure then gets the result of its own
        call (per-structure wrap-around target and max(D); pesto_forward_structures), as in the reference's bulk loops."""
        h = self._ensure()
        lib = _lib.load()
        n0 = self.config["em"]["N0"]
        n_out = self.config["dm"]["N2"]
        offs = None
        if sizes is not None:
            offs = np.zeros(len(sizes) + 1, dtype=np.int32)
            offs[1:] = np.cumsum([int(v) for v in sizes])
            if int(offs[-1]) != int(X.shape[0]):
                raise ValueError(f"sizes sum to {int(offs[-1])}, X has {int(X.shape[0])} atoms")

        def call(N, k, xp, ip, kind, qp, rp, zp, ptr_kind, stream):
            if offs is None:
                return lib.pesto_forward(h, N, R, k, xp, ip, kind, qp, rp, zp, ptr_kind, stream)
            return lib.pesto_forward_structures(h, N, R, k, len(offs) - 1, offs.ctypes.data, xp, ip, kind, qp, rp, zp, ptr_kind, stream)

        if _is_torch(X) and X.is_cuda:
            import torch
            if X.device.index != self._gpu:
                raise RuntimeError(f"inputs are on cuda:{X.device.index} but the model is on cuda:{self._gpu} (use .to())")
            Xc = X.detach().to(torch.float32).contiguous()
            ids = ids_topk.detach()
            if ids.dtype not in (torch.int32, torch.int64):
                ids = ids.to(torch.int64)
            ids = ids.contiguous()
            qc = q0.detach().to(torch.float32).contiguous()
            roa = res_of_atom if _is_torch(res_of_atom) else torch.as_tensor(np.asarray(res_of_atom), device=X.device)
            roa = roa.to(device=X.device, dtype=torch.int32).contiguous()
            N, k = ids.shape
            self._check_shapes(N, Xc.shape, qc.shape, roa.shape, n0)
            z = torch.empty((R, n_out), dtype=torch.float32, device=X.device)
            stream = torch.cuda.current_stream(X.device).cuda_stream
            _lib.check(call(N, k, Xc.data_ptr(), ids.data_ptr(), _lib.IDS_INT64 if ids.dtype == torch.int64 else _lib.IDS_INT32,
                            qc.data_ptr(), roa.data_ptr(), z.data_ptr(), _lib.PTR_DEVICE, stream))
            # precision "auto": checked before the call returned (z is final). async_auto: the check is made by the NEXT call on the handle
            # (or synchronize()), which may repeat flagged structures on the fp32 kernels into the same z - the launch's buffers stay
            # referenced until then.
            self._last_device_call = (Xc, ids, qc, roa, z) if self.async_auto else None
            self._tick_auto_bill()
            return z
        # host path: CPU torch tensors or numpy arrays
        as_torch = _is_torch(X)
        Xn = np.ascontiguousarray(X.detach().numpy() if as_torch else X, dtype=np.float32)
        idn = ids_topk.detach().numpy() if _is_torch(ids_topk) else np.asarray(ids_topk)
        if idn.dtype not in (np.int32, np.int64):
            idn = idn.astype(np.int64)
        idn = np.ascontiguousarray(idn)
        qn = np.ascontiguousarray(q0.detach().numpy() if _is_torch(q0) else q0, dtype=np.float32)
        roa = np.ascontiguousarray(res_of_atom.detach().cpu().numpy() if _is_torch(res_of_atom) else res_of_atom, dtype=np.int32)
        N, k = idn.shape
        self._check_shapes(N, Xn.shape, qn.shape, roa.shape, n0)
        z = np.empty((R, n_out), dtype=np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(call(N, k, p(Xn), p(idn), _lib.IDS_INT64 if idn.dtype == np.int64 else _lib.IDS_INT32,
                        p(qn), p(roa), p(z), _lib.PTR_HOST, None))
        self._tick_auto_bill()
        if as_torch:
            import torch
            return torch.from_numpy(z)
        return z

    @staticmethod
    def _check_shapes(N, xs, qs, rs, n0):
        if tuple(xs) != (N, 3):
            raise ValueError(f"X must be [N,3] with N={N}, got {tuple(xs)}")
        if tuple(qs) != (N, n0):
            raise ValueError(f"q0 must be [N,{n0}], got {tuple(qs)}")
        if tuple(rs) != (N,):
            raise ValueError(f"residue mask must cover N={N} atoms, got {tuple(rs)}")

    # ------------------------------------------------------------------ list of structures -> one launch
    def forward_batch(self, structures, independent=False):
        """[z_b] for structures = [(X, ids_topk0, q0, M), ...] exactly as the reference hands them to collate_batch_features
        (src/dataset.py:91-112; ids_topk0 0-based [N_b, min(64, N_b)] from extract_topology, M the structure's own [N_b, R_b]
        mask): collated on the device and run as ONE batch. Host arrays (numpy / CPU tensors) in, numpy out.
        independent=False: the reference's forward on the collated batch (one max(D), padding wraps to the batch's last atom);
        independent=True: every structure as in its own call, the semantics of the reference's bulk inference loops
        (apply_model.ipynb:139-167) - results do not depend on the grouping (PESTO_BATCH_INDEPENDENT)."""
        h = self._ensure()
        lib = _lib.load()
        n0, n_out = self.config["em"]["N0"], self.config["dm"]["N2"]
        nb = len(structures)
        if nb < 1:
            return []
        keep, zs = [], []
        arr = lambda ct: (ct * nb)()
        Np, Rp, kp = arr(ctypes.c_int64), arr(ctypes.c_int64), arr(ctypes.c_int32)
        Xp, Ip, Qp, Ap, Zp = (arr(ctypes.c_void_p) for _ in range(5))
        kind = None
        for b, (X, ids, q0, M) in enumerate(structures):
            Mh = M.detach().cpu().numpy() if _is_torch(M) else np.asarray(M)
            if Mh.ndim == 1:                      # the compact forms forward_batch_submit takes: res_of_atom [N] and uint8 feature indices
                roa, R = Mh.astype(np.int32), (int(Mh.max()) + 1 if Mh.size else 0)
            else:
                roa, R = mask_to_segments(Mh) if self.validate else (Mh.argmax(1).astype(np.int32), int(Mh.shape[1]))
            Xn = np.ascontiguousarray(X.detach().cpu().numpy() if _is_torch(X) else X, dtype=np.float32)
            idn = ids.detach().cpu().numpy() if _is_torch(ids) else np.asarray(ids)
            idn = np.ascontiguousarray(idn.astype(np.int32) if idn.dtype != np.int64 else idn)
            if kind is None:
                kind = idn.dtype
            elif idn.dtype != kind:
                idn = np.ascontiguousarray(idn.astype(kind))
            qh = q0.detach().cpu().numpy() if _is_torch(q0) else np.asarray(q0)
            if qh.dtype == np.uint8:              # indices -> the dense one-hot rows this (synchronous) entry point uploads
                offs = self.ONEHOT_OFFSETS[n0]
                dense = np.zeros((qh.shape[0], n0), np.float32)
                for c, o in enumerate(offs):
                    dense[np.arange(qh.shape[0]), o + qh[:, c].astype(np.int64)] = 1.0
                qh = dense
            qn = np.ascontiguousarray(qh, dtype=np.float32)
            roa = np.ascontiguousarray(roa.detach().cpu().numpy() if _is_torch(roa) else roa, dtype=np.int32)
            N = Xn.shape[0]
            if idn.ndim != 2 or idn.shape[0] != N:
                raise ValueError(f"structure {b}: ids_topk must be [N, k] with N={N}")
            self._check_shapes(N, Xn.shape, qn.shape, roa.shape, n0)
            z = np.empty((R, n_out), dtype=np.float32)
            keep.append((Xn, idn, qn, roa))
            zs.append(z)
            Np[b], Rp[b], kp[b] = N, R, idn.shape[1]
            Xp[b], Ip[b], Qp[b], Ap[b], Zp[b] = Xn.ctypes.data, idn.ctypes.data, qn.ctypes.data, roa.ctypes.data, z.ctypes.data
        _lib.check(lib.pesto_forward_batch(h, nb, Np, Rp, kp, Xp, Ip, _lib.IDS_INT64 if kind == np.int64 else _lib.IDS_INT32, Qp, Ap, Zp,
                                           _lib.BATCH_INDEPENDENT if independent else _lib.BATCH_COLLATED, None))
        self._tick_auto_bill(every=1)
        return zs

    # ------------------------------------------------------------------ pipelined launches (SURVEY 8b threading row)
    ONEHOT_OFFSETS = {30: (0,), 123: (0, 30, 59)}       # feature blocks of encode_features (src/data_encoding.py:78-84): element | resname | atom name

    def forward_batch_submit(self, structures, independent=True, compact=True):
        """First half of forward_batch (pesto_forward_batch_submit): packs the structures into a pinned staging slot and queues the
        H2D copy, the forward and the D2H copy; returns a ticket at once. Up to two tickets may be in flight, so the packing and the copy
        of launch t + 1 overlap the kernels of launch t (the reference feeds its loop from DataLoader workers the same way,
        interfaceome/apply_model.py:50-82). ``compact``: neighbour ids travel as uint16 and one-hot features as byte indices (145
        instead of 392 bytes per atom over PCIe; expanded on the GPU, same bits) - the library finds the indices itself while it packs
        the rows (a launch with a row that is not one-hot in the reference's feature blocks travels dense).
        Per structure (X, ids_topk0, q, M) as forward_batch takes them; to skip the host work a caller that already has them may
        pass q as the uint8 index array [N, n_index] (encode_features' argmax per block) and M as res_of_atom [N] (int32, the
        residue column of every atom, 0 .. R-1) - the Python layer then touches no array element."""
        h = self._ensure()
        lib = _lib.load()
        n0, n_out = self.config["em"]["N0"], self.config["dm"]["N2"]
        nb = len(structures)
        if nb < 1:
            raise ValueError("no structures")
        offs = self.ONEHOT_OFFSETS.get(n0)
        arr = lambda ct: (ct * nb)()
        Np, Rp, kp = arr(ctypes.c_int64), arr(ctypes.c_int64), arr(ctypes.c_int32)
        Xp, Ip, Qp, Jp, Ap, Zp = (arr(ctypes.c_void_p) for _ in range(6))
        zs, keep = [], []
        host = lambda a: a.detach().cpu().numpy() if _is_torch(a) else np.asarray(a)
        qhs = [host(s[2]) for s in structures]            # (one device-to-host copy per structure when q lives on the GPU - ADVICE r4)
        n_idx_given = sum(1 for qh in qhs if qh.dtype == np.uint8)
        if n_idx_given not in (0, nb):
            raise ValueError("either every structure passes q as uint8 indices or none does")
        if n_idx_given and offs is None:
            raise ValueError(f"no one-hot feature blocks known for N0 = {n0}")
        # the reference's own forms - bool / float mask M (encode_structure, src/data_encoding.py:61-75), int64 / int32 ids (extract_topology,
        # :87-102), float one-hot q (encode_features, :78-84) - go to the native layers as they are: the dense mask is reduced by ONE native
        # pass (libpesto_io, a quarter of the bytes for the bool mask), the ids are narrowed to uint16 and range-checked by the packer's own
        # copy (PESTO_IDS_NARROW), the one-hot rows are found while they are packed. No numpy pass touches an element here.
        ids_kinds = set()
        for b, (X, ids, q0, M) in enumerate(structures):
            Mh = host(M)
            if Mh.ndim == 1:                        # res_of_atom given: the library checks its range, the forward that every residue has an atom
                roa = np.ascontiguousarray(Mh, dtype=np.int32)
                R = int(roa.max()) + 1 if roa.size else 0
            else:
                # dense mask: always through the checked native reduction (it costs what an unchecked argmax costs; validate=False used to
                # skip the check to save four numpy passes)
                roa, R = mask_to_segments(Mh)
                roa = np.ascontiguousarray(roa, dtype=np.int32)
            Xn = np.ascontiguousarray(host(X), dtype=np.float32)
            idr = host(ids)
            if idr.dtype not in (np.int32, np.int64, np.uint16):
                idr = idr.astype(np.int32)
            if idr.dtype == np.uint16 and not compact:
                idr = idr.astype(np.int32)
            idn = np.ascontiguousarray(idr)
            ids_kinds.add(idn.dtype)
            N = Xn.shape[0]
            if idn.ndim != 2 or idn.shape[0] != N:
                raise ValueError(f"structure {b}: ids_topk must be [N, k] with N={N}")
            qh = qhs[b]
            if n_idx_given:
                qn = np.ascontiguousarray(qh)
                if qn.shape != (N, len(offs)):
                    raise ValueError(f"structure {b}: q as indices must be uint8 [N, {len(offs)}], got {qn.shape}")
                self._check_shapes(N, Xn.shape, (N, n0), roa.shape, n0)
            else:
                qn = np.ascontiguousarray(qh, dtype=np.float32)
                self._check_shapes(N, Xn.shape, qn.shape, roa.shape, n0)
            z = np.empty((R, n_out), dtype=np.float32)
            keep.append((Xn, idn, qn, roa))
            zs.append(z)
            Np[b], Rp[b], kp[b] = N, R, idn.shape[1]
            Xp[b], Ip[b], Ap[b], Zp[b] = Xn.ctypes.data, idn.ctypes.data, roa.ctypes.data, z.ctypes.data
            (Jp if n_idx_given else Qp)[b] = qn.ctypes.data
        # dense q + block offsets: the library detects one-hot rows while packing (compact only)
        use_offs = offs is not None and (n_idx_given or compact)
        io = (ctypes.c_int32 * 3)(*(list(offs) + [0] * (3 - len(offs)))) if use_offs else None
        if len(ids_kinds) != 1:      # (mixed tables: one common kind)
            common = np.dtype(np.int64) if np.dtype(np.int64) in ids_kinds else np.dtype(np.int32)
            for b in range(nb):
                Xn, idn, qn, roa = keep[b]
                if idn.dtype != common:
                    idn = np.ascontiguousarray(idn.astype(common))
                    keep[b] = (Xn, idn, qn, roa)
                    Ip[b] = idn.ctypes.data
            ids_kinds = {common}
        kind = {np.dtype(np.uint16): _lib.IDS_UINT16, np.dtype(np.int32): _lib.IDS_INT32, np.dtype(np.int64): _lib.IDS_INT64}[np.dtype(next(iter(ids_kinds)))]
        if compact and kind != _lib.IDS_UINT16:
            kind |= _lib.IDS_NARROW
        t = ctypes.c_int32(-1)
        _lib.check(lib.pesto_forward_batch_submit(h, nb, Np, Rp, kp, Xp, Ip, kind, None if n_idx_given else Qp, Jp if n_idx_given else None,
                                                  len(offs) if use_offs else 0, io, Ap, Zp,
                                                  _lib.BATCH_INDEPENDENT if independent else _lib.BATCH_COLLATED, ctypes.byref(t)))
        self._tickets = getattr(self, "_tickets", {})
        self._tickets[t.value] = zs          # the logits arrays must stay alive until the wait (the packed inputs need not)
        return t.value

    def forward_batch_wait(self, ticket):
        """Second half: blocks until the launch behind ``ticket`` has finished and returns its [z_b] (numpy)."""
        zs = getattr(self, "_tickets", {}).pop(ticket, None)
        if zs is None:
            raise ValueError(f"no launch in flight under ticket {ticket}")
        _lib.check(_lib.load().pesto_forward_batch_wait(self._ensure(), ticket))
        self._tick_auto_bill(every=1)
        return zs

    # ------------------------------------------------------------------ trajectory frames
    def forward_frames(self, X_frames, ids_topk, q0, M, frame_axis=0, frames_per_launch=0):
        """z [F, R, N2] for F coordinate frames of the same atoms with ONE topology: what the reference's MD loop
        ``for i in frames: z_i = model(X_traj[:, i], ids_topk, q, M)`` (md_analysis/apply_model_md.ipynb cell 6) computes, with
        several frames per kernel launch. ``X_frames``: [F, N, 3] (frame_axis=0) or the reference's [N, F, 3] trajectory
        tensor (frame_axis=1); strided views are read in place (no copy). Other arguments as forward()."""
        roa, R = self._segments(M)
        return self.forward_frames_segments(X_frames, ids_topk, q0, roa, R, frame_axis, frames_per_launch)

    def forward_frames_segments(self, X_frames, ids_topk, q0, res_of_atom, R, frame_axis=0, frames_per_launch=0):
        h = self._ensure()
        lib = _lib.load()
        n0 = self.config["em"]["N0"]
        n_out = self.config["dm"]["N2"]
        if frame_axis not in (0, 1) or len(X_frames.shape) != 3 or X_frames.shape[2] != 3:
            raise ValueError("X_frames must be [F,N,3] (frame_axis=0) or [N,F,3] (frame_axis=1)")
        F, N = (X_frames.shape[0], X_frames.shape[1]) if frame_axis == 0 else (X_frames.shape[1], X_frames.shape[0])
        if _is_torch(X_frames) and X_frames.is_cuda:
            import torch
            dev = X_frames.device
            if dev.index != self._gpu:
                raise RuntimeError(f"inputs are on cuda:{dev.index} but the model is on cuda:{self._gpu} (use .to())")
            Xc = X_frames.detach()
            if Xc.dtype != torch.float32 or Xc.stride(2) != 1:
                Xc = Xc.to(torch.float32).contiguous()
            fs, as_ = (Xc.stride(0), Xc.stride(1)) if frame_axis == 0 else (Xc.stride(1), Xc.stride(0))
            ids = ids_topk.detach()
            if ids.dtype not in (torch.int32, torch.int64):
                ids = ids.to(torch.int64)
            ids = ids.to(dev).contiguous()
            qc = q0.detach().to(device=dev, dtype=torch.float32).contiguous()
            roa = res_of_atom if _is_torch(res_of_atom) else torch.as_tensor(np.asarray(res_of_atom))
            roa = roa.to(device=dev, dtype=torch.int32).contiguous()
            if ids.shape[0] != N:
                raise ValueError(f"ids_topk has {ids.shape[0]} rows, frames have {N} atoms")
            self._check_shapes(N, (N, 3), qc.shape, roa.shape, n0)
            z = torch.empty((F, R, n_out), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.pesto_forward_frames(h, N, R, ids.shape[1], F, Xc.data_ptr(), fs, as_, ids.data_ptr(),
                                                _lib.IDS_INT64 if ids.dtype == torch.int64 else _lib.IDS_INT32, qc.data_ptr(),
                                                roa.data_ptr(), z.data_ptr(), frames_per_launch, _lib.PTR_DEVICE, stream))
            self._last_device_call = (Xc, ids, qc, roa, z) if self.async_auto else None      # (as in forward_segments)
            return z
        as_torch = _is_torch(X_frames)
        Xn = np.asarray(X_frames.detach().numpy() if as_torch else X_frames)
        if Xn.dtype != np.float32 or Xn.strides[2] != 4 or Xn.strides[0] % 4 or Xn.strides[1] % 4:
            Xn = np.ascontiguousarray(Xn, dtype=np.float32)
        fs, as_ = (Xn.strides[0] // 4, Xn.strides[1] // 4) if frame_axis == 0 else (Xn.strides[1] // 4, Xn.strides[0] // 4)
        idn = ids_topk.detach().numpy() if _is_torch(ids_topk) else np.asarray(ids_topk)
        if idn.dtype not in (np.int32, np.int64):
            idn = idn.astype(np.int64)
        idn = np.ascontiguousarray(idn)
        qn = np.ascontiguousarray(q0.detach().numpy() if _is_torch(q0) else q0, dtype=np.float32)
        roa = np.ascontiguousarray(res_of_atom.detach().cpu().numpy() if _is_torch(res_of_atom) else res_of_atom, dtype=np.int32)
        if idn.shape[0] != N:
            raise ValueError(f"ids_topk has {idn.shape[0]} rows, frames have {N} atoms")
        self._check_shapes(N, (N, 3), qn.shape, roa.shape, n0)
        z = np.empty((F, R, n_out), dtype=np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(lib.pesto_forward_frames(h, N, R, idn.shape[1], F, p(Xn), fs, as_, p(idn),
                                            _lib.IDS_INT64 if idn.dtype == np.int64 else _lib.IDS_INT32, p(qn), p(roa), p(z),
                                            frames_per_launch, _lib.PTR_HOST, None))
        if as_torch:
            import torch
            return torch.from_numpy(z)
        return z

    # ------------------------------------------------------------------ post-processing on the GPU
    def postprocess(self, z, M_or_res_of_atom=None):
        """(p, bfactor): p = sigmoid(z) [R, N2] (apply_model.ipynb:160) and, when the residue mask M [N,R] (or res_of_atom [N]) is
        given, bfactor [N2, N] with bfactor[c][i] = p[residue of atom i][c] - encode_bfactor's per-residue branch
        (src/structure.py:208-218) for every output channel at once. ROCm tensors stay on the GPU (asynchronous)."""
        h = self._ensure()
        lib = _lib.load()
        n_out = self.config["dm"]["N2"]
        R = int(z.shape[0])
        roa = None
        if M_or_res_of_atom is not None:
            m = M_or_res_of_atom
            if len(m.shape) == 2:
                if int(m.shape[1]) != R:
                    raise ValueError(f"M has {m.shape[1]} residue columns, z has {R} rows")
                roa = m.argmax(1)
            else:
                roa = m
        if _is_torch(z) and z.is_cuda:
            import torch
            zc = z.detach().to(torch.float32).contiguous()
            p = torch.empty_like(zc)
            bf, rp, N = None, 0, 1
            if roa is not None:
                roa = (roa if _is_torch(roa) else torch.as_tensor(np.asarray(roa))).to(device=z.device, dtype=torch.int32).contiguous()
                N = int(roa.shape[0])
                bf = torch.empty((n_out, N), dtype=torch.float32, device=z.device)
                rp = roa.data_ptr()
            stream = torch.cuda.current_stream(z.device).cuda_stream
            _lib.check(lib.pesto_postprocess(h, N, R, zc.data_ptr(), rp, p.data_ptr(), bf.data_ptr() if bf is not None else 0,
                                             _lib.PTR_DEVICE, stream))
            return p, bf
        as_torch = _is_torch(z)
        zn = np.ascontiguousarray(z.detach().numpy() if as_torch else z, dtype=np.float32)
        p = np.empty_like(zn)
        bf, N, rp = None, 1, None
        if roa is not None:
            roa = np.ascontiguousarray(roa.detach().cpu().numpy() if _is_torch(roa) else roa, dtype=np.int32)
            N = int(roa.shape[0])
            bf = np.empty((n_out, N), dtype=np.float32)
            rp = roa.ctypes.data_as(ctypes.c_void_p)
        _lib.check(lib.pesto_postprocess(h, N, R, zn.ctypes.data_as(ctypes.c_void_p), rp, p.ctypes.data_as(ctypes.c_void_p),
                                         bf.ctypes.data_as(ctypes.c_void_p) if bf is not None else None, _lib.PTR_HOST, None))
        if as_torch:
            import torch
            return torch.from_numpy(p), (torch.from_numpy(bf) if bf is not None else None)
        return p, bf

    # ------------------------------------------------------------------ k-NN topology + collate on the GPU
    def knn_collate(self, X, sizes, k=64):
        """ids_topk [sum(sizes), 64] for a concatenated batch: what ``extract_topology(X_s, 64)[0]`` per structure followed by
        ``collate_batch_features`` produce on the host in the reference (src/data_encoding.py:87-102, src/dataset.py:100-109):
        exact k nearest neighbours within each structure, ascending distance, D < 1e-2 entries last, 1-based batch-global ids,
        0 = padding. X: [sum(sizes), 3] float32 (ROCm tensor -> int64 ROCm tensor on the current stream; CPU tensor / numpy -> same kind)."""
        h = self._ensure()
        lib = _lib.load()
        sizes = [int(v) for v in sizes]
        offs = np.zeros(len(sizes) + 1, dtype=np.int32)
        offs[1:] = np.cumsum(sizes)
        n = int(offs[-1])
        if _is_torch(X) and X.is_cuda:
            import torch
            Xc = X.detach().to(torch.float32).contiguous()
            if tuple(Xc.shape) != (n, 3):
                raise ValueError(f"X must be [{n},3], got {tuple(Xc.shape)}")
            ids = torch.empty((n, 64), dtype=torch.int64, device=X.device)
            stream = torch.cuda.current_stream(X.device).cuda_stream
            _lib.check(lib.pesto_knn_collate(h, n, len(sizes), offs.ctypes.data, Xc.data_ptr(), k, ids.data_ptr(), _lib.IDS_INT64,
                                             _lib.PTR_DEVICE, stream))
            return ids
        as_torch = _is_torch(X)
        Xn = np.ascontiguousarray(X.detach().numpy() if as_torch else X, dtype=np.float32)
        if Xn.shape != (n, 3):
            raise ValueError(f"X must be [{n},3], got {Xn.shape}")
        ids = np.empty((n, 64), dtype=np.int64)
        _lib.check(lib.pesto_knn_collate(h, n, len(sizes), offs.ctypes.data, Xn.ctypes.data, k, ids.ctypes.data, _lib.IDS_INT64,
                                         _lib.PTR_HOST, None))
        if as_torch:
            import torch
            return torch.from_numpy(ids)
        return ids

    def knn_tie_rows(self, X, sizes, ids_topk, k=64):
        """uint8 [sum(sizes)]: bit 0 / 1 / 2 / 3 set where an exact float32 distance tie of that row straddles the cut after column
        8 / 16 / 32 / 64 of ``ids_topk`` (the table of knn_collate for the same X / sizes) - the rows on which the reference's torch.topk
        may have picked the other of two equally distant atoms for a layer's neighbourhood (pesto_knn_tie_rows). numpy in, numpy out."""
        h = self._ensure()
        lib = _lib.load()
        sizes = [int(v) for v in sizes]
        offs = np.zeros(len(sizes) + 1, dtype=np.int32)
        offs[1:] = np.cumsum(sizes)
        n = int(offs[-1])
        if _is_torch(X) and X.is_cuda:                # ROCm tensors: uint8 ROCm tensor on the current stream
            import torch
            Xc = X.detach().to(torch.float32).contiguous()
            idc = ids_topk.detach()
            idc = (idc if idc.dtype in (torch.int32, torch.int64) else idc.to(torch.int64)).contiguous()
            if tuple(Xc.shape) != (n, 3) or tuple(idc.shape) != (n, 64):
                raise ValueError(f"X must be [{n},3] and ids_topk [{n},64]")
            fl = torch.empty((n,), dtype=torch.uint8, device=X.device)
            _lib.check(lib.pesto_knn_tie_rows(h, n, len(sizes), offs.ctypes.data, Xc.data_ptr(), k, idc.data_ptr(),
                                              _lib.IDS_INT64 if idc.dtype == torch.int64 else _lib.IDS_INT32, fl.data_ptr(), _lib.PTR_DEVICE,
                                              torch.cuda.current_stream(X.device).cuda_stream))
            return fl
        Xn = np.ascontiguousarray(X.detach().cpu().numpy() if _is_torch(X) else X, dtype=np.float32)
        idn = ids_topk.detach().cpu().numpy() if _is_torch(ids_topk) else np.asarray(ids_topk)
        idn = np.ascontiguousarray(idn if idn.dtype in (np.int32, np.int64) else idn.astype(np.int64))
        if Xn.shape != (n, 3) or idn.shape != (n, 64):
            raise ValueError(f"X must be [{n},3] and ids_topk [{n},64]")
        flags = np.zeros(n, np.uint8)
        _lib.check(lib.pesto_knn_tie_rows(h, n, len(sizes), offs.ctypes.data, Xn.ctypes.data, k, idn.ctypes.data,
                                          _lib.IDS_INT64 if idn.dtype == np.int64 else _lib.IDS_INT32, flags.ctypes.data, _lib.PTR_HOST, None))
        return flags

    # ------------------------------------------------------------------ per-stage access (tests)
    def stage_embed(self, q0):
        q0 = np.ascontiguousarray(q0, np.float32)
        out = np.empty((q0.shape[0], 32), np.float32)
        _lib.check(_lib.load().pesto_stage_embed(self._ensure(), q0.shape[0], q0.ctypes.data, out.ctypes.data))
        return out

    def stage_unpack(self, X, ids_topk):
        X = np.ascontiguousarray(X, np.float32)
        ids = np.ascontiguousarray(ids_topk)
        kind = _lib.IDS_INT64 if ids.dtype == np.int64 else _lib.IDS_INT32
        if kind == _lib.IDS_INT32:
            ids = np.ascontiguousarray(ids, np.int32)
        n, k = ids.shape
        D = np.empty((n + 1, k), np.float32)
        R = np.empty((n + 1, k, 3), np.float32)
        _lib.check(_lib.load().pesto_stage_unpack(self._ensure(), n, k, X.ctypes.data, ids.ctypes.data, kind, D.ctypes.data, R.ctypes.data))
        return D, R

    def stage_layer(self, layer, q, p):
        q = np.ascontiguousarray(q, np.float32).copy()
        p = np.ascontiguousarray(p, np.float32).copy()
        _lib.check(_lib.load().pesto_stage_layer(self._ensure(), layer, q.ctypes.data, p.ctypes.data))
        return q, p

    def stage_pool(self, q, p, res_of_atom, R):
        q = np.ascontiguousarray(q, np.float32)
        p = np.ascontiguousarray(p, np.float32)
        roa = np.ascontiguousarray(res_of_atom, np.int32)
        qr = np.empty((R, 32), np.float32)
        pr = np.empty((R, 3, 32), np.float32)
        z = np.empty((R, self.config["dm"]["N2"]), np.float32)
        _lib.check(_lib.load().pesto_stage_pool(self._ensure(), q.shape[0], R, q.ctypes.data, p.ctypes.data, roa.ctypes.data,
                                               qr.ctypes.data, pr.ctypes.data, z.ctypes.data))
        return qr, pr, z

    # ------------------------------------------------------------------ timing hooks (bench.py)
    def set_timing(self, enabled=True, per_kernel=False):
        _lib.check(_lib.load().pesto_set_timing(self._ensure(), (2 if per_kernel else 1) if enabled else 0))

    def get_kernel_timing(self):
        """{class: (summed ms, launches)} of the most recent forward timed with set_timing(True, per_kernel=True)."""
        ms, n = (ctypes.c_double * 5)(), (ctypes.c_int32 * 5)()
        _lib.check(_lib.load().pesto_get_kernel_timing(self._ensure(), ms, n))
        names = ("node", "edge_nn8", "edge_nn16", "edge_nn32", "edge_nn64")
        return {k: (ms[i], n[i]) for i, k in enumerate(names)}

    def get_timing(self):
        a, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
        _lib.check(_lib.load().pesto_get_timing(self._ensure(), ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return {"layers_ms": a.value, "total_ms": b.value, "n_layer_launches": n.value}

    def synchronize(self):
        _lib.check(_lib.load().pesto_synchronize(self._ensure()))


def blob_keys(config):
    return [k for k, _ in blob_schema(config)]

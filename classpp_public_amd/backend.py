"""Thin Python host layer over the C ABI (include/cpt.h): owns a cpt_handle, keeps bulk arrays in HBM as torch
tensors (PyTorch is only the device-memory / stream / torch.distributed plumbing here, not the compute path).
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import CptStepstat


class CptError(RuntimeError):
    """Computation failure inside the backend (the reference raises std::runtime_error -> CosmoComputationError)."""


class CptInputError(ValueError):
    """Invalid / unsupported input (the reference raises std::invalid_argument -> CosmoSevereError)."""


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Backend:
    def __init__(self, inputs, device="cuda:0"):
        if not torch.cuda.is_available():
            raise CptError("no HIP device visible to torch: the cpt backend has no CPU fallback")
        self.lib = capi.lib()
        self.inp = inputs
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.lib.cpt_create(C.byref(inputs.config), C.byref(inputs.tables), C.byref(h))
        if rc != capi.CPT_OK:
            msg = self.lib.cpt_create_error().decode()
            raise (CptInputError if rc in (capi.CPT_ERR_INVALID, capi.CPT_ERR_UNSUPPORTED) else CptError)(msg)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.cpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fence(self):
        """The library runs on its own non-blocking HIP stream (cpt_api.hip) and returns with that stream drained.  Device buffers
        handed to it may still be in flight on torch's side (an RCCL all_gather and the index_copy_ that assembles the full sources,
        a freshly recycled block of the caching allocator): drain torch's current stream before every entry point."""
        torch.cuda.current_stream(self.device).synchronize()

    def _check(self, rc):
        if rc != capi.CPT_OK:
            msg = self.lib.cpt_last_error(self.h).decode()
            raise (CptInputError if rc in (capi.CPT_ERR_INVALID, capi.CPT_ERR_UNSUPPORTED) else CptError)(msg)

    # ---- hot path A ----
    def perturb_solve(self, k=None, tau=None, want_sources=True):
        """-> (sources [tp][ntau][nk] torch f64 on device or None, stats ndarray of CptStepstat, status int32[nk])"""
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        tau = np.ascontiguousarray(self.inp.tau if tau is None else tau, dtype=np.float64)
        nk, ntau = k.size, tau.size
        out = None
        ptr = None
        if want_sources:
            out = torch.empty((self.inp.config.tp_size, ntau, nk), dtype=torch.float64, device=self.device)
            ptr = C.c_void_p(out.data_ptr())
        stats = (CptStepstat * nk)()
        status = np.zeros(nk, dtype=np.int32)
        self._fence()
        rc = self.lib.cpt_perturb_solve_batch(self.h, _dptr(k), nk, _dptr(tau), ntau, ptr, stats, _iptr(status))
        self._check(rc)
        return out, stats, status

    # ---- hot path B ----
    def transfer(self, sources=None, k=None, tau=None, q=None, l=None, k_size_cl=None):
        """sources: torch f64 device tensor [tp][ntau][nk] (reference layout) or None (use resident sources).
        -> transfer [tt][nl][nq] torch f64 on device"""
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        tau = np.ascontiguousarray(self.inp.tau if tau is None else tau, dtype=np.float64)
        q = np.ascontiguousarray(self.inp.q if q is None else q, dtype=np.float64)
        l = np.ascontiguousarray(self.inp.l if l is None else l, dtype=np.int32)
        k_size_cl = self.inp.k_size_cl if k_size_cl is None else k_size_cl
        sp = None
        if sources is not None:
            assert sources.is_cuda and sources.dtype == torch.float64 and sources.is_contiguous()
            assert tuple(sources.shape) == (self.inp.config.tp_size, tau.size, k.size), sources.shape
            sp = C.c_void_p(sources.data_ptr())
        out = torch.empty((self.inp.config.tt_size, l.size, q.size), dtype=torch.float64, device=self.device)
        self._fence()
        rc = self.lib.cpt_transfer_batch(self.h, sp, _dptr(k), k.size, k_size_cl, _dptr(tau), tau.size, _dptr(q), q.size,
                                         _iptr(l), l.size, C.c_void_p(out.data_ptr()))
        self._check(rc)
        return out

    # ---- the whole pass, fused: one library call, one stream synchronisation ----
    def step(self, lensing=None, want_pk=None):
        """k-modes -> sources -> transfer functions -> C_l (-> lensed C_l) (-> P(k)) through cpt_step.
        lensing: None or (l_unlensed_max, delta_l_max[, accurate, num_mu_minus_lmax, tol]); want_pk: default = the configuration has delta_m.
        -> dict(transfer, cl, cl_lensed, pk, stats, status); device tensors are owned by the backend and reused by the next step()"""
        from .capi import CptLensingParams, CptStepIo
        inp = self.inp
        if want_pk is None:
            want_pk = inp.config.index_tp_delta_m >= 0
        key = (tuple(lensing) if lensing else None, bool(want_pk))
        st = getattr(self, "_step_state", None)
        if st is None or st["key"] != key:
            k = np.ascontiguousarray(inp.k, dtype=np.float64); tau = np.ascontiguousarray(inp.tau, dtype=np.float64)
            q = np.ascontiguousarray(inp.q, dtype=np.float64); l = np.ascontiguousarray(inp.l, dtype=np.int32)
            dev, f64 = self.device, torch.float64
            st = {"key": key, "k": k, "tau": tau, "q": q, "l": l,
                  "transfer": torch.empty((inp.config.tt_size, l.size, q.size), dtype=f64, device=dev),
                  "cl": torch.empty((l.size, inp.spectra.ct_size), dtype=f64, device=dev),
                  "cl_lensed": None, "pk": torch.empty(k.size, dtype=f64, device=dev) if want_pk else None,
                  "stats": (CptStepstat * k.size)(), "status": np.zeros(k.size, dtype=np.int32), "lp": None}
            io = CptStepIo()
            io.k = _dptr(k); io.nk = k.size; io.k_size_cl = inp.k_size_cl
            io.tau_sampling = _dptr(tau); io.ntau = tau.size; io.q = _dptr(q); io.nq = q.size; io.l = _iptr(l); io.nl = l.size
            io.sp = C.pointer(inp.spectra)
            if lensing:
                a = list(lensing) + [500, False, 70, 0.0][len(lensing) - 1:]
                lp = CptLensingParams(int(a[0]), int(a[1]), int(bool(a[2])), int(a[3]), float(a[4]))
                n = self.lib.cpt_lensing_l_size(_iptr(l), l.size, C.byref(lp))
                if n < 1:
                    raise CptInputError("cpt_lensing_l_size failed")
                st["lp"] = lp
                st["cl_lensed"] = torch.empty((n, inp.spectra.ct_size), dtype=f64, device=dev)
                io.lp = C.pointer(lp)
                io.cl_lensed_dev = st["cl_lensed"].data_ptr()
            io.transfer_dev = st["transfer"].data_ptr(); io.cl_dev = st["cl"].data_ptr()
            io.pk_dev = st["pk"].data_ptr() if want_pk else None
            io.stats = st["stats"]; io.status = _iptr(st["status"])
            st["io"] = io
            self._step_state = st
            self._fence()   # the fresh output tensors may be recycled blocks still in flight on torch's stream
        self._check(self.lib.cpt_step(self.h, C.byref(st["io"])))
        return st

    # ---- multi-GPU: the two exchanges of the sharded path as RCCL operations behind the C ABI (csrc/cpt_comm.hip) ----
    def comm_unique_id(self):
        """rank 0: the id every rank passes to comm_init (distribute the 128 bytes by any means)"""
        buf = C.create_string_buffer(128)
        rc = self.lib.cpt_comm_get_unique_id(buf)
        if rc != capi.CPT_OK:
            raise CptError(self.lib.cpt_create_error().decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._check(self.lib.cpt_comm_init(self.h, C.c_char_p(unique_id), int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def allgather_sources(self, nk_all, ntau=None):
        """every rank has integrated its k shard (perturb_solve(k=k_all[rank::world], want_sources=False)): afterwards the full sources are resident"""
        self._check(self.lib.cpt_allgather_sources(self.h, int(nk_all), int(self.inp.ntau if ntau is None else ntau)))

    def gather_transfer(self, local, nl_all):
        """local: device [tt][nl_local][nq] of this rank's multipoles -> on rank 0 the full device table [tt][nl_all][nq], None elsewhere"""
        assert local.is_cuda and local.is_contiguous()
        out = torch.empty((local.shape[0], int(nl_all), local.shape[2]), dtype=torch.float64, device=self.device) if self.rank == 0 else None
        self._fence()
        self._check(self.lib.cpt_gather_transfer(self.h, C.c_void_p(local.data_ptr()), int(nl_all), int(local.shape[2]),
                                                 C.c_void_p(out.data_ptr()) if out is not None else None))
        return out

    def gather_cl(self, cl_local, nl_all):
        """cl_local: device [nl_local][ct] of this rank's multipoles -> on rank 0 the full device table [nl_all][ct], None elsewhere"""
        assert cl_local.is_cuda and cl_local.is_contiguous()
        out = torch.empty((int(nl_all), cl_local.shape[1]), dtype=torch.float64, device=self.device) if self.rank == 0 else None
        self._fence()
        self._check(self.lib.cpt_gather_cl(self.h, C.c_void_p(cl_local.data_ptr()), int(nl_all), int(cl_local.shape[1]),
                                           C.c_void_p(out.data_ptr()) if out is not None else None))
        return out

    def dbg_pad_rows(self, x, n_max):
        out = torch.empty((x.shape[0], n_max, x.shape[2]), dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_dbg_pad_rows(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.shape[0], x.shape[1], n_max, x.shape[2]))
        return out

    def dbg_uninterleave(self, blocks, n_all):
        world, nb, n_max, ninner = blocks.shape
        out = torch.empty((nb, n_all, ninner), dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_dbg_uninterleave(self.h, C.c_void_p(blocks.data_ptr()), C.c_void_p(out.data_ptr()), world, nb, n_max, n_all, ninner))
        return out

    def step_gpu_ms(self):
        """milliseconds from the first to the last kernel of the last step() on the library's stream"""
        return self.kernel_ms(3)[0]

    # ---- "next" rows: observables ----
    def cl(self, transfer, q=None):
        """transfer: device [tt][nl][nq] -> C_l table [nl][ct_size] on device (cpt_cl_batch)"""
        q = np.ascontiguousarray(self.inp.q if q is None else q, dtype=np.float64)
        nl = transfer.shape[1]
        out = torch.empty((nl, self.inp.spectra.ct_size), dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_cl_batch(self.h, C.byref(self.inp.spectra), C.c_void_p(transfer.data_ptr()), _dptr(q), q.size, nl,
                                          C.c_void_p(out.data_ptr())))
        return out

    def cl_cross(self, transfer1, transfer2, amplitude, tilt, running, q=None):
        """the cross-correlation spectra of two scalar initial conditions (cpt_cl_cross_batch): transfer tables [tt][nl][nq] of the two
        (device), amplitude / tilt / running of the primordial cross spectrum -> [nl][ct_size] on device"""
        import copy
        q = np.ascontiguousarray(self.inp.q if q is None else q, dtype=np.float64)
        nl = transfer1.shape[1]
        assert transfer1.shape == transfer2.shape and transfer1.is_contiguous() and transfer2.is_contiguous()
        sp = copy.copy(self.inp.spectra)
        sp.A_s, sp.n_s, sp.alpha_s = float(amplitude), float(tilt), float(running)
        out = torch.empty((nl, sp.ct_size), dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_cl_cross_batch(self.h, C.byref(sp), C.c_void_p(transfer1.data_ptr()), C.c_void_p(transfer2.data_ptr()), _dptr(q),
                                                q.size, nl, C.c_void_p(out.data_ptr())))
        return out

    def lensed_cl(self, cl, l_unlensed_max, delta_l_max=500, accurate=False, num_mu_minus_lmax=70, tol_gauss_legendre=0.0, l=None):
        """unlensed C_l table [nl][ct] (device) -> lensed table [l_size][ct] on device (cpt_lensing_batch)"""
        from .capi import CptLensingParams
        l = np.ascontiguousarray(self.inp.l if l is None else l, dtype=np.int32)
        lp = CptLensingParams(int(l_unlensed_max), int(delta_l_max), int(bool(accurate)), int(num_mu_minus_lmax), float(tol_gauss_legendre))
        lptr = l.ctypes.data_as(C.POINTER(C.c_int))
        n = self.lib.cpt_lensing_l_size(lptr, l.size, C.byref(lp))
        if n < 1:
            raise CptInputError("cpt_lensing_l_size failed")
        out = torch.empty((n, self.inp.spectra.ct_size), dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_lensing_batch(self.h, C.byref(self.inp.spectra), C.byref(lp), lptr, l.size, C.c_void_p(cl.data_ptr()),
                                               C.c_void_p(out.data_ptr())))
        return out

    def pk_linear(self, k=None, cb=False):
        """linear P(k) today of total matter, or of baryons + cold dark matter (cb=True, with non-cold species)"""
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        out = torch.empty(k.size, dtype=torch.float64, device=self.device)
        self._fence()
        self._check((self.lib.cpt_pk_cb_linear if cb else self.lib.cpt_pk_linear)(self.h, C.byref(self.inp.spectra), _dptr(k), k.size, C.c_void_p(out.data_ptr())))
        return out

    def sigma(self, R, k=None, k_per_decade=80.0, cb=False):
        """sigma(R [Mpc]) of the linear matter field at z = 0 (cpt_sigma); sigma8 = sigma(8 / h)"""
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        out = C.c_double()
        self._fence()
        self._check((self.lib.cpt_sigma_cb if cb else self.lib.cpt_sigma)(self.h, C.byref(self.inp.spectra), _dptr(k), k.size, float(R), float(k_per_decade), C.byref(out)))
        return out.value

    def pk_at_tau(self, tau_z, ln_tau_size, k=None, cb=False):
        """linear P(k, z) of the redshift whose conformal time is tau_z (0 < z <= z_max_pk): spline in ln tau over the last ln_tau_size sampling
        times, on the device (cpt_pk_at_tau); the sources must be those of this handle's last perturb_solve / step"""
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        out = torch.empty(k.size, dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_pk_at_tau(self.h, C.byref(self.inp.spectra), _dptr(k), k.size, int(ln_tau_size), float(tau_z), int(bool(cb)), C.c_void_p(out.data_ptr())))
        return out

    def sigma_at_tau(self, R, tau_z, ln_tau_size, k=None, k_per_decade=80.0, cb=False):
        """sigma(R [Mpc], z) of the linear matter field at the redshift whose conformal time is tau_z (cpt_sigma_at_tau)"""
        k = np.ascontiguousarray(self.inp.k if k is None else k, dtype=np.float64)
        out = C.c_double()
        self._fence()
        self._check(self.lib.cpt_sigma_at_tau(self.h, C.byref(self.inp.spectra), _dptr(k), k.size, int(ln_tau_size), float(tau_z), int(bool(cb)), float(R),
                                              float(k_per_decade), C.byref(out)))
        return out.value

    def get_sources(self, ntau, nk):
        out = torch.empty((self.inp.config.tp_size, ntau, nk), dtype=torch.float64, device=self.device)
        self._fence()
        self._check(self.lib.cpt_get_sources(self.h, C.c_void_p(out.data_ptr())))
        return out

    def kernel_ms(self, stage):
        ms, n = C.c_double(), C.c_int()
        self._check(self.lib.cpt_last_kernel_ms(self.h, stage, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def transfer_work(self):
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._check(self.lib.cpt_last_transfer_work(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- unit-test hooks ----
    def dbg_bessel(self, l, xmax, cap_nx=20000):
        l = np.ascontiguousarray(l, dtype=np.int32)
        nx = C.c_int()
        phi = np.zeros((l.size, cap_nx))
        dphi = np.zeros((l.size, cap_nx))
        chi = np.zeros(l.size)
        self._check(self.lib.cpt_dbg_bessel(self.h, _iptr(l), l.size, float(xmax), C.byref(nx), _dptr(phi), _dptr(dphi),
                                            _dptr(chi), cap_nx))
        n = nx.value
        return phi.reshape(-1)[: l.size * n].reshape(l.size, n), dphi.reshape(-1)[: l.size * n].reshape(l.size, n), chi

    def dbg_lookup(self, tau):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        out = np.zeros((tau.size, 16))
        self._check(self.lib.cpt_dbg_lookup(self.h, _dptr(tau), tau.size, _dptr(out)))
        return out

    def dbg_derivs(self, k, tau, tca_on, rsa_on, ufa_on, y):
        yy = np.zeros(64)
        yy[: len(y)] = y
        y = yy
        dy = np.zeros(64)
        neq = C.c_int()
        self._check(self.lib.cpt_dbg_derivs(self.h, float(k), float(tau), int(tca_on), int(rsa_on), int(ufa_on), _dptr(y),
                                            _dptr(dy), C.byref(neq)))
        return dy[: neq.value].copy()

    def dbg_solve(self, k, tau, tca_on, rsa_on, ufa_on, hg, b, full=False):
        bb = np.zeros(64)
        bb[: len(b)] = b
        x = np.zeros(64)
        self._check(self.lib.cpt_dbg_solve(self.h, float(k), float(tau), int(tca_on), int(rsa_on), int(ufa_on), float(hg),
                                           _dptr(bb), _dptr(x)))
        return x.copy() if full else x[: len(b)].copy()

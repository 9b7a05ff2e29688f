"""ctypes binding of classpp_public_amd/host/libcpt_host.so (include/cpt_host.h): the host-side grid builders."""
import ctypes as C
import os

import numpy as np

from .capi import CptConfig, CptTables

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libcpt_host.so")
_d, _i = C.c_double, C.c_int


class CptGridParams(C.Structure):
    """struct cpt_grid_params (include/cpt_host.h)"""
    _fields_ = [
        ("k_min_tau0", _d), ("k_max_tau0_over_l_max", _d), ("k_step_sub", _d), ("k_step_super", _d),
        ("k_step_transition", _d), ("k_step_super_reduction", _d), ("k_per_decade_for_pk", _d),
        ("k_per_decade_for_bao", _d), ("k_bao_center", _d), ("k_bao_width", _d),
        ("has_cls", _i), ("has_pk_matter", _i), ("l_scalar_max", _i),
        ("k_max_for_pk", _d), ("rs_rec", _d), ("tau_ini_thermo", _d),
        ("start_sources_at_tau_c_over_tau_h", _d), ("perturb_sampling_stepsize", _d),
        ("l_linstep", _d), ("l_logstep", _d), ("q_linstep", _d), ("q_logstep_spline", _d), ("q_logstep_open", _d),
        ("l_tensor_max", _i), ("q_logstep_trapzd", _d), ("q_numstep_transition", _d),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        pc, pt, pg = C.POINTER(CptConfig), C.POINTER(CptTables), C.POINTER(CptGridParams)
        pd, pi = C.POINTER(_d), C.POINTER(_i)
        L.cpt_host_k_list.argtypes = [pc, pg, pd, _i, pi, pi, pi]
        L.cpt_host_tau_sampling.argtypes = [pc, pt, pg, pd, _i, pi]
        L.cpt_host_l_list.argtypes = [pc, pg, pi, _i, pi]
        L.cpt_host_q_list.argtypes = [pc, pg, _d, _d, pd, _i, pi]
        L.cpt_host_error.restype = C.c_char_p
        _lib = L
    return _lib


def grid_params(inp):
    """cpt_grid_params of a named configuration (values dumped from the reference's precision / perturbs structs)."""
    d, t = inp.d, inp.t
    g = CptGridParams()
    for f in ("k_min_tau0", "k_max_tau0_over_l_max", "k_step_sub", "k_step_super", "k_step_transition",
              "k_step_super_reduction", "k_per_decade_for_pk", "k_per_decade_for_bao", "k_bao_center", "k_bao_width",
              "start_sources_at_tau_c_over_tau_h", "perturb_sampling_stepsize", "l_linstep", "l_logstep", "q_linstep",
              "q_logstep_spline", "q_logstep_open"):
        setattr(g, f, float(d["ppr." + f].reshape(-1)[0]))
    g.has_cls = 1 if inp.has_cls else 0
    g.has_pk_matter = int(d["ppt.has_pk_matter"][0])
    g.l_scalar_max = int(d["ppt.l_scalar_max"][0])
    g.k_max_for_pk = float(d["ppt.k_max_for_pk"][0])
    g.rs_rec = float(t["th.rs_rec"][0])
    g.tau_ini_thermo = float(t["th.tau_ini"][0])
    # tensors: l_max of the mode (older fixtures do not dump ppt.l_tensor_max: the l list ends exactly at it)
    g.l_tensor_max = int(d["ppt.l_tensor_max"][0]) if "ppt.l_tensor_max" in d else (int(inp.l[-1]) if inp.config.mode == 1 and inp.has_cls else 0)
    g.q_logstep_trapzd = float(d["ppr.q_logstep_trapzd"].reshape(-1)[0]); g.q_numstep_transition = float(d["ppr.q_numstep_transition"].reshape(-1)[0])
    return g


def _check(rc):
    if rc != 0:
        raise ValueError(lib().cpt_host_error().decode())


def k_list(inp, g=None):
    g = g or grid_params(inp)
    out = np.zeros(100000)
    n, ncl, ncmb = _i(), _i(), _i()
    _check(lib().cpt_host_k_list(C.byref(inp.config), C.byref(g), out.ctypes.data_as(C.POINTER(_d)), out.size, C.byref(n),
                                 C.byref(ncl), C.byref(ncmb)))
    return out[: n.value].copy(), ncl.value, ncmb.value


def tau_sampling(inp, g=None):
    g = g or grid_params(inp)
    out = np.zeros(100000)
    n = _i()
    _check(lib().cpt_host_tau_sampling(C.byref(inp.config), C.byref(inp.tables), C.byref(g), out.ctypes.data_as(C.POINTER(_d)),
                                       out.size, C.byref(n)))
    return out[: n.value].copy()


def l_list(inp, g=None):
    g = g or grid_params(inp)
    out = np.zeros(10000, dtype=np.int32)
    n = _i()
    _check(lib().cpt_host_l_list(C.byref(inp.config), C.byref(g), out.ctypes.data_as(C.POINTER(_i)), out.size, C.byref(n)))
    return out[: n.value].copy()


def q_list(inp, k_min, k_max_cl, g=None):
    g = g or grid_params(inp)
    out = np.zeros(1000000)
    n = _i()
    _check(lib().cpt_host_q_list(C.byref(inp.config), C.byref(g), float(k_min), float(k_max_cl),
                                 out.ctypes.data_as(C.POINTER(_d)), out.size, C.byref(n)))
    return out[: n.value].copy()

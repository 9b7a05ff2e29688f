"""From cosmological parameters to C_l and P(k) with nothing handed over from the reference (SURVEY S8f-1 closed):

    parameters --host--> background table (ndf15 in ln a) --host--> thermodynamics table (RECFAST + reionization)
               --host--> k, tau, l, q grids --GPU--> sources S(k,tau) --GPU--> Delta_l(q) --GPU--> C_l (+ lensing), P(k)

`ParameterInputs(name)` reads ONLY parameters and flags of a configuration (the values of the reference's input structs, plus YHe
and the reionization redshift / optical depth from the .ini) and computes every table and grid the hot path consumes with
classpp_public_amd/host/ (libcpt_host.so).  What stays outside: .ini parsing, the BBN helium table, shooting, non-cold species in
the background (their momentum quadrature), HyRec.
"""
import os

import numpy as np

from . import hostlib
from .inputs import GOLDEN, Inputs


def read_ini(path):
    out = {}
    for line in open(path):
        line = line.split("#")[0].strip()
        if "=" in line:
            k, v = line.split("=", 1)
            out[k.strip()] = v.strip()
    return out


class ParameterInputs(Inputs):
    """Inputs whose spline tables and sampling grids are computed on the host from the cosmological parameters."""

    def __init__(self, name, golden_dir=GOLDEN):
        self.name = name
        self.d = dict(np.load(os.path.join(golden_dir, name + ".npz")))
        ini = read_ini(os.path.join(golden_dir, name + ".ini"))
        cp = hostlib.cosmo_params(self)                       # struct background values (pba.* of the dump)
        tp = hostlib.CptThermoParams()
        hostlib.lib().cpt_host_thermo_defaults.argtypes = [hostlib.C.POINTER(hostlib.CptThermoParams)]
        hostlib.lib().cpt_host_thermo_defaults.restype = None
        hostlib.lib().cpt_host_thermo_defaults(hostlib.C.byref(tp))
        tp.YHe = float(ini["YHe"])
        tp.reio_parametrization = int(self.d["pth.reio_parametrization"][0])
        if "tau_reio" in ini:
            tp.reio_from_tau, tp.tau_reio = 1, float(ini["tau_reio"])
        else:
            tp.reio_from_tau, tp.z_reio = 0, float(ini["z_reio"])
        tables = {}
        tables.update(hostlib.background(self, cp))
        tables.update(hostlib.thermodynamics(self, cp, tp))
        super().__init__(name, golden_dir, tables=tables)
        self.l_tensor_max = int(ini["l_max_tensors"]) if "l_max_tensors" in ini else None
        # the sampling grids, rebuilt from the tables just computed (never read from the fixture)
        self.k, self.k_size_cl, _ = hostlib.k_list(self)
        self.tau = hostlib.tau_sampling(self)
        if self.has_cls:
            self.l = hostlib.l_list(self)
            self.q = hostlib.q_list(self, self.k[0], self.k[self.k_size_cl - 1])

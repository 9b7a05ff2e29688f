#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY.

The reference against ITSELF: how far do its own outputs move when its integration tolerance is halved?

    python oracle/make_noise_fixtures.py [cfg ...]        # default: lcdm long_full ncdm

For every configuration the unmodified reference (oracle/_ref, built by oracle/Makefile) is run on tests/golden/<cfg>.ini with
`tol_perturb_integration = 5e-6` appended (default 1e-5: include/precisions.h:237) and compared with the committed fixture of the same
.ini at the default tolerance.  What is kept (tests/golden/noise_<cfg>.npz, data only):

  src_dev_max[tp][k], src_dev_rms[tp][k]   max / rms over tau of |S_halved - S_default|, relative to max_tau |S_default|, for EVERY k
  transfer_dev[tt]                         max over (l, q) of |Delta_halved - Delta_default| relative to the maximum over q of the row
  cl_dev[ct], pk_dev[k], sigma8_dev        relative moves of the C_l table (cross spectra: of max |C_l|), of P(k, z = 0), of sigma8
  steps_default, steps_halved              (not available from the reference: left out)

Both runs are equally valid answers to "integrate to rtol"; a restatement or a GPU kernel whose step sequence differs from the
reference's lands anywhere inside that band.  tests/test_noise_floor.py derives the source / transfer / P(k) bands used by the parity
tests from these files instead of from prose.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import DRIVER, GOLD, load_bin  # noqa: E402


def run(ini_text, name):
    with tempfile.TemporaryDirectory() as td:
        ini = os.path.join(td, name + ".ini")
        open(ini, "w").write(ini_text)
        out = os.path.join(td, name + ".bin")
        subprocess.check_call([DRIVER, "dump", ini, out], cwd=GOLD)
        return load_bin(out)


def rel_dev(a, b, axis):
    scale = np.max(np.abs(b), axis=axis, keepdims=True)
    scale[scale == 0] = 1.0
    d = np.abs(a - b) / scale
    return d.max(axis=axis), np.sqrt(np.mean(((a - b) / scale) ** 2, axis=axis))


def main():
    cfgs = sys.argv[1:] or ["lcdm", "long_full", "ncdm"]
    for cfg in cfgs:
        text = open(os.path.join(GOLD, cfg + ".ini")).read()
        assert "tol_perturb_integration" not in text
        base = run(text, cfg)
        half = run(text + "\ntol_perturb_integration = 5.e-6\n", cfg + "_half")
        # (the default-tolerance run must BE the committed fixture: same reference, same .ini)
        fix = np.load(os.path.join(GOLD, cfg + ".npz"))
        assert np.array_equal(base["sp.cl_table"], fix["sp.cl_table"]), "the rebuilt reference no longer reproduces tests/golden/%s.npz" % cfg
        out = {}
        s0, s1 = base["pt.sources"], half["pt.sources"]          # [tp][tau][k]
        out["src_dev_max"], out["src_dev_rms"] = rel_dev(s1, s0, axis=1)
        t0, t1 = base["tr.transfer"], half["tr.transfer"]        # [tt][l][q]
        dmax, _ = rel_dev(t1, t0, axis=2)
        out["transfer_dev"] = dmax.max(axis=1)
        c0, c1 = base["sp.cl_table"], half["sp.cl_table"]        # [l][ct]
        cl_dev = np.zeros(c0.shape[1])
        for ct in range(c0.shape[1]):
            cross = ct in [int(base[k][0]) for k in ("sp.index_ct_te", "sp.index_ct_tp", "sp.index_ct_ep") if k in base and int(base[k][0]) >= 0]
            col0, col1 = c0[:, ct], c1[:, ct]
            if not np.any(col0):
                continue
            cl_dev[ct] = np.max(np.abs(col1 - col0)) / np.max(np.abs(col0)) if cross else np.max(np.abs(col1 / col0 - 1))
        out["cl_dev"] = cl_dev
        if "nl.pk_lin_z0" in base:
            out["pk_dev"] = np.abs(half["nl.pk_lin_z0"] / base["nl.pk_lin_z0"] - 1)
            out["sigma8_dev"] = np.abs(half["nl.sigma8"] / base["nl.sigma8"] - 1)
        for key in ("pt.index_tp_t0", "pt.index_tp_t1", "pt.index_tp_t2", "pt.index_tp_p", "pt.index_tp_delta_m", "pt.index_tp_phi_plus_psi",
                    "pt.index_tp_delta_cb") + tuple("pt.index_tp_" + n for n in ("delta_tot", "delta_g", "delta_b", "delta_cdm", "delta_ur", "theta_tot", "theta_g",
                                                                                         "theta_b", "theta_cdm", "theta_ur", "phi", "psi", "delta_ncdm1", "theta_ncdm1")):
            if key in base:
                out[key] = base[key]
        out["tol_default"] = np.array([1e-5]); out["tol_halved"] = np.array([5e-6])
        np.savez_compressed(os.path.join(GOLD, "noise_%s.npz" % cfg), **out)
        print(cfg, "sources (max over k) per type:", np.array2string(out["src_dev_max"].max(axis=1), precision=2),
              "| transfer", np.array2string(out["transfer_dev"], precision=2), "| C_l", np.array2string(cl_dev, precision=2),
              "| P(k)", "%.2e" % out["pk_dev"].max() if "pk_dev" in out else "-")


if __name__ == "__main__":
    main()

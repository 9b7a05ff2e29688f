#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY.

Generates the committed golden fixtures under tests/golden/ from the UNMODIFIED reference, compiled by
oracle/Makefile into oracle/_ref/ (never committed) and driven by oracle/ref_driver.cpp.

    python oracle/make_fixtures.py            # rebuilds every tests/golden/*.npz from tests/golden/*.ini

A fixture is data only: hot-path inputs (background/thermodynamics spline tables, grids, parameters) and
expected outputs (source functions, transfer functions, C_l, P(k)) for one .ini.  Large outputs of the two
full-size configs are sub-sampled (16 k columns of sources_, 32 q columns + 8 l rows of transfer_); the
`small` config keeps everything so that each stage can be tested in isolation on the GPU box.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(HERE, "_ref", "ref_driver")


def load_bin(path):
    d = {}
    with open(path, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            parts = line.decode().split()
            name, dt, nd = parts[0], parts[1], int(parts[2])
            shape = [int(x) for x in parts[3:3 + nd]]
            n = int(np.prod(shape))
            dtype = np.float64 if dt == "f8" else np.int32
            d[name] = np.frombuffer(f.read(n * dtype().itemsize), dtype=dtype).reshape(shape).copy()
    return d


def run_reference(cfg, tmpdir="/tmp"):
    ini = os.path.join(GOLD, cfg + ".ini")
    out = os.path.join(tmpdir, "cpt_ref_%s.bin" % cfg)
    subprocess.check_call([DRIVER, "dump", ini, out], cwd=GOLD)
    d = load_bin(out)
    os.remove(out)
    return d


TABLE_KEYS = ("bg.", "th.", "ncdm.")


def pick_k_subset(nk, n=16):
    idx = np.unique(np.round(np.linspace(0, nk - 1, n)).astype(int))
    return idx


def main():
    cfgs = sys.argv[1:] or ["small", "lcdm", "explanatory"]
    tables_written = False
    for cfg in cfgs:
        d = run_reference(cfg)
        out = {}
        tables = {}
        for k, v in d.items():
            if k.startswith(TABLE_KEYS):
                tables[k] = v
            elif k in ("pt.sources", "tr.transfer"):
                pass
            else:
                out[k] = v
        src = d["pt.sources"]  # [tp][tau][k]
        if cfg in ("small", "tens", "tens_curved", "ncdm_small", "ncdm3_small", "long_small", "tca_mb", "ncdm_permille_small", "small_tk", "newt_tk"):
            out["pt.sources"] = src
            if "tr.transfer" in d:
                out["tr.transfer"] = d["tr.transfer"]
        else:
            nk = src.shape[2]
            ks = pick_k_subset(nk, 16)
            out["pt.sources_k_index"] = ks.astype(np.int32)
            out["pt.sources_subset"] = np.ascontiguousarray(src[:, :, ks])
            # delta_m(k, tau0) column (P(k) input), all k
            if int(d["pt.index_tp_delta_m"][0]) >= 0:
                out["pt.delta_m_today"] = np.ascontiguousarray(src[int(d["pt.index_tp_delta_m"][0]), -1, :])
            if "tr.transfer" in d:
                t = d["tr.transfer"]  # [tt][l][q]
                nq, nl = t.shape[2], t.shape[1]
                qs = np.unique(np.round(np.linspace(0, nq - 1, 32)).astype(int))
                ls = np.unique(np.round(np.linspace(0, nl - 1, 8)).astype(int))
                out["tr.transfer_q_index"] = qs.astype(np.int32)
                out["tr.transfer_l_index"] = ls.astype(np.int32)
                out["tr.transfer_at_q"] = np.ascontiguousarray(t[:, :, qs])
                out["tr.transfer_at_l"] = np.ascontiguousarray(t[:, ls, :])
        if cfg in ("curved_full", "tens_curved"):
            old = np.load(os.path.join(GOLD, "tables_curved.npz"))
            for k in tables:
                assert np.array_equal(old[k], tables[k]), "tables differ between configs: " + k
        if cfg.startswith("iso_") or cfg in ("newt", "tens", "explanatory_mpk", "newt_full", "tens_full", "long_small", "long_full", "tca_mb", "lcdm_zpk",
                                             "lcdm_tk", "small_tk", "newt_tk", "lcdm_zpk_tk"):
            # same cosmology as small/lcdm/explanatory: the tables must be the committed ones
            old = np.load(os.path.join(GOLD, "tables_lcdm.npz"))
            for k in tables:
                assert np.array_equal(old[k], tables[k]), "tables differ between configs: " + k
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **out)
            print(cfg, {k: v.shape for k, v in out.items() if v.size > 1000})
            continue
        if cfg in ("curved_full", "tens_curved"):
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **out)
            print(cfg, {k: v.shape for k, v in out.items() if v.size > 1000})
            continue
        if cfg == "ncdm3_st":
            # BASELINE config 4 as the reference runs it: scalars + tensors in one run.  Only the totals at every integer l are kept
            # (unlensed and lensed); the per-mode inputs and outputs are those of ncdm3 (scalars) and ncdm3_tens (tensors).
            keep = {k: v for k, v in d.items() if k.startswith(("sp.cl_", "le.cl_")) and k not in ("sp.cl_table", "le.cl_lens")}
            for k in ("sp.l_max_tot", "le.l_lensed_max"):
                keep[k] = d[k]
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **keep)
            print(cfg, {k: v.shape for k, v in keep.items() if v.size > 10})
            continue
        if cfg.startswith("sc_"):
            # classy-level scenarios (tests/test_classy.py): what a user of the reference's Python wrapper reads, nothing else
            keep = {k: v for k, v in d.items() if k.startswith(("sp.cl_", "le.cl_")) and k not in ("sp.cl_table", "le.cl_lens")}
            for k in ("sp.l_max_tot", "le.l_lensed_max", "nl.pk_lin_z0", "nl.sigma8", "pt.k", "pba.h"):
                if k in d:
                    keep[k] = d[k]
            for k in ("th.z_reionization", "th.tau_reionization", "th.z_rec", "th.rs_rec", "th.ra_rec", "bg.age", "bg.conformal_age", "bg.Neff", "bg.Omega0_m"):
                if k in tables:
                    keep[k] = np.atleast_1d(tables[k])
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **keep)
            print(cfg, {k: v.shape for k, v in keep.items() if v.size > 10})
            continue
        if cfg == "lcdm_taureio":
            # reionization given by its optical depth (bisection of th.cpp:2222-2318): only the thermodynamics outcome is kept
            # (scalars + every 40th row of the table), the cosmology is that of lcdm.ini
            keep = {k: v for k, v in tables.items() if k.startswith("th.") and v.size == 1}
            keep["th.row_index"] = np.arange(0, tables["th.z_table"].size, 40).astype(np.int32)
            keep["th.z_table_rows"] = tables["th.z_table"][keep["th.row_index"]]
            keep["th.thermodynamics_table_rows"] = tables["th.thermodynamics_table"][keep["th.row_index"]]
            keep["pth.tau_reio"] = np.array([0.0925])
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **keep)
            print(cfg, {k: v.shape for k, v in keep.items() if v.size > 10})
            continue
        if cfg.startswith("ncdm"):
            # massive neutrinos (BASELINE configs 3, 4): one table file per cosmology (1 species / 3 species)
            tname = "tables_ncdm3.npz" if cfg.startswith("ncdm3") else "tables_ncdm1.npz"
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **out)
            if cfg in ("ncdm_small", "ncdm3_small"):
                np.savez_compressed(os.path.join(GOLD, tname), **tables)
            elif cfg.startswith("ncdm_permille"):
                # (l_max_ncdm enters the momentum sampling test of the reference only through tolerances, not the tables: same cosmology as ncdm)
                old = np.load(os.path.join(GOLD, tname))
                for k in tables:
                    if not k.startswith("ncdm."):
                        assert np.array_equal(old[k], tables[k]), "tables differ between configs: " + k
            else:
                old = np.load(os.path.join(GOLD, tname))
                for k in tables:
                    assert np.array_equal(old[k], tables[k]), "tables differ between configs: " + k
            print(cfg, {k: v.shape for k, v in out.items() if v.size > 1000})
            continue
        if cfg in ("curved", "open"):
            # non-flat cosmology: its own table file; full sources and transfer table (small precision file)
            for key in [k for k in out if k.startswith(("tr.transfer_at", "pt.sources_subset", "pt.sources_k_index"))]:
                del out[key]
            out["pt.sources"] = src
            out["tr.transfer"] = d["tr.transfer"]
            np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **out)
            np.savez_compressed(os.path.join(GOLD, "tables_%s.npz" % cfg), **tables)
            print(cfg, {k: v.shape for k, v in out.items() if v.size > 1000})
            continue
        np.savez_compressed(os.path.join(GOLD, cfg + ".npz"), **out)
        # the three configs share one cosmology -> one table file; verified identical below
        tpath = os.path.join(GOLD, "tables_lcdm.npz")
        if not tables_written and cfg in ("small", "lcdm", "explanatory"):
            if os.path.exists(tpath) and len(cfgs) < 3:
                old = np.load(tpath)
                for k in tables:
                    assert np.array_equal(old[k], tables[k]), "tables differ between configs: " + k
            else:
                np.savez_compressed(tpath, **tables)
            tables_written = True
            ref_tables = tables
        else:
            for k in tables:
                assert np.array_equal(ref_tables[k], tables[k]), "tables differ between configs: " + k
        print(cfg, {k: v.shape for k, v in out.items() if v.size > 1000})


if __name__ == "__main__":
    main()

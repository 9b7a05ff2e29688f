#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY.

Imports the output files the reference repository itself carries for its explanatory.ini run
(output/explanatory00_cl.dat, output/explanatory00_cl_lensed.dat, written with `format = class`,
`headers = yes`: dimensionless l(l+1)/2pi C_l, columns l TT EE TE BB phiphi TPhi Ephi) into
tests/golden/ref_output_explanatory00.npz.  These are golden vectors of the reference's own making: numbers only,
no code.  The parameters of that run are those of output/explanatory00_parameters.ini = tests/golden/explanatory.ini.

    python oracle/import_reference_output.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
COLUMNS = ("l", "tt", "ee", "te", "bb", "pp", "tp", "ep")


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {}
    for key, name in (("cl", "explanatory00_cl.dat"), ("cl_lensed", "explanatory00_cl_lensed.dat")):
        a = np.loadtxt(os.path.join(ref, "output", name))
        assert a.shape[1] == len(COLUMNS) and a[0, 0] == 2 and np.all(np.diff(a[:, 0]) == 1)
        out[key] = a
    out["columns"] = np.array(COLUMNS)
    np.savez_compressed(os.path.join(GOLD, "ref_output_explanatory00.npz"), **out)
    print("wrote ref_output_explanatory00.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

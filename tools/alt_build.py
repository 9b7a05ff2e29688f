#!/usr/bin/env python3
"""Build an ALTERNATIVE libcpt (diagnostics, A/B experiments) beside the product library, reusing the product's objects for every
translation unit that is not named:
    python tools/alt_build.py <tag> [-DDEFINE ...] [--tu cpt_perturb_sets_bins2.hip ...]
 -> classpp_public_amd/csrc/libcpt_<tag>.so   (use with tools/kernel_time.py, tools/interval_run.py, tools/prof_run.py)
Default --tu: the three register-set units.  The product library must have been built (python __graft_entry__.py)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "classpp_public_amd", "csrc")
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    tag = sys.argv[1]
    defines = [a for a in sys.argv[2:] if a.startswith("-")]
    tus = []
    if "--tu" in sys.argv:
        tus = [a for a in sys.argv[sys.argv.index("--tu") + 1:] if not a.startswith("-")]
        defines = [a for a in defines if a != "--tu"]
    tus = tus or ["cpt_perturb_sets_tails.hip", "cpt_perturb_sets_bins2.hip", "cpt_perturb_sets_bins5.hip", "cpt_perturb_sets_both3.hip"]
    bdir = os.path.join(CSRC, "build", "alt_" + tag)
    os.makedirs(bdir, exist_ok=True)
    jobs, objs = [], []
    for s in G.HIP_SOURCES:
        if s in tus:
            obj = os.path.join(bdir, s.replace(".hip", ".o"))
            jobs.append((os.path.join(CSRC, s), obj, "/opt/rocm/bin/hipcc", defines))
        else:
            obj = os.path.join(CSRC, "build", s.replace(".hip", ".o"))
            assert os.path.exists(obj), "build the product library first: " + obj
        objs.append(obj)
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(G._hipcc_object, jobs))
    out = os.path.join(CSRC, "libcpt_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs + ["-ldl"])
    print(out)


if __name__ == "__main__":
    main()

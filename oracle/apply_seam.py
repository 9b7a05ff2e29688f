#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY.  Applies the reference-side patch described in include/reference_side/cpt_seam.cpp - ten inserted lines and two
edited loop headers in seven files - to a SCRATCH CHECKOUT of the reference's source/ tools/ include/ under /tmp (outside the repository: nothing of
the reference is copied into it).  `make -C oracle seam` compiles that checkout + cpt_seam.cpp (a changed class layout must be seen by every
translation unit, and a quoted #include finds the header next to the including file first - hence a whole checkout, not seven files) and links it
with libcpt_host.so into oracle/_ref/libclass_cpt.so + ref_driver_cpt: the unmodified downstream modules of the reference - PrimordialModule,
NonlinearModule, SpectraModule, LensingModule - running on the sources_ / transfer_ tables the GPU backend filled."""
import shutil
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cpt_seam"
if os.path.isdir(OUT):
    shutil.rmtree(OUT)
for sub in ("source", "tools", "include"):
    shutil.copytree(os.path.join(REF, sub), os.path.join(OUT, sub))


def patch(name, edits, sub="source"):
    s = open(os.path.join(REF, sub, name)).read()
    for kind, anchor, text, after in edits:
        start = s.index(after) if after else 0
        i = s.index(anchor, start)
        if kind == "insert_after":
            j = s.index("\n", i) + 1
            s = s[:j] + text + "\n" + s[j:]
        elif kind == "insert_before":
            j = s.rfind("\n", 0, i) + 1
            s = s[:j] + text + "\n" + s[j:]
        elif kind == "replace":
            s = s[:i] + text + s[i + len(anchor):]
    open(os.path.join(OUT, sub, name), "w").write(s)


patch("perturbations_module.h", [
    ("insert_before", "class PerturbationsModule : public BaseModule {", "namespace cpt { class PerturbationsModule; }   /* MI355X backend seam */", None),
    ("insert_after", "  double k_max_;", "  std::shared_ptr<const cpt::PerturbationsModule> cpt_gpu_;   /* MI355X backend seam: the GPU module whose sources fill sources_ */", None),
    ("insert_after", "  int perturb_init();", "  int cpt_fill_sources();   /* MI355X backend seam (cpt_seam.cpp) */", None),
])
patch("perturbations_module.cpp", [
    ("insert_after", "  Tools::TaskSystem task_system(pba->number_of_threads);", "  const bool cpt_done_ = cpt_fill_sources() == 1;   /* MI355X backend seam */", "int PerturbationsModule::perturb_init()"),
    ("replace", "for (index_md = 0; index_md < md_size_; index_md++) {", "for (index_md = 0; !cpt_done_ && index_md < md_size_; index_md++) {", "const bool cpt_done_ = cpt_fill_sources() == 1;"),
])
patch("transfer_module.h", [
    ("insert_after", "  int transfer_init();", "  int cpt_fill_transfer();   /* MI355X backend seam (cpt_seam.cpp) */", None),
])
patch("transfer_module.cpp", [
    ("insert_after", "  Tools::TaskSystem task_system(pba->number_of_threads);", "  const bool cpt_done_ = cpt_fill_transfer() == 1;   /* MI355X backend seam */", "int TransferModule::transfer_init()"),
    ("replace", "for (index_q = 0; index_q < q_size_; index_q++) {", "for (index_q = 0; !cpt_done_ && index_q < q_size_; index_q++) {", "const bool cpt_done_ = cpt_fill_transfer() == 1;"),
])
# the adapter (include/reference_side/cpt_adapter.h) reads spline tables and momentum grids that have no accessor: one friend declaration per class
FWD = "namespace cpt { struct Inputs; } class InputModule; class BackgroundModule; class ThermodynamicsModule;   /* MI355X backend seam */"
FRIEND = "  friend cpt::Inputs MakeCptInputs(const InputModule&, const BackgroundModule&, const ThermodynamicsModule&);   /* MI355X backend seam */"
patch("background_module.h", [
    ("insert_before", "class BackgroundModule : public BaseModule {", FWD, None),
    ("insert_after", "public:", FRIEND, "class BackgroundModule : public BaseModule {"),
])
patch("thermodynamics_module.h", [
    ("insert_before", "class ThermodynamicsModule : public BaseModule {", FWD, None),
    ("insert_after", "public:", FRIEND, "class ThermodynamicsModule : public BaseModule {"),
])
patch("non_cold_dark_matter.h", [
    ("insert_before", "class NonColdDarkMatter {", FWD, None),
    ("insert_after", "public:", FRIEND, "class NonColdDarkMatter {"),
], sub="tools")
print("patched scratch checkout under", OUT)


def build(out_dir, c_src, cpp_src):
    """compile the scratch checkout + cpt_seam.cpp in parallel and link oracle/_ref/libclass_cpt.so + ref_driver_cpt"""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    obj = os.path.join(OUT, "obj")
    os.makedirs(obj, exist_ok=True)
    inc = ["-I" + os.path.join(OUT, d) for d in ("include", "tools", "source")] + ["-I" + os.path.join(REF, "main")]
    flags = ["-O3", "-g", "-fPIC", "-D__CLASSDIR__=\"%s\"" % REF]
    jobs = []
    for f in c_src:
        jobs.append(["gcc"] + flags + inc + ["-c", os.path.join(OUT, f), "-o", os.path.join(obj, os.path.basename(f)[:-2] + ".o")])
    for f in cpp_src:
        jobs.append(["g++", "-std=c++17"] + flags + inc + ["-c", os.path.join(OUT, f), "-o", os.path.join(obj, os.path.basename(f)[:-4] + ".opp")])
    jobs.append(["g++", "-std=c++17"] + flags + inc + ["-I" + os.path.join(root, "include"), "-c", os.path.join(root, "include", "reference_side", "cpt_seam.cpp"),
                 "-o", os.path.join(obj, "cpt_seam.opp")])

    def run(cmd):
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(" ".join(cmd) + "\n" + p.stderr)
            raise SystemExit(1)
    with ThreadPoolExecutor(max_workers=8) as pool:
        list(pool.map(run, jobs))
    host, csrc = os.path.join(root, "classpp_public_amd", "host"), os.path.join(root, "classpp_public_amd", "csrc")
    rpath = ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../classpp_public_amd/host", "-Wl,-rpath,$ORIGIN/../../classpp_public_amd/csrc"]
    objs = [j[-1] for j in jobs]
    run(["g++", "-shared", "-fPIC", "-o", os.path.join(out_dir, "libclass_cpt.so")] + objs + ["-L" + host, "-lcpt_host", "-L" + csrc, "-lcpt"] + rpath + ["-lm", "-lpthread"])
    run(["g++", "-std=c++17"] + flags + inc + ["-o", os.path.join(out_dir, "ref_driver_cpt"), os.path.join(here, "ref_driver.cpp"), "-L" + out_dir, "-lclass_cpt",
         "-L" + host, "-lcpt_host", "-L" + csrc, "-lcpt"] + rpath + ["-lm", "-lpthread"])
    print("built", os.path.join(out_dir, "ref_driver_cpt"))


if len(sys.argv) > 3 and sys.argv[3] == "build":
    build(sys.argv[4], sys.argv[5].split(), sys.argv[6].split())

"""CPU-side checks of the drop-in boundary: the in-tree HIP library loads, exports every symbol that include/cpt.h
declares, validates its inputs before touching a device, and fails LOUDLY (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from classpp_public_amd import capi
from classpp_public_amd.inputs import Inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cpt_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = capi.lib()
    names = declared_symbols()
    assert set(names) == set(capi.EXPORTS), (set(names) ^ set(capi.EXPORTS))
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layout_matches_header():
    """field order of the ctypes mirror == field order of the C struct (parsed from the header)"""
    src = open(os.path.join(ROOT, "include", "cpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, cls in (("cpt_config", capi.CptConfig), ("cpt_tables", capi.CptTables), ("cpt_stepstat", capi.CptStepstat)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(const\s+)?(double|int)\s*\*?\s*(.*)", decl, flags=re.S)
            assert m, decl
            for name in m.group(3).split(","):
                fields.append(re.sub(r"\[.*?\]", "", name).strip().lstrip("*").strip())   # (array declarators dropped)
        assert fields == [f[0] for f in cls._fields_], cname


def test_struct_sizes_and_offsets_match_the_compiled_header(tmp_path):
    """sizeof / offsetof of EVERY struct that crosses the C ABI, as gcc lays them out from include/cpt.h, against the ctypes mirrors -
    including cpt_step_io (the argument block of the fused cpt_step that bench.py times), cpt_spectra_params and cpt_lensing_params"""
    import subprocess
    pairs = (("cpt_config", capi.CptConfig), ("cpt_tables", capi.CptTables), ("cpt_stepstat", capi.CptStepstat),
             ("cpt_spectra_params", capi.CptSpectraParams), ("cpt_lensing_params", capi.CptLensingParams), ("cpt_step_io", capi.CptStepIo))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cpt.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in [(f[0], f[1]) for f in cls._fields_]:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), (cname, got[cname], C.sizeof(cls))
        for f in cls._fields_:
            assert int(got["%s.%s" % (cname, f[0])]) == getattr(cls, f[0]).offset, (cname, f[0])


def test_unsupported_physics_is_rejected_before_any_device_work():
    inp = Inputs("small")
    lib = capi.lib()
    for field, value, code in (("sgnK", 1, capi.CPT_ERR_INVALID), ("has_fld", 1, capi.CPT_ERR_UNSUPPORTED),
                               ("gauge", 7, capi.CPT_ERR_INVALID), ("l_max_g", 3, capi.CPT_ERR_INVALID),
                               ("tp_size", 0, capi.CPT_ERR_INVALID)):
        cfg = capi.CptConfig.from_buffer_copy(inp.config)
        setattr(cfg, field, value)
        h = C.c_void_p()
        rc = lib.cpt_create(C.byref(cfg), C.byref(inp.tables), C.byref(h))
        assert rc == code, (field, rc)
        assert not h.value
        assert len(lib.cpt_create_error()) > 0


def test_no_cpu_fallback():
    """Without a GPU the product path must refuse to run (never route through the oracle or any CPU code)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    inp = Inputs("small")
    lib = capi.lib()
    h = C.c_void_p()
    rc = lib.cpt_create(C.byref(inp.config), C.byref(inp.tables), C.byref(h))
    assert rc == capi.CPT_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.cpt_create_error()
    from classpp_public_amd.backend import Backend, CptError
    with pytest.raises(CptError):
        Backend(inp)
    # the product sources never reference the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "classpp_public_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "libcpt_oracle" not in txt and "oracle/" not in txt.replace("oracle/make_fixtures.py", ""), f


def test_host_header_symbols_and_struct_layouts():
    """include/cpt_host.h (libcpt_host.so): every declared function is exported, and the ctypes mirrors of the structs have the
    header's field order"""
    from classpp_public_amd import hostlib
    src = open(os.path.join(ROOT, "include", "cpt_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(cpt_host_[a-z_0-9]+)\s*\(", src)))
    assert len(names) >= 12
    lib = hostlib.lib()
    for n in names:
        assert hasattr(lib, n), n
    for cname, cls in (("cpt_grid_params", hostlib.CptGridParams), ("cpt_cosmo_params", hostlib.CptCosmoParams),
                       ("cpt_background", hostlib.CptBackground), ("cpt_thermo_params", hostlib.CptThermoParams),
                       ("cpt_thermo", hostlib.CptThermo)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(const\s+)?(double|int)\s*\*?\s*(.*)", decl, flags=re.S)
            assert m, decl
            for name in m.group(3).split(","):
                fields.append(re.sub(r"\[.*?\]", "", name).strip().lstrip("*").strip())
        assert fields == [f[0] for f in cls._fields_], (cname, [a for a, b in zip(fields, [f[0] for f in cls._fields_]) if a != b][:3])


def test_cxx_boundary_headers_compile_standalone(tmp_path):
    """include/cpt.h is C (compiled as C11), include/cpt_host.h + cpt_modules.hpp are the C++ boundary: each must compile on its own,
    and the shim demo (the reference-side usage of tests/test_gpu_host_shim.py) must compile against them - without a GPU."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    c_src = tmp_path / "abi.c"
    c_src.write_text('#include "cpt.h"\n#include "cpt_host.h"\nint main(void) { cpt_config c; cpt_grid_params g; (void)c; (void)g; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I" + inc, str(c_src)])
    cpp_src = tmp_path / "shim.cpp"
    cpp_src.write_text('#include "cpt_modules.hpp"\nint main() { cpt::Inputs in{}; return in.n_ic == 1 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + inc, str(cpp_src)])
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", os.path.join(ROOT, "tests", "host_shim_demo.cpp")])

"""classpp_public_amd -- MI355X-native backend for the perturbations -> transfer hot path of CLASS++.

csrc/      hand-written HIP kernels (gfx950) + the C ABI declared in include/cpt.h  -> csrc/libcpt.so
capi.py    ctypes mirror of the C ABI
inputs.py  hot-path inputs (tables, grids, parameters) for the named configurations
modules.py host-side mirror of the reference's PerturbationsModule / TransferModule data contract
"""

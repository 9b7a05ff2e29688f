// cpt_modules.hpp -- C++ shim classes with the constructor semantics and the PUBLIC DATA CONTRACT of the reference's
// PerturbationsModule (source/perturbations_module.h:9-178) and TransferModule (source/transfer_module.h:9-57), built
// on the C ABI of include/cpt.h (GPU kernels) and include/cpt_host.h (grids).  Same member names, same array layouts,
// same ownership (malloc'd by the module, freed in its destructor, consumers never write), same error behaviour: all
// work happens in the constructor; unsupported / inconsistent input throws std::invalid_argument, a failure inside the
// computation throws std::runtime_error (pm.cpp:37-39, tm.cpp:43-45; classy maps them to CosmoSevereError /
// CosmoComputationError).  What differs by necessity: the constructors take a cpt::Inputs (flat POD config + spline
// tables + grid parameters) instead of InputModulePtr / BackgroundModulePtr / ThermodynamicsModulePtr; INTEGRATION.md
// shows the adapter that fills it from those reference modules.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>

#include "cpt_host.h"

namespace cpt {

// One process per GPU: rank `rank` of `world` integrates the k-modes k_[rank], k_[rank + world], ... and projects the
// multipoles l_[rank], l_[rank + world], ...; the two exchanges run inside libcpt.so (RCCL, include/cpt.h).  comm_id: the
// CPT_COMM_ID_BYTES that rank 0 obtained from cpt_comm_get_unique_id() and handed to every rank (a file, MPI, any launcher).
// After construction every rank's PerturbationsModule holds the FULL sources_ (the transfer stage needs every k); the full transfer_
// table is assembled on rank 0, the other ranks keep their own multipoles (rows of the others are zero).
struct Shard {
  int rank = 0, world = 1;
  const void* comm_id = nullptr;
};

struct Inputs {
  cpt_config config;      // physics, precision and index maps of the first mode (scalars when present, else tensors); config.ic: its
                          // initial condition when n_ic == 1
  cpt_tables tables;      // host pointers into the caller's (reference modules') tables; only read during construction
  cpt_grid_params grid;
  Shard shard;            // default: single GPU
  // ---- more than one initial condition / mode (perturb_indices_of_perturbs, pm.cpp:843-1235) ----
  // scalar initial conditions in the reference's order ad, bi, cdi, nid, niv (pm.cpp:1153-1170): ic_size_[scalars] = n_ic.  Every
  // initial condition is its own device handle (the modes of different initial conditions are independent integrations).
  int n_ic = 1;
  int ic[5] = {CPT_IC_AD, CPT_IC_BI, CPT_IC_CDI, CPT_IC_NID, CPT_IC_NIV};   // read when n_ic > 1
  // modes = s,t: md_size_ = 2, index_md_scalars_ = 0, index_md_tensors_ = 1; config_tensors is the tensor mode's own config
  // (mode = CPT_MODE_TENSORS, its source / transfer index maps, l_max_*_ten, ...)
  bool with_tensors = false;
  cpt_config config_tensors;
};

// Background + thermodynamics tables computed on the host (include/cpt_host.h) instead of taken from the
// reference's BackgroundModule / ThermodynamicsModule: owns the tables, fills the table part of an Inputs and the scalars of
// cpt_config / cpt_grid_params that derive from them.  Throws like the modules (std::invalid_argument / std::runtime_error).
class HostTables {
 public:
  HostTables(const cpt_cosmo_params& cosmo, const cpt_thermo_params& thermo);
  ~HostTables();
  HostTables(const HostTables&) = delete;
  void fill(Inputs& in) const;   // in.tables (pointers into this object: keep it alive), in.config.{tau0,...}, in.grid.{rs_rec,...}
  cpt_background background{};
  cpt_thermo thermo{};
};

class PerturbationsModule {
 public:
  explicit PerturbationsModule(const Inputs& in);
  ~PerturbationsModule();
  PerturbationsModule(const PerturbationsModule&) = delete;

  // ---- data contract of source/perturbations_module.h ----
  int index_md_scalars_ = 0, index_md_tensors_ = 0, md_size_ = 1;   // (tensors-only: both 0, like the reference's index counter)
  short has_scalars_ = 1, has_tensors_ = 0;
  int index_ic_ad_ = -1, index_ic_bi_ = -1, index_ic_cdi_ = -1, index_ic_nid_ = -1, index_ic_niv_ = -1, index_ic_ten_ = 0;
  int* ic_size_ = nullptr;                                          // [md]
  int index_tp_t0_ = -1, index_tp_t1_ = -1, index_tp_t2_ = -1, index_tp_p_ = -1, index_tp_delta_m_ = -1,
      index_tp_phi_plus_psi_ = -1, index_tp_delta_cb_ = -1;   // (delta_cb: requested with delta_m when ncdm is present, pm.cpp:996)
  int* tp_size_ = nullptr;                                          // [md]
  short has_source_t_ = 0, has_source_p_ = 0, has_source_delta_m_ = 0, has_source_phi_plus_psi_ = 0;
  double*** sources_ = nullptr;  // sources_[md][ic*tp_size+tp][index_tau*k_size+index_k]
  double* ln_tau_ = nullptr;
  int ln_tau_size_ = 1;
  double* tau_sampling_ = nullptr;
  int tau_size_ = 0;
  int* k_size_cl_ = nullptr;
  int* k_size_ = nullptr;
  double** k_ = nullptr;
  double k_min_ = 0., k_max_ = 0.;
  mutable char error_message_[2048];

  // ---- extras of this backend ----
  // device handle of (mode, initial condition) holding its sources resident in HBM (k-major); handle() = the first one
  cpt_handle* handle(int index_md = 0, int index_ic = 0) const { return h_[index_md][index_ic]; }
  const cpt_stepstat* stepstat() const { return stats_; }  // per-k work counters (the evolver's stepstat[6]) of the first (mode, ic)
  double kernel_ms() const;                                // summed over the handles

 private:
  void build(const Inputs& in);
  void release() noexcept;          // frees whatever has been built (the constructor calls it before rethrowing, the destructor at the end)
  cpt_handle* h_[2][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}};
  cpt_stepstat* stats_ = nullptr;   // (sharded: the counters of this rank's modes, in the order k_[rank], k_[rank + world], ...)
  int* k_size_cmb_ = nullptr;
  Shard shard_;
  friend class TransferModule;
};

class TransferModule {
 public:
  TransferModule(const Inputs& in, std::shared_ptr<const PerturbationsModule> perturbations_module);
  ~TransferModule();
  TransferModule(const TransferModule&) = delete;

  // ---- data contract of source/transfer_module.h ----
  int index_tt_t0_ = -1, index_tt_t1_ = -1, index_tt_t2_ = -1, index_tt_e_ = -1, index_tt_lcmb_ = -1, index_tt_b_ = -1;
  int* tt_size_ = nullptr;
  int l_size_max_ = 0;
  int** l_size_tt_ = nullptr;
  int* l_size_ = nullptr;
  int* l_ = nullptr;
  int q_size_ = 0;
  double* q_ = nullptr;
  double** k_ = nullptr;
  int index_q_flat_approximation_ = 0;
  double** transfer_ = nullptr;  // transfer_[md][((ic*tt_size+tt)*l_size+l)*q_size+q]
  mutable char error_message_[2048];

  double kernel_ms() const;

 private:
  void build(const Inputs& in);
  void release() noexcept;
  std::shared_ptr<const PerturbationsModule> perturbations_module_;
};

}  // namespace cpt

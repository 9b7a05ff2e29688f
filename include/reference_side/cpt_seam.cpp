// cpt_seam.cpp -- the translation unit a CLASS++ maintainer ADDS to the reference tree (source/) to make the MI355X backend the compute
// engine of PerturbationsModule and TransferModule while every downstream module (PrimordialModule, NonlinearModule, SpectraModule,
// LensingModule, OutputModule, classy) keeps using the reference's own classes and pointer types (seam S1-S3).
//
// The patch to the reference's own files is ten inserted lines and two edited loop headers (oracle/apply_seam.py applies it to scratch copies
// under oracle/_ref/seam/ and `make -C oracle seam` compiles and links the patched tree - the check that this file and the patch type-check
// against the reference's headers; tests/test_gpu_seam.py then RUNS the patched reference on the GPU box):
//
//   source/perturbations_module.h   after `int perturb_init();`                                + `int cpt_fill_sources();`
//                                   after `double k_max_;` (public data)                        + `std::shared_ptr<const cpt::PerturbationsModule> cpt_gpu_;`
//                                   before `class PerturbationsModule`                          + `namespace cpt { class PerturbationsModule; }`
//   source/perturbations_module.cpp perturb_init(), after `Tools::TaskSystem task_system(...)`  + `const bool cpt_done_ = cpt_fill_sources() == 1;`
//                                   the loop `for (index_md = 0; index_md < md_size_; ...` there  -> `for (index_md = 0; !cpt_done_ && index_md < md_size_; ...`
//   source/background_module.h, source/thermodynamics_module.h, tools/non_cold_dark_matter.h: in each class, after `public:`
//                                   + `friend cpt::Inputs MakeCptInputs(const InputModule&, const BackgroundModule&, const ThermodynamicsModule&);`
//                                   (+ the forward declarations in front of the class: the adapter reads the second-derivative tables and the
//                                   momentum grids, which have no accessor)
//   source/transfer_module.h        after `int transfer_init();`                                + `int cpt_fill_transfer();`
//   source/transfer_module.cpp      transfer_init(), after `Tools::TaskSystem task_system(...)` + `const bool cpt_done_ = cpt_fill_transfer() == 1;`
//                                   the loop `for (index_q = 0; index_q < q_size_; ...` there     -> `for (index_q = 0; !cpt_done_ && index_q < q_size_; ...`
//
// i.e. the reference still builds its index maps, its k / tau / q / l grids and its tables exactly as before (perturb_indices_of_perturbs,
// perturb_timesampling_for_sources, perturb_get_k_list, transfer_indices_of_transfers ...: host code, milliseconds); only the two parallel
// loops - the k loop of perturb_init (pm.cpp:668-718) and the q loop of transfer_init (tm.cpp:287-318) - are replaced by the GPU, which fills
// the reference's own sources_ / transfer_ arrays.  The backend is selected at run time (CPT_BACKEND=mi355x in the environment here; an input
// key in a real integration); without it both functions return 0 and the reference runs unchanged.  No `friend` declaration is needed: the two
// functions are members.
//
// Link: -lcpt_host -lcpt (libcpt.so pulls libamdhip64).
#include <cstdlib>
#include <cstring>

#include "background_module.h"
#include "thermodynamics_module.h"
#include "perturbations_module.h"
#include "transfer_module.h"

#include "cpt_modules.hpp"
#include "reference_side/cpt_adapter.h"

namespace {
bool cpt_backend_requested() {
  const char* e = getenv("CPT_BACKEND");
  return e && strcmp(e, "mi355x") == 0;
}
}  // namespace

// returns 1: sources_ filled by the GPU; 0: backend not requested; _FAILURE_-like negative never: errors throw like the module constructors do
int PerturbationsModule::cpt_fill_sources() {
  if (!cpt_backend_requested()) return 0;
  const cpt::Inputs in = MakeCptInputs(*input_module_, *background_module_, *thermodynamics_module_);
  auto gpu = std::make_shared<const cpt::PerturbationsModule>(in);   // every k-mode of every (mode, initial condition) on the GPU
  // the two sides build the same grids (bit for bit, tests/test_host_grids.py); a mismatch would mean the adapter handed over something else
  if (gpu->md_size_ != md_size_ || gpu->tau_size_ != tau_size_ || memcmp(gpu->tau_sampling_, tau_sampling_, sizeof(double) * tau_size_) != 0)
    throw std::runtime_error("cpt seam: the backend's time sampling differs from perturb_timesampling_for_sources");
  for (int md = 0; md < md_size_; md++) {
    if (gpu->k_size_[md] != k_size_[md] || gpu->ic_size_[md] != ic_size_[md] || gpu->tp_size_[md] != tp_size_[md] ||
        memcmp(gpu->k_[md], k_[md], sizeof(double) * k_size_[md]) != 0)
      throw std::runtime_error("cpt seam: the backend's k list or index layout differs from perturb_get_k_list / perturb_indices_of_perturbs");
    for (int i = 0; i < ic_size_[md] * tp_size_[md]; i++)
      memcpy(sources_[md][i], gpu->sources_[md][i], sizeof(double) * (size_t)tau_size_ * k_size_[md]);
  }
  cpt_gpu_ = gpu;   // (kept: the transfer stage reads the sources that are still resident in HBM)
  return 1;
}

int TransferModule::cpt_fill_transfer() {
  if (!cpt_backend_requested() || !perturbations_module_->cpt_gpu_) return 0;
  const cpt::Inputs in = MakeCptInputs(*input_module_, *background_module_, *thermodynamics_module_);
  const cpt::TransferModule gpu(in, perturbations_module_->cpt_gpu_);
  if (gpu.q_size_ != q_size_ || memcmp(gpu.q_, q_, sizeof(double) * q_size_) != 0 || gpu.l_size_max_ != l_size_max_ ||
      memcmp(gpu.l_, l_, sizeof(int) * l_size_max_) != 0)
    throw std::runtime_error("cpt seam: the backend's q or l list differs from transfer_get_q_list / transfer_get_l_list");
  for (int md = 0; md < perturbations_module_->md_size_; md++) {
    if (gpu.tt_size_[md] != tt_size_[md] || gpu.l_size_[md] != l_size_[md])
      throw std::runtime_error("cpt seam: the backend's transfer-type layout differs from transfer_indices_of_transfers");
    memcpy(transfer_[md], gpu.transfer_[md], sizeof(double) * (size_t)perturbations_module_->ic_size_[md] * tt_size_[md] * l_size_[md] * q_size_);
  }
  return 1;
}

// Hot path A (placeholder while the transfer stage is brought up): replaced by the ndf15 wavefront kernel.
#include "cpt_internal.h"
int cpt_perturb_impl(cpt_handle* h, const double*, int, const double*, int, double*, cpt_stepstat*, int*) {
  return cpt_fail(h, CPT_ERR_UNSUPPORTED, "perturbation stage not built yet");
}
int cpt_dbg_lookup_impl(cpt_handle* h, const double*, int, double*) { return cpt_fail(h, CPT_ERR_UNSUPPORTED, "not built yet"); }
int cpt_dbg_derivs_impl(cpt_handle* h, double, double, int, int, int, const double*, double*, int*) {
  return cpt_fail(h, CPT_ERR_UNSUPPORTED, "not built yet");
}

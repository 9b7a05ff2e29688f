// One family of the register-set kernels of hot path A (cpt_perturb_sets.inc), in a translation unit of its own: up to two momentum-bin sets (one non-cold species at the default sampling)
#define CPT_SETS_VARIANT 2
#include "cpt_perturb_sets.inc"

// One family of the register-set kernels of hot path A (cpt_perturb_sets.inc), in a translation unit of its own: up to five momentum-bin sets (three non-cold species)
#define CPT_SETS_VARIANT 5
#include "cpt_perturb_sets.inc"

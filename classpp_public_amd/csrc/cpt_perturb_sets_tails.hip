// One family of the register-set kernels of hot path A (cpt_perturb_sets.inc), in a translation unit of its own: the three l >= 3 tails of hierarchies longer than one wavefront as register sets
#define CPT_SETS_VARIANT 0
#include "cpt_perturb_sets.inc"

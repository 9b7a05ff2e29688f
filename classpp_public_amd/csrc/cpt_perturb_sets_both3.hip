// One family of the register-set kernels of hot path A (cpt_perturb_sets.inc), in a translation unit of its own: the three l >= 3 tails AND up to
// three momentum-bin sets (hierarchies longer than one wavefront together with massive neutrinos: permille-class precision settings)
#define CPT_SETS_VARIANT 13
#include "cpt_perturb_sets.inc"

// Fused coupling layer, MFMA VGPR-accumulator form (all geometries except C=48).  See coupling_fused_impl.inc.
#define NFDP_FUSED_VFORM 1
#include "coupling_fused_impl.inc"

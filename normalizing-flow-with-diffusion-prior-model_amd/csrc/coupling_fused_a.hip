// Fused coupling layer, AGPR-accumulator form (C=48), weight packer and the extern "C" surface.
// See coupling_fused_impl.inc.
#define NFDP_FUSED_VFORM 0
#include "coupling_fused_impl.inc"

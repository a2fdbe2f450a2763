// mlp_fwd_pair_bg.hip -- the wavefront-pair kernel, background, inference instantiation; the source is mlp_fwd_pair.hip.
#define MNR_PAIR_TU 1
#include "mlp_fwd_pair.hip"

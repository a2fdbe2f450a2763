// mlp_fwd_pair_train.hip -- the wavefront-pair kernel, foreground, tape-writing (forward of the W = 512 training path) instantiation; the source is mlp_fwd_pair.hip.
#define MNR_PAIR_TU 2
#include "mlp_fwd_pair.hip"

// mlp_fwd_pair_train_bg.hip -- the wavefront-pair kernel, background, tape-writing instantiation; the source is mlp_fwd_pair.hip.
#define MNR_PAIR_TU 3
#include "mlp_fwd_pair.hip"

// mlp_fwd_pair_train.hip -- the tape-writing instantiations of the wavefront-pair kernel (forward of the W = 512 training path), compiled
// beside the inference ones: the source is mlp_fwd_pair.hip.
#define MNR_PAIR_TRAIN_TU 1
#include "mlp_fwd_pair.hip"

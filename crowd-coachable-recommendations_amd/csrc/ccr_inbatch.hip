// ccr_inbatch.hip -- in-batch-negative contrastive loss (placeholder until the kernels land this round)
#include "ccr_common.h"
using namespace ccr;
extern "C" int ccr_inbatch_ce_fwd(const uint16_t *, const uint16_t *, const uint16_t *, int, int, float, float *, float *, void *) {
    set_error("ccr_inbatch_ce_fwd: not built yet");
    return CCR_ERR_INVALID;
}
extern "C" int ccr_inbatch_ce_bwd(const uint16_t *, const uint16_t *, const uint16_t *, const float *, int, int, float, float,
                                  float *, float *, float *, void *) {
    set_error("ccr_inbatch_ce_bwd: not built yet");
    return CCR_ERR_INVALID;
}

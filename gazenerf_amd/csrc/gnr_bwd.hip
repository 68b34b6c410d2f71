// gnr_bwd.hip -- backward of the hot path (placeholder until the dgrad chain + wgrad land).
#include "gnr_internal.h"

namespace gnr {
int fail(const char* fmt, ...);

size_t bwd_scratch_bytes(const GnrProblem*, int) { return 256; }

int run_bwd(const GnrProblem*, int, const GnrWeights* const*, const GnrOutputGrads*, const GnrInputGrads*,
            const GnrWeightGrads* const*, void*, size_t, void*, size_t, hipStream_t) {
    return fail("gnr_bwd: not implemented in this build");
}
}  // namespace gnr

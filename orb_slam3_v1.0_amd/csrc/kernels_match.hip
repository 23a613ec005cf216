// kernels_match.hip -- Hamming matchers (placeholder until the kernels land in this round).
#include "match.h"

namespace orbfe {

void match_scratch_free(MatchScratch& m)
{
    if (m.d) (void)hipFree(m.d);
    if (m.hpin) (void)hipHostFree(m.hpin);
    m = MatchScratch();
}

int match_projection_run(MatchScratch&, hipStream_t, const orbfe_frame_view*, int, const orbfe_map_point*,
                         const uint8_t*, const int*, float, int, float, float, int*, int*, std::string& err)
{
    err = "match_projection: not built yet";
    return ORBFE_ERR_UNSUPPORTED;
}

int match_bow_run(MatchScratch&, hipStream_t, int, const int*, const int*, const int*, const int*, int,
                  const uint8_t*, const float*, const uint8_t*, int, const uint8_t*, const float*, float, int, int*,
                  int*, std::string& err)
{
    err = "match_bow: not built yet";
    return ORBFE_ERR_UNSUPPORTED;
}

}  // namespace orbfe

// curved rays inside a signed-distance boundary (BND = 1), the record layouts of large fields: BRICK27 (buffer loads below 4 GiB, global loads
// above) and CELL8 with global loads (>= 4 GiB) -- the layouts MER_LAYOUT_AUTO picks for the paper's SDF scenes
// (src/medium/heterogeneousrefractive.cpp:366-375,476-493).  A translation unit of its own: it compiles beside mer_render_sdf.hip.
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_sdf_curved_records(int rifk, int stepper, int sigma, KernelSet &k) {
    if (rifk == RIFK_BRICK27_BUF) return fill_curved<RIFK_BRICK27_BUF, 1>(stepper, sigma, true, k);
    if (rifk == RIFK_BRICK27) return fill_curved<RIFK_BRICK27, 1>(stepper, sigma, true, k);
    if (rifk == RIFK_CELL8) return fill_curved<RIFK_CELL8, 1>(stepper, sigma, true, k);
    return false;
}
}  // namespace mer

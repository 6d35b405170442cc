// curved rays inside a signed-distance boundary (BND = 1): dense (global loads), CELL8 (buffer loads) and B-spline RIFs
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_sdf_curved(int rifk, int stepper, int sigma, KernelSet &k) {
    if (rifk == MER_RIF_TRILINEAR || rifk == RIFK_DENSE_BUF) return fill_curved<MER_RIF_TRILINEAR, 1>(stepper, sigma, true, k);
    if (rifk == RIFK_CELL8_BUF) return fill_curved<RIFK_CELL8_BUF, 1>(stepper, sigma, true, k);
    if (rifk == MER_RIF_BSPLINE3) return fill_curved<MER_RIF_BSPLINE3, 1>(stepper, sigma, true, k);
    return false;
}
}  // namespace mer

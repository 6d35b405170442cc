// curved rays, trilinear RIF in the BRICK27 / BRICK125 layouts: global loads (any size) / buffer loads (< 4 GiB)
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_brick(int rifk, int stepper, int sigma, bool extra, KernelSet &k) {
    return rifk == RIFK_BRICK27_BUF ? fill_curved<RIFK_BRICK27_BUF, 0>(stepper, sigma, extra, k) : fill_curved<RIFK_BRICK27, 0>(stepper, sigma, extra, k);
}
}  // namespace mer

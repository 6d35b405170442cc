// curved rays, trilinear RIF in the CELL8 layout: global loads (any size) / buffer loads (< 4 GiB)
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_cell8(int rifk, int stepper, int sigma, bool extra, KernelSet &k) {
    return rifk == RIFK_CELL8_BUF ? fill_curved<RIFK_CELL8_BUF, 0>(stepper, sigma, extra, k) : fill_curved<RIFK_CELL8, 0>(stepper, sigma, extra, k);
}
}  // namespace mer

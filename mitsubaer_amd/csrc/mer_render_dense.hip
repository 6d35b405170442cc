// curved rays, trilinear RIF in the dense (VOL payload) layout: global loads (any size) / buffer loads (< 4 GiB)
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_dense(int rifk, int stepper, int sigma, bool extra, KernelSet &k) {
    return rifk == RIFK_DENSE_BUF ? fill_curved<RIFK_DENSE_BUF, 0>(stepper, sigma, extra, k) : fill_curved<MER_RIF_TRILINEAR, 0>(stepper, sigma, extra, k);
}
}  // namespace mer

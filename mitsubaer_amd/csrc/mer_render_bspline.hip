// curved rays, cubic B-spline RIF (splinevolume)
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_bspline(int stepper, int sigma, bool extra, KernelSet &k) { return fill_curved<MER_RIF_BSPLINE3, 0>(stepper, sigma, extra, k); }
}  // namespace mer

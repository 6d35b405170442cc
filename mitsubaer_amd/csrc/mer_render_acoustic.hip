// curved rays through the analytic acoustic RIF (acousticrifvolume)
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_acoustic(int stepper, int sigma, bool extra, KernelSet &k) { return fill_curved<RIFK_ACOUSTIC, 0>(stepper, sigma, extra, k); }
}  // namespace mer

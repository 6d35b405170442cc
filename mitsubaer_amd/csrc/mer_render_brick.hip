// curved rays, trilinear RIF in the BRICK27 / BRICK125 layouts: global loads (any size) / buffer loads (< 4 GiB)
#include "mer_render_groups.hpp"
namespace mer {
bool kernels_brick(int rifk, int stepper, int sigma, bool extra, KernelSet &k) {
    return rifk == RIFK_BRICK27_BUF ? fill_curved<RIFK_BRICK27_BUF, 0>(stepper, sigma, extra, k) : fill_curved<RIFK_BRICK27, 0>(stepper, sigma, extra, k);
}
}  // namespace mer

#ifdef MER_PROFILE
// section profile of this translation unit's K_event instances (mer_wavefront.hpp, PROF): out[0..9] = wave cycles per section, out[15] = waves with work
extern "C" int mer_debug_prof(unsigned long long *out, int reset) {
    static unsigned long long all[256 * 16];
    if (out) {
        if (hipMemcpyFromSymbol(all, HIP_SYMBOL(mer::mer_prof), sizeof all) != hipSuccess) return 1;
        for (int k = 0; k < 16; k++) { out[k] = 0; for (int r = 0; r < 256; r++) out[k] += all[r * 16 + k]; }
    }
    if (reset) { for (auto &v : all) v = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(mer::mer_prof), all, sizeof all) != hipSuccess) return 1; }
    return 0;
}
#endif

// straight rays (heterogeneous / homogeneous medium, cube / sphere / signed-distance boundary) + the eight K_gen instances
#include "mer_render_groups.hpp"
namespace mer {
GenKernel gen_kernel_for(bool curved, bool extra, int bnd) {
    if (bnd) return curved ? gen_kernel<true, true, 1> : gen_kernel<false, true, 1>;
    if (curved) return extra ? gen_kernel<true, true, 0> : gen_kernel<true, false, 0>;
    return extra ? gen_kernel<false, true, 0> : gen_kernel<false, false, 0>;
}
bool kernels_straight(int sigma, int bnd, bool extra, KernelSet &k) {
    if (bnd == 0 && sigma == MER_SIGMA_GRID) fill_kernels<false, MER_RIF_TRILINEAR, MER_STEP_VERLET, MER_SIGMA_GRID, 0>(extra, k);
    else if (bnd == 0) fill_kernels<false, MER_RIF_TRILINEAR, MER_STEP_VERLET, MER_SIGMA_HOMOGENEOUS, 0>(extra, k);
    else if (sigma == MER_SIGMA_GRID) fill_kernels<false, MER_RIF_TRILINEAR, MER_STEP_VERLET, MER_SIGMA_GRID, 1>(extra, k);
    else fill_kernels<false, MER_RIF_TRILINEAR, MER_STEP_VERLET, MER_SIGMA_HOMOGENEOUS, 1>(extra, k);
    return true;
}
}  // namespace mer

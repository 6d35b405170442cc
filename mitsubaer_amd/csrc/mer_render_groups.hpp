// mer_render_groups.hpp -- shared by the mer_render_<group>.hip files: each of them instantiates the wavefront kernels
// (mer_wavefront.hpp) of one group of (CURVED, RIF, STEPPER, SIGMA, BND) combinations and hands their addresses to the host loop
// (mer_render.hip).  One group per translation unit: the groups compile in parallel.
#pragma once
#include "mer_internal.hpp"
#include "mer_wavefront.hpp"

namespace mer {

// K_gen depends on (CURVED, EXTRA, BND) only: its eight instances live in mer_render_straight.hip
GenKernel gen_kernel_for(bool curved, bool extra, int bnd);

template <bool CURVED, int RIF, int STEPPER, int SIGMA, int BND>
static inline void fill_kernels(bool extra, KernelSet &k) {
    constexpr bool X = BND != 0;          // the signed-distance boundary exists in the EXTRA kernels only
    k.gen = gen_kernel_for(CURVED, extra || X, BND);
    if (extra || X) k.event = event_kernel<CURVED, RIF, STEPPER, SIGMA, true, BND>;
    else k.event = event_kernel<CURVED, RIF, STEPPER, SIGMA, X, BND>;
    k.event_inline = nullptr;
    if constexpr (!CURVED && SIGMA == MER_SIGMA_GRID) {
        if (extra || X) k.event_inline = event_kernel<CURVED, RIF, STEPPER, SIGMA, true, BND, true>;
        else k.event_inline = event_kernel<CURVED, RIF, STEPPER, SIGMA, X, BND, true>;
    }
    k.march = march_kernel<CURVED, RIF, STEPPER, SIGMA, BND>;
    if constexpr (RIF == RIFK_BRICK27_BUF) k.march_lds = march_kernel<CURVED, RIFK_BRICK27_LDS, STEPPER, SIGMA, BND>;
    else k.march_lds = nullptr;
    if constexpr (CURVED) { k.connect = connect_stage_kernel<RIF, STEPPER, SIGMA, BND, BND != 0>; k.connect_cross = connect_stage_kernel<RIF, STEPPER, SIGMA, BND, true>; }
    else k.connect = k.connect_cross = nullptr;
}
// the four (STEPPER, SIGMA) combinations of one curved fetch kind
template <int RIF, int BND>
static inline bool fill_curved(int stepper, int sigma, bool extra, KernelSet &k) {
    if (stepper == MER_STEP_VERLET && sigma == MER_SIGMA_GRID) fill_kernels<true, RIF, MER_STEP_VERLET, MER_SIGMA_GRID, BND>(extra, k);
    else if (stepper == MER_STEP_RK4 && sigma == MER_SIGMA_GRID) fill_kernels<true, RIF, MER_STEP_RK4, MER_SIGMA_GRID, BND>(extra, k);
    else if (stepper == MER_STEP_VERLET && sigma == MER_SIGMA_HOMOGENEOUS) fill_kernels<true, RIF, MER_STEP_VERLET, MER_SIGMA_HOMOGENEOUS, BND>(extra, k);
    else if (stepper == MER_STEP_RK4 && sigma == MER_SIGMA_HOMOGENEOUS) fill_kernels<true, RIF, MER_STEP_RK4, MER_SIGMA_HOMOGENEOUS, BND>(extra, k);
    else return false;
    return true;
}

}  // namespace mer

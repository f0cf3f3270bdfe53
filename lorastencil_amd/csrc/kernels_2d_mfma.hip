// kernels_2d_mfma.hip -- low-rank (U X) V formulation of the 2D sweeps on v_mfma_f64_16x16x4_f64.
// (placeholder until the MFMA kernel lands; lora_plan_set_variant(MFMA) is refused meanwhile)
#include <hip/hip_runtime.h>

#include "engine.h"

namespace lora {

hipError_t launch_2d_mfma(const Plan &, const double *, double *, int, int, hipStream_t) {
    return hipErrorNotSupported;
}

const char *kernel_name_2d_mfma(const Plan &) { return "stencil2d_mfma_kernel"; }

bool mfma_available() { return false; }

}  // namespace lora

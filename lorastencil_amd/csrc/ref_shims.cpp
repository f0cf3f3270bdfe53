// ref_shims.cpp -- C++-linkage operators with the reference's exact signatures, forwarding to the C ABI.
// Error policy of the reference's CUDA_CHECK (2d_utils.h:22-36): report and exit(1).
#include "lorastencil_ref_shims.h"

#include <cstdio>
#include <cstdlib>

#include "lorastencil.h"

namespace {
void check(int status, const char *what) {
    if (status == LORA_OK) return;
    std::printf("LoRAStencil HIP Error:\n    Call:       %s\n    Error code: %d\n    Error text: %s %s\n", what,
                status, lora_strerror(status), lora_last_error());
    std::exit(1);
}
}  // namespace

void gpu_1d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
              const int time, const int input_n) {
    check(lora_gpu_1d1r(in, out, params, time, input_n), "gpu_1d1r");
}

void gpu_1d2r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
              const int time, const int input_n) {
    check(lora_gpu_1d2r(in, out, params, time, input_n), "gpu_1d2r");
}

void gpu_star_2d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                   const int times, const int input_m, const int input_n) {
    check(lora_gpu_star_2d1r(in, out, params, times, input_m, input_n), "gpu_star_2d1r");
}

void gpu_star_2d3r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                   const int times, const int input_m, const int input_n) {
    check(lora_gpu_star_2d3r(in, out, params, times, input_m, input_n), "gpu_star_2d3r");
}

void gpu_box_2d3r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                  const int times, const int input_m, const int input_n) {
    check(lora_gpu_box_2d3r(in, out, params, times, input_m, input_n), "gpu_box_2d3r");
}

void gpu_box_3d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                  const int times, const int input_h, const int input_m, const int input_n) {
    check(lora_gpu_box_3d1r(in, out, params, times, input_h, input_m, input_n), "gpu_box_3d1r");
}

void gpu_star_3d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                   const int times, const int input_h, const int input_m, const int input_n) {
    check(lora_gpu_star_3d1r(in, out, params, times, input_h, input_m, input_n), "gpu_star_3d1r");
}

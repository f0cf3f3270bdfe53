// Where does a step of the register-resident 3D kernel spend its time?  kernels_3d_lanes.hip compiled with one part of
// the step removed at a time (-DLORA_L3_ABLATE=bits: 1 barriers, 2 stores, 4 plane loads, 8 cross-lane moves, 16 EDGE
// steps), timed on star3d1r 512^3 and the separable box 768^3.  The results of an ablated build are wrong on purpose.
//   for b in 0 1 2 4 8 16 6 7 15 31; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLORA_L3_ABLATE=$b -I../../include \
//       -I../../lorastencil_amd/csrc -o bin/lanes3_ablate_$b lanes3_ablate.hip; done
#include "kernels_3d_lanes.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

static void run(bool sep, int d, int zc) {
    lora::Plan p;
    p.ndim = 3;
    p.dims[0] = p.dims[1] = p.dims[2] = d;
    p.dtype = LORA_F64;
    p.boundary = LORA_BC_REFERENCE;
    p.tapset = sep ? lora::TAPS3D_BOX : lora::TAPS3D_STAR;
    p.sep64_valid = sep;
    p.fused_z_chunk = zc;
    for (int k = 0; k < 27; ++k) p.w[k] = 0.0;
    p.w[4] = p.w[10] = p.w[12] = p.w[14] = p.w[16] = p.w[22] = 0.125;
    p.w[13] = 0.25;
    const double c[3] = {0.25, 0.5, 0.25};
    for (int k = 0; k < 9; ++k) p.sep64[k] = c[k % 3];
    const size_t count = (size_t) (d + 2) * (d + 4) * (d + 8);
    double *b0, *b1;
    if (hipMalloc(&b0, count * 8) != hipSuccess || hipMalloc(&b1, count * 8) != hipSuccess) exit(1);
    std::vector<double> h(count);
    for (size_t i = 0; i < count; ++i) h[i] = (double) ((i * 2654435761u) % 1000) / 1000.0;
    hipMemcpy(b0, h.data(), count * 8, hipMemcpyHostToDevice);
    hipMemset(b1, 0, count * 8);
    for (int i = 0; i < 3; ++i) {
        lora::launch_3d_lanes(p, 4, b0, b1, 0, d, nullptr);
        lora::launch_3d_lanes(p, 4, b1, b0, 0, d, nullptr);
    }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = 10;
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) {
        lora::launch_3d_lanes(p, 4, b0, b1, 0, d, nullptr);
        lora::launch_3d_lanes(p, 4, b1, b0, 0, d, nullptr);
    }
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (2 * it);
    printf("ablate=%d %s %d^3 zc=%d: %.1f us per launch, %.1f GStencils/s\n", LORA_L3_ABLATE, sep ? "sep-box" : "star", d, zc, us,
           (double) d * d * d * 4 / us / 1e3);
    hipFree(b0);
    hipFree(b1);
}

int main(int argc, char **argv) {
    if (argc > 1) {  // star 512^3 at the given z-chunk lengths
        for (int i = 1; i < argc; ++i) run(false, 512, atoi(argv[i]));
        return 0;
    }
    run(false, 512, 0);
    run(true, 768, 0);
    return 0;
}

// cli_selfcheck.cpp -- the `--check` mode of the CLIs: the reference's CHECK_ERROR block
// (1d/main.cu:135-173, 2d/main.cu:282-328, 3d/main.cu:201-246).  One sweep of the GPU operator is compared
// with a naive CPU loop over the same padded input, absolute tolerance 1e-7, every offending point printed.
// This CPU loop is a checker for the harness only; nothing in liblorastencil_hip calls it.
#include <cmath>
#include <cstdio>
#include <vector>

#include "lorastencil.h"
#include "lorastencil_ref_shims.h"

namespace {

void naive_1d(const double *in, double *out, const double *w, int cols) {
    for (int c = 4; c < cols - 4; ++c) {
        double s = 0.0;
        for (int t = 0; t < 9; ++t) s += w[t] * in[c - 4 + t];
        out[c] = s;
    }
}

void naive_2d(const double *in, double *out, const double *w, int rows, int cols) {
    for (int r = 4; r < rows - 4; ++r)
        for (int c = 4; c < cols - 4; ++c) {
            double s = 0.0;
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx)
                    s += w[(dy + 3) * 7 + dx + 3] * in[(size_t) (r + dy) * cols + c + dx];
            out[(size_t) r * cols + c] = s;
        }
}

void naive_3d(const double *in, double *out, const double *w, int heights, int rows, int cols) {
    const size_t plane = (size_t) rows * cols;
    for (int h = 1; h < heights - 1; ++h)
        for (int r = 2; r < rows - 2; ++r)
            for (int c = 4; c < cols - 4; ++c) {
                double s = 0.0;
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx)
                            s += w[(dz + 1) * 9 + (dy + 1) * 3 + dx + 1] *
                                 in[(h + dz) * plane + (size_t) (r + dy) * cols + c + dx];
                out[h * plane + (size_t) r * cols + c] = s;
            }
}

}  // namespace

long selfcheck_compare(int shape, const std::vector<double> &input, const double *params, const int *dims) {
    const size_t count = input.size();
    std::vector<double> naive(count, 0.0), lora(count, 0.0);
    const int nd = lora_shape_ndim(shape);
    long bad = 0;
    if (nd == 1) {
        const int cols = dims[0] + 8;
        naive_1d(input.data(), naive.data(), params, cols);
        if (shape == LORA_1D1R)
            gpu_1d1r(input.data(), lora.data(), params, 1, dims[0]);
        else
            gpu_1d2r(input.data(), lora.data(), params, 1, dims[0]);
        std::printf("Comparing naive and lora\n");
        for (int c = 0; c < cols - 4; ++c)
            if (std::fabs(naive[c] - lora[c]) > 1e-7) {
                std::printf("col = %d, naive = %lf, lora = %lf\n", c, naive[c], lora[c]);
                ++bad;
            }
    } else if (nd == 2) {
        const int rows = dims[0] + 8, cols = dims[1] + 8;
        naive_2d(input.data(), naive.data(), params, rows, cols);
        if (shape == LORA_STAR2D1R)
            gpu_star_2d1r(input.data(), lora.data(), params, 1, dims[0], dims[1]);
        else if (shape == LORA_STAR2D3R)
            gpu_star_2d3r(input.data(), lora.data(), params, 1, dims[0], dims[1]);
        else
            gpu_box_2d3r(input.data(), lora.data(), params, 1, dims[0], dims[1]);
        std::printf("Comparing naive and lora\n");
        for (int r = 4; r < rows - 4; ++r)
            for (int c = 4; c < cols - 4; ++c) {
                const size_t k = (size_t) r * cols + c;
                if (std::fabs(naive[k] - lora[k]) > 1e-7) {
                    std::printf("row = %d, col = %d, naive = %lf, lora = %lf\n", r, c, naive[k], lora[k]);
                    ++bad;
                }
            }
    } else {
        const int heights = dims[0] + 2, rows = dims[1] + 4, cols = dims[2] + 8;
        naive_3d(input.data(), naive.data(), params, heights, rows, cols);
        if (shape == LORA_BOX3D1R)
            gpu_box_3d1r(input.data(), lora.data(), params, 1, dims[0], dims[1], dims[2]);
        else
            gpu_star_3d1r(input.data(), lora.data(), params, 1, dims[0], dims[1], dims[2]);
        std::printf("Comparing naive and lora\n");
        for (int h = 1; h < heights - 1; ++h)
            for (int r = 2; r < rows - 2; ++r)
                for (int c = 4; c < cols - 4; ++c) {
                    const size_t k = ((size_t) h * rows + r) * cols + c;
                    if (std::fabs(naive[k] - lora[k]) > 1e-7) {
                        std::printf("height = %d, row = %d, col = %d, naive = %lf, output = %lf\n", h, r, c, naive[k],
                                    lora[k]);
                        ++bad;
                    }
                }
    }
    return bad;
}

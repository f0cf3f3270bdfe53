#include <cstdio>
#include <cmath>
#include <vector>
#include "lorastencil.h"
int main() {
    double p[49], w[49], u[28], v[28], res, U[49], V[49], S[7];
    for (int s = 0; s < LORA_NUM_SHAPES; ++s) {
        int n = lora_default_params(s, p);
        if (n != lora_shape_ntaps(s)) return 1;
        if (lora_effective_weights(s, p, w) != n) return 2;
        if (lora_effective_weights(s, nullptr, w) != n) return 3;
    }
    lora_default_params(LORA_BOX2D3R, p);
    if (lora_factorize_7x7(p, u, v, &res) != 0 || res != 0.0) return 4;
    for (int k = 0; k < 49; ++k) p[k] = std::sin(k * 1.7) + (k % 5);
    lora_factorize_7x7(p, u, v, &res);
    if (lora_svd_7x7(p, U, V, S) != 0) return 5;
    for (int k = 0; k < 49; ++k) p[k] = 0.0;
    lora_factorize_7x7(p, u, v, &res);   // zero pivots: NaNs allowed, no memory errors
    lora_svd_7x7(p, U, V, S);
    lora_rng g; lora_rng_seed(&g, 1);
    std::vector<double> a(100000);
    lora_fill_rand(a.data(), a.size(), 100, &g);
    std::vector<uint16_t> b(a.size());
    lora_f64_to_bf16(a.data(), b.data(), a.size());
    lora_bf16_to_f64(b.data(), a.data(), a.size());
    int dims[3] = {7, 9, 16};
    if (lora_padded_count(LORA_BOX3D1R, dims) != 9u * 13u * 24u) return 6;
    if (lora_shape_from_name("star2d1r") != LORA_STAR2D1R || lora_shape_from_name(nullptr) >= 0) return 7;
    float cba[9];
    lora_default_params(LORA_BOX3D1R, p);
    lora_effective_weights(LORA_BOX3D1R, p, w);
    if (lora_separable_3x3x3(w, cba) != 1 || cba[1] != 2.0f) return 8;      // reference box3d1r: c = (1, 2, 1)
    lora_effective_weights(LORA_STAR3D1R, nullptr, w);
    if (lora_separable_3x3x3(w, cba) != 0) return 9;
    for (int k = 0; k < 27; ++k) w[k] = 0.0;
    if (lora_separable_3x3x3(w, cba) != 0 || lora_separable_3x3x3(nullptr, cba) >= 0) return 10;
    std::puts("SAN_OK");
    return 0;
}

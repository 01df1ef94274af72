// Launch wrappers of the HiFT vocoder kernels (hift_kernels.hip); the GEMMs are launch_gemm's (kernels.hpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace t3k {

struct HiftWindowArgs {
    const float* x = nullptr;   // [L_in][ldx], C channels used
    int L_in = 0, C = 0, ldx = 0;
    float* col = nullptr;       // [L_out][Kp]
    int L_out = 0, k = 0, Kp = 0;
    int stride = 1, dil = 1, pad = 0;   // conv1d geometry; pad = left padding
    int up = 0;                 // > 0: conv_transpose1d with this stride (then `pad` = its padding argument)
    int act = 0;                // activation applied to x on the way: 0 none, 1 leaky_relu(slope), 2 Snake(alpha[c])
    float slope = 0.f;
    const float* alpha = nullptr;
};
hipError_t launch_hift_window(const HiftWindowArgs& a, hipStream_t s);
// mode 0 elu(a), 1 |a|, 2 (a + b + c) / 3, 3 a + b
hipError_t launch_hift_elem(const float* a, const float* b, const float* c, float* out, size_t n, int mode, hipStream_t s);
// f0 [T] -> source [T * up] (f0_upsamp + source_module_hn_nsf, R/hifigan.R:216-307, 609-614); base: scratch [n_harm][T] doubles;
// phase [n_harm]; noise [n_harm][T * up] or nullptr (no noise term); lw [n_harm], lb [1]: l_linear
hipError_t launch_hift_source(const float* f0, double* base, const float* phase, const float* noise, const float* lw, const float* lb, int T,
                              int n_harm, int up, float sr, float voiced_threshold, float* out, hipStream_t s);
// STFT of the source: out [L / 4 + 1][ld] = 9 real | 9 imaginary | zeros
hipError_t launch_hift_stft(const float* src, int L, float* out, int ld, hipStream_t s);
// conv_post rows [frames][18] -> audio [4 * (frames - 1)], clamped to +-limit; spec: scratch [frames][18]
hipError_t launch_hift_istft(const float* post, int frames, float* spec, float* audio, float limit, hipStream_t s);

}  // namespace t3k

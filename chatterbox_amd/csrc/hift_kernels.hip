// gfx950 kernels of the HiFT vocoder (R/hifigan.R:13-627; SURVEY §8 f4, third piece).  Sequences are rows [L][C] fp32
// (time-major, channels contiguous); every convolution and transposed convolution is a GEMM over window rows
// (hift_window_kernel builds them, with the activation in front of the convolution -- Snake, leaky ReLU -- applied on the
// way), on launch_gemm's exact-fp32 MFMA kernels.  The rest is small per-sample work: the harmonic-plus-noise source, the
// 16-point STFT of the source, the inverse STFT of the decoder's magnitude / phase rows.
#include "hift_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>

namespace t3k {

namespace {

constexpr int kFft = 16, kHop = 4, kBins = kFft / 2 + 1;   // istft_n_fft 16, istft_hop_len 4 (R/hifigan.R:806-807)

__device__ __forceinline__ float hann16(int n) { return 0.5f * (1.0f - cosf(6.283185307179586f * (float)n / 16.0f)); }

// Window rows of a conv1d / conv_transpose1d as GEMM rows: col[t][j * C + c] = act(x[src(t, j)][c]) or 0 outside the input,
// columns [k * C, Kp) zero (K padded to the GEMM's 64-element step; the weights are padded alike at load).
//   conv (up == 0):       src = t * stride + j * dil - pad                     (R/hifigan.R:113-123, 158-166, 382, 428-432, 452)
//   transposed (up > 0):  q = t + pad - j, src = q / up where q >= 0 and q % up == 0  (R/hifigan.R:402-405)
// act: 0 none, 1 leaky_relu(slope), 2 Snake x + sin(x a)^2 / (a + 1e-9) per channel (R/hifigan.R:75-86); act(0) = 0 for all.
__global__ __launch_bounds__(256) void hift_window_kernel(HiftWindowArgs a) {
    const int kp4 = a.Kp >> 2;
    const size_t total = (size_t)a.L_out * kp4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int t = (int)(e / kp4), col = (int)(e % kp4) << 2;
        const int j = col / a.C, c = col - j * a.C;
        float4 v{0.f, 0.f, 0.f, 0.f};
        if (j < a.k) {
            int src;
            bool ok;
            if (a.up > 0) {
                const int q = t + a.pad - j;
                ok = q >= 0 && q % a.up == 0;
                src = q / a.up;
            } else {
                src = t * a.stride + j * a.dil - a.pad;
                ok = src >= 0;
            }
            if (ok && src < a.L_in) {
                v = *reinterpret_cast<const float4*>(a.x + (size_t)src * a.ldx + c);
                if (a.act == 1) {
                    v.x = v.x >= 0.f ? v.x : v.x * a.slope; v.y = v.y >= 0.f ? v.y : v.y * a.slope;
                    v.z = v.z >= 0.f ? v.z : v.z * a.slope; v.w = v.w >= 0.f ? v.w : v.w * a.slope;
                } else if (a.act == 2) {
                    const float4 al = *reinterpret_cast<const float4*>(a.alpha + c);
                    float s0 = sinf(v.x * al.x), s1 = sinf(v.y * al.y), s2 = sinf(v.z * al.z), s3 = sinf(v.w * al.w);
                    v.x = v.x + (1.0f / (al.x + 1e-9f)) * (s0 * s0); v.y = v.y + (1.0f / (al.y + 1e-9f)) * (s1 * s1);
                    v.z = v.z + (1.0f / (al.z + 1e-9f)) * (s2 * s2); v.w = v.w + (1.0f / (al.w + 1e-9f)) * (s3 * s3);
                }
            }
        }
        *reinterpret_cast<float4*>(a.col + (size_t)t * a.Kp + col) = v;
    }
}

// mode 0: elu (F0 predictor, R/hifigan.R:159-167), 1: abs (:179), 2: out = (a + b + c) / 3 in the reference's order (:552-561),
// 3: out = a + b (:549)
__global__ __launch_bounds__(256) void hift_elem_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                        float* __restrict__ out, size_t n, int mode) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float v = a[e];
        float y;
        if (mode == 0) y = v > 0.f ? v : expm1f(v);
        else if (mode == 1) y = fabsf(v);
        else if (mode == 2) y = __fdiv_rn(__fadd_rn(__fadd_rn(v, b[e]), c[e]), 3.0f);
        else y = __fadd_rn(v, b[e]);
        out[e] = y;
    }
}

// SineGen's phase accumulation (R/hifigan.R:229-234): per harmonic, the running sum of f0 * (h + 1) / sr over the samples
// BEFORE each frame (f0 is piecewise constant over `up` samples: nn_upsample, nearest), accumulated in double as libtorch's
// CPU cumsum does.  One thread per harmonic walks the frames.
__global__ void hift_phase_base_kernel(const float* __restrict__ f0, int T, int n_harm, int up, float sr, double* __restrict__ base) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_harm) return;
    double acc = 0.0;
    for (int fr = 0; fr < T; ++fr) {
        base[(size_t)h * T + fr] = acc;
        const float v = __fdiv_rn(__fmul_rn(f0[fr], (float)(h + 1)), sr);
        acc += (double)v * (double)up;
    }
}

// source_module_hn_nsf (R/hifigan.R:216-256, 291-307): one thread per output sample
__global__ __launch_bounds__(256) void hift_source_kernel(const float* __restrict__ f0, const double* __restrict__ base,
                                                          const float* __restrict__ phase, const float* __restrict__ noise,
                                                          const float* __restrict__ lw, const float* __restrict__ lb, int T, int n_harm, int up,
                                                          float sr, float voiced_threshold, float* __restrict__ out) {
    const size_t n = (size_t)T * up;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int fr = (int)(i / up), k = (int)(i % up);
        const float f = f0[fr];
        const float uv = f > voiced_threshold ? 1.0f : 0.0f;
        const float namp = __fadd_rn(__fmul_rn(uv, 0.003f), __fdiv_rn(__fmul_rn(1.0f - uv, 0.1f), 3.0f));
        float acc = 0.f;
        for (int h = 0; h < n_harm; ++h) {
            const float v = __fdiv_rn(__fmul_rn(f, (float)(h + 1)), sr);
            const float cs = (float)(base[(size_t)h * T + fr] + (double)v * (double)(k + 1));      // the prefix, rounded to float
            const float frac = cs - floorf(cs);                                                     // cumsum %% 1
            const float theta = __fmul_rn(6.2831855f, frac);
            const float ph = h == 0 ? 0.f : phase[h];
            float sw = __fmul_rn(0.1f, sinf(__fadd_rn(theta, ph)));
            const float nz = noise != nullptr ? noise[(size_t)h * n + i] : 0.f;
            sw = __fadd_rn(__fmul_rn(sw, uv), __fmul_rn(namp, nz));
            acc = fmaf(sw, lw[h], acc);
        }
        out[i] = tanhf(acc + lb[0]);
    }
}

// torch_stft(n_fft 16, hop 4, hann, center = TRUE, reflect) of the source, rows [frames][ld]: 9 real bins, 9 imaginary bins,
// zero padding up to ld (R/hifigan.R:464-481, 516-517)
__global__ __launch_bounds__(256) void hift_stft_kernel(const float* __restrict__ s, int L, int frames, float* __restrict__ out, int ld) {
    const size_t total = (size_t)frames * ld;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int fr = (int)(e / ld), ch = (int)(e % ld);
        float v = 0.f;
        if (ch < 2 * kBins) {
            const int f = ch < kBins ? ch : ch - kBins;
            float acc = 0.f;
            for (int n = 0; n < kFft; ++n) {
                int idx = fr * kHop + n - kFft / 2;          // position in the unpadded signal
                if (idx < 0) idx = -idx;                      // reflect (without repeating the edge sample)
                if (idx >= L) idx = 2 * (L - 1) - idx;
                const float x = s[idx] * hann16(n);
                const float ang = -6.283185307179586f * (float)((f * n) & 15) / 16.0f;
                acc += x * (ch < kBins ? cosf(ang) : sinf(ang));
            }
            v = acc;
        }
        out[e] = v;
    }
}

// conv_post rows [frames][18] -> complex spectrum rows: magnitude = min(exp(x[0..8]), 100), phase = sin(x[9..17]),
// re = mag cos(phase), im = mag sin(phase)  (R/hifigan.R:569-571, 485-491)
__global__ __launch_bounds__(256) void hift_spec_kernel(const float* __restrict__ x, int frames, float* __restrict__ spec) {
    const size_t total = (size_t)frames * kBins;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t fr = e / kBins;
        const int f = (int)(e % kBins);
        const float mag = fminf(expf(x[fr * 2 * kBins + f]), 100.0f);
        const float ph = sinf(x[fr * 2 * kBins + kBins + f]);
        spec[fr * 2 * kBins + f] = mag * cosf(ph);
        spec[fr * 2 * kBins + kBins + f] = mag * sinf(ph);
    }
}

// torch_istft(center = TRUE) + clamp (R/hifigan.R:493-500, 577): every output sample gathers the <= 4 frames that cover it --
// the 16-point inverse real DFT of each (imaginary parts of bins 0 and 8 ignored), times the window, over the window envelope
__global__ __launch_bounds__(256) void hift_istft_kernel(const float* __restrict__ spec, int frames, float* __restrict__ audio, int n_out,
                                                         float limit) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_out; i += gridDim.x * 256) {
        const int s = i + kFft / 2;
        int f_lo = (s - (kFft - 1) + kHop - 1) / kHop, f_hi = s / kHop;
        if (s - (kFft - 1) < 0) f_lo = 0;
        if (f_hi > frames - 1) f_hi = frames - 1;
        float y = 0.f, env = 0.f;
        for (int fr = f_lo; fr <= f_hi; ++fr) {
            const int m = s - fr * kHop;
            const float* sp = spec + (size_t)fr * 2 * kBins;
            float acc = sp[0] + ((m & 1) ? -sp[kBins - 1] : sp[kBins - 1]);
            for (int f = 1; f < kBins - 1; ++f) {
                const float ang = 6.283185307179586f * (float)((f * m) & 15) / 16.0f;
                acc += 2.0f * (sp[f] * cosf(ang) - sp[kBins + f] * sinf(ang));
            }
            const float w = hann16(m);
            y += w * (acc * (1.0f / 16.0f));
            env += w * w;
        }
        const float v = y / env;
        audio[i] = fminf(fmaxf(v, -limit), limit);
    }
}

inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

hipError_t launch_hift_window(const HiftWindowArgs& a, hipStream_t s) {
    if (!a.x || !a.col || a.L_in <= 0 || a.L_out <= 0 || a.C <= 0 || (a.C & 3) || a.ldx < a.C || (a.ldx & 3) || a.k <= 0 ||
        a.Kp < a.k * a.C || (a.Kp & 3) || a.stride <= 0 || a.dil <= 0 || a.up < 0 || (a.act == 2 && !a.alpha))
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(hift_window_kernel, dim3(grid_for((size_t)a.L_out * (a.Kp >> 2))), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_hift_elem(const float* a, const float* b, const float* c, float* out, size_t n, int mode, hipStream_t s) {
    if (!a || !out || n == 0 || mode < 0 || mode > 3 || (mode >= 2 && !b) || (mode == 2 && !c)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hift_elem_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, out, n, mode);
    return hipGetLastError();
}

hipError_t launch_hift_source(const float* f0, double* base, const float* phase, const float* noise, const float* lw, const float* lb, int T,
                              int n_harm, int up, float sr, float voiced_threshold, float* out, hipStream_t s) {
    if (!f0 || !base || !phase || !lw || !lb || !out || T <= 0 || n_harm <= 0 || n_harm > 64 || up <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hift_phase_base_kernel, dim3(1), dim3(64), 0, s, f0, T, n_harm, up, sr, base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hift_source_kernel, dim3(grid_for((size_t)T * up)), dim3(256), 0, s, f0, base, phase, noise, lw, lb, T, n_harm, up, sr,
                       voiced_threshold, out);
    return hipGetLastError();
}

hipError_t launch_hift_stft(const float* src, int L, float* out, int ld, hipStream_t s) {
    if (!src || !out || L < kFft || ld < 2 * kBins) return hipErrorInvalidValue;
    const int frames = L / kHop + 1;
    hipLaunchKernelGGL(hift_stft_kernel, dim3(grid_for((size_t)frames * ld)), dim3(256), 0, s, src, L, frames, out, ld);
    return hipGetLastError();
}

hipError_t launch_hift_istft(const float* post, int frames, float* spec, float* audio, float limit, hipStream_t s) {
    if (!post || !spec || !audio || frames < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hift_spec_kernel, dim3(grid_for((size_t)frames * kBins)), dim3(256), 0, s, post, frames, spec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n_out = kHop * (frames - 1);
    hipLaunchKernelGGL(hift_istft_kernel, dim3(grid_for((size_t)n_out)), dim3(256), 0, s, spec, frames, audio, n_out, limit);
    return hipGetLastError();
}

}  // namespace t3k

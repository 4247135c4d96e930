// Generic-architecture RevResNet ops: the reference's constructor accepts any nBlocks / nStrides / nChannels / mult / kernel /
// in_channel / hidden_dim / sp_steps (models/RevResNet.py:166-201); the tuned kernels of conv.hip / conv3.hip implement the one
// published architecture.  These kernels run every OTHER architecture: plain NCHW fp32 tensors, exact fp32 FMA arithmetic,
// written for correctness (LDS-tiled direct convolution, nothing architecture-specific) — the slow, complete path.
//
//   vst_generic_conv          ReflectionPad2d((K-1)/2) + Conv2d(K, stride, bias) [+ ReLU] [+ residual: out = old + sign * conv]
//                             (residual_block.conv / .forward / .inverse, models/RevResNet.py:79-116)
//   vst_generic_squeeze / _unsqueeze   models/RevResNet.py:34-43 (and channel_reduction's spread, :139-152)
//   vst_generic_copy_channels          split / merge / injective_pad as channel-range copies (:8-31); vst_generic_zero
#include "common.h"

namespace {

constexpr int G_TILE = 16, G_CO = 8, G_CI = 8, G_KMAX = 7;

__device__ __forceinline__ int reflect_any(int v, int n) {          // ReflectionPad2d index for any pad < n
    if (v < 0) v = -v;
    if (v >= n) v = 2 * n - 2 - v;
    return v;
}

// one workgroup = a 16 x 16 output tile x 8 output channels; input channels staged 8 at a time with their halo
__global__ __launch_bounds__(256) void generic_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* old, float sign, int relu,
                                                           float* out, int Cin, int Cout, int H, int W, int Ho, int Wo, int K,
                                                           int stride) {
    extern __shared__ float lds[];
    const int P = (G_TILE - 1) * stride + K;                 // staged patch edge
    float* patch = lds;                                      // [G_CI][P][P]
    float* wl = lds + G_CI * P * P;                          // [G_CO][G_CI][K*K]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, tid = threadIdx.x;
    const int ncot = (Cout + G_CO - 1) / G_CO;
    const int b = blockIdx.z / ncot, co0 = (blockIdx.z % ncot) * G_CO;
    const int ox = blockIdx.x * G_TILE + tx, oy = blockIdx.y * G_TILE + ty;
    const int pad = (K - 1) / 2;
    const int iy0 = blockIdx.y * G_TILE * stride - pad, ix0 = blockIdx.x * G_TILE * stride - pad;
    float acc[G_CO];
#pragma unroll
    for (int c = 0; c < G_CO; ++c) acc[c] = 0.f;
    for (int ci0 = 0; ci0 < Cin; ci0 += G_CI) {
        __syncthreads();
        for (int e = tid; e < G_CI * P * P; e += 256) {
            const int ci = e / (P * P), r = e % (P * P), py = r / P, px = r % P;
            float v = 0.f;
            if (ci0 + ci < Cin) {
                int gy = reflect_any(iy0 + py, H), gx = reflect_any(ix0 + px, W);
                gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);    // (positions that only feed outputs outside the image)
                gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
                v = x[(((size_t)b * Cin + ci0 + ci) * H + gy) * W + gx];
            }
            patch[e] = v;
        }
        for (int e = tid; e < G_CO * G_CI * K * K; e += 256) {
            const int co = e / (G_CI * K * K), r = e % (G_CI * K * K), ci = r / (K * K), t = r % (K * K);
            wl[e] = (co0 + co < Cout && ci0 + ci < Cin) ? w[(((size_t)(co0 + co) * Cin) + ci0 + ci) * K * K + t] : 0.f;
        }
        __syncthreads();
        for (int ci = 0; ci < G_CI; ++ci)
            for (int ky = 0; ky < K; ++ky)
                for (int kx = 0; kx < K; ++kx) {
                    const float v = patch[(ci * P + ty * stride + ky) * P + tx * stride + kx];
#pragma unroll
                    for (int c = 0; c < G_CO; ++c) acc[c] = fmaf(wl[(c * G_CI + ci) * K * K + ky * K + kx], v, acc[c]);
                }
    }
    if (ox >= Wo || oy >= Ho) return;
#pragma unroll
    for (int c = 0; c < G_CO; ++c) {
        if (co0 + c >= Cout) break;
        const size_t o = (((size_t)b * Cout + co0 + c) * Ho + oy) * Wo + ox;
        float v = acc[c] + bias[co0 + c];
        if (relu) v = v > 0.f ? v : 0.f;
        out[o] = old ? old[o] + sign * v : v;
    }
}

// y[b, (i*2+j)*D + d, h, w] = x[b, d, 2h+i, 2w+j]   (TO_SQ), or the inverse copy
template <bool TO_SQ>
__global__ __launch_bounds__(256) void generic_squeeze_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int D,
                                                              int H, int W) {          // H, W: of the UNsqueezed tensor
    const size_t total = (size_t)B * D * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int xw = idx % W;
        size_t r = idx / W;
        const int yh = r % H; r /= H;
        const int d = r % D;
        const int b = r / D;
        const int i = yh & 1, j = xw & 1;
        const size_t sq = ((((size_t)b * 4 * D) + (i * 2 + j) * D + d) * (H / 2) + (yh >> 1)) * (W / 2) + (xw >> 1);
        if (TO_SQ) dst[sq] = src[idx]; else dst[idx] = src[sq];
    }
}

__global__ __launch_bounds__(256) void generic_copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                                    int Csrc, int c0, int n, size_t HW, int Cdst, int d0) {
    const size_t total = (size_t)B * n * HW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t p = idx % HW;
        size_t r = idx / HW;
        const int k = r % n;
        const int b = r / n;
        dst[((size_t)b * Cdst + d0 + k) * HW + p] = src[((size_t)b * Csrc + c0 + k) * HW + p];
    }
}

unsigned grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int vst_generic_conv(const float* x, const float* w, const float* bias, const float* old, float sign, int relu, float* out, int B,
                     int Cin, int Cout, int H, int W, int K, int stride, void* stream) {
    if (!x || !w || !bias || !out) return VST_E_ARG;
    if (B <= 0 || Cin <= 0 || Cout <= 0 || (K & 1) == 0 || K < 1 || K > G_KMAX || (stride != 1 && stride != 2)) return VST_E_SHAPE;
    const int pad = (K - 1) / 2;
    if (H <= pad || W <= pad) return VST_E_SHAPE;            // ReflectionPad2d needs pad < size
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    const int P = (G_TILE - 1) * stride + K;
    const size_t lds = ((size_t)G_CI * P * P + (size_t)G_CO * G_CI * K * K) * sizeof(float);
    const dim3 grid((Wo + G_TILE - 1) / G_TILE, (Ho + G_TILE - 1) / G_TILE, B * ((Cout + G_CO - 1) / G_CO));
    generic_conv_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, old, sign, relu, out, Cin, Cout, H, W, Ho, Wo, K, stride);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_generic_squeeze(const float* x, float* y, int B, int D, int H, int W, void* stream) {
    if (!x || !y) return VST_E_ARG;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return VST_E_SHAPE;
    generic_squeeze_kernel<true><<<grid_for((size_t)B * D * H * W), 256, 0, (hipStream_t)stream>>>(x, y, B, D, H, W);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_generic_unsqueeze(const float* y, float* x, int B, int D, int H, int W, void* stream) {   // D, H, W: of the OUTPUT x
    if (!x || !y) return VST_E_ARG;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return VST_E_SHAPE;
    generic_squeeze_kernel<false><<<grid_for((size_t)B * D * H * W), 256, 0, (hipStream_t)stream>>>(y, x, B, D, H, W);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_generic_copy_channels(const float* src, float* dst, int B, int C_src, int c0, int n, long HW, int C_dst, int d0,
                              void* stream) {
    if (!src || !dst) return VST_E_ARG;
    if (B <= 0 || n <= 0 || HW <= 0 || c0 < 0 || d0 < 0 || c0 + n > C_src || d0 + n > C_dst) return VST_E_SHAPE;
    generic_copy_channels_kernel<<<grid_for((size_t)B * n * HW), 256, 0, (hipStream_t)stream>>>(src, dst, B, C_src, c0, n, (size_t)HW,
                                                                                               C_dst, d0);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_generic_zero(float* dst, size_t n_floats, void* stream) {
    if (!dst) return VST_E_ARG;
    return (int)hipMemsetAsync(dst, 0, n_floats * sizeof(float), (hipStream_t)stream);
}

}  // extern "C"

// Lab luminance-preserving post-process of the delldu fork:
//   out = lab2rgb(cat(L(content), ab(clamp(stylized, 0, 1))))
// project/image_style/vstnet.py:189-220 with the colour maths of project/image_style/color.py:18-113.
// Pointwise and HBM-bound: 9 floats (36 B) per pixel, so one thread does 4 pixels with float4 traffic per plane.
#include "common.h"

namespace {

__device__ __forceinline__ float srgb_to_linear(float v) {                   // color.py:24-25
    return v > 0.04045f ? powf((v + 0.055f) / 1.055f, 2.4f) : v / 12.92f;
}

__device__ __forceinline__ float lab_f(float t) {                            // color.py:44-46
    return t > 0.008856f ? powf(t, (float)(1.0 / 3.0)) : 7.787f * t + (float)(16.0 / 116.0);
}

__device__ __forceinline__ float lab_finv(float f) {                         // color.py:83-85
    return f > 0.2068966f ? f * f * f : (f - (float)(16.0 / 116.0)) / 7.787f;
}

__device__ __forceinline__ float linear_to_srgb(float v) {                   // color.py:66-73
    v = fmaxf(v, 0.0f);
    return v > 0.0031308f ? 1.055f * powf(v, (float)(1.0 / 2.4)) - 0.055f : 12.92f * v;
}

__device__ __forceinline__ float clamp_pm1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }
__device__ __forceinline__ float clamp_01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

__device__ __forceinline__ void rgb_to_xyz(float r, float g, float b, float& x, float& y, float& z) {   // color.py:27-29
    r = srgb_to_linear(r), g = srgb_to_linear(g), b = srgb_to_linear(b);
    x = 0.412453f * r + 0.357580f * g + 0.180423f * b;
    y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
    z = 0.019334f * r + 0.119193f * g + 0.950227f * b;
}

// one pixel: content rgb (c*) and stylised rgb (s*) -> output rgb
__device__ __forceinline__ void luminance_px(float cr, float cg, float cb, float sr, float sg, float sb, float& orr,
                                             float& og, float& ob) {
    float x, y, z;
    rgb_to_xyz(cr, cg, cb, x, y, z);
    const float L_rs = clamp_pm1(((116.0f * lab_f(y) - 16.0f) - 50.0f) / 50.0f);         // only L of the content is used
    rgb_to_xyz(clamp_01(sr), clamp_01(sg), clamp_01(sb), x, y, z);                       // decoder clamp, vstnet.py:322
    const float fx = lab_f(x / 0.95047f), fy = lab_f(y), fz = lab_f(z / 1.08883f);
    const float a_rs = clamp_pm1(500.0f * (fx - fy) / 110.0f);
    const float b_rs = clamp_pm1(200.0f * (fy - fz) / 110.0f);
    // lab2rgb, color.py:76-91,107-113
    const float L = L_rs * 50.0f + 50.0f, a = a_rs * 110.0f, b = b_rs * 110.0f;
    const float yi = (L + 16.0f) / 116.0f;
    const float xi = a / 500.0f + yi;
    const float zi = fmaxf(0.0f, yi - b / 200.0f);
    const float X = lab_finv(xi) * 0.95047f, Y = lab_finv(yi), Z = lab_finv(zi) * 1.08883f;
    orr = clamp_01(linear_to_srgb(3.24048134f * X - 1.53715152f * Y - 0.49853633f * Z));
    og = clamp_01(linear_to_srgb(-0.96925495f * X + 1.87599f * Y + 0.04155593f * Z));
    ob = clamp_01(linear_to_srgb(0.05564664f * X - 0.20404134f * Y + 1.05731107f * Z));
}

// planes: [B][3][n] fp32, n = H*W.  VEC=4 needs n % 4 == 0 (keeps every plane 16-byte aligned).
template <int VEC>
__global__ __launch_bounds__(256) void lab_luminance_kernel(const float* __restrict__ content,
                                                            const float* __restrict__ stylized,
                                                            float* __restrict__ out, int n) {
    const size_t img = (size_t)blockIdx.y * 3 * n;
    const int p = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= n) return;
    float c[3][VEC], s[3][VEC], o[3][VEC];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (VEC == 4) {
            const float4 cv = *reinterpret_cast<const float4*>(content + img + (size_t)ch * n + p);
            const float4 sv = *reinterpret_cast<const float4*>(stylized + img + (size_t)ch * n + p);
            c[ch][0] = cv.x, c[ch][1 % VEC] = cv.y, c[ch][2 % VEC] = cv.z, c[ch][3 % VEC] = cv.w;
            s[ch][0] = sv.x, s[ch][1 % VEC] = sv.y, s[ch][2 % VEC] = sv.z, s[ch][3 % VEC] = sv.w;
        } else {
            c[ch][0] = content[img + (size_t)ch * n + p];
            s[ch][0] = stylized[img + (size_t)ch * n + p];
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) luminance_px(c[0][v], c[1][v], c[2][v], s[0][v], s[1][v], s[2][v], o[0][v], o[1][v], o[2][v]);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (VEC == 4)
            *reinterpret_cast<float4*>(out + img + (size_t)ch * n + p) =
                make_float4(o[ch][0], o[ch][1 % VEC], o[ch][2 % VEC], o[ch][3 % VEC]);
        else
            out[img + (size_t)ch * n + p] = o[ch][0];
    }
}

}  // namespace

extern "C" int vst_lab_luminance(const float* content, const float* stylized, float* out, int B, int H, int W,
                                 void* stream) {
    if (!content || !stylized || !out) return VST_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 4 || B > 65535) return VST_E_SHAPE;
    const int n = H * W;
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (((uintptr_t)content | (uintptr_t)stylized | (uintptr_t)out) & 15) == 0;
    if ((n & 3) == 0 && aligned)
        lab_luminance_kernel<4><<<dim3((n / 4 + 255) / 256, B), 256, 0, st>>>(content, stylized, out, n);
    else
        lab_luminance_kernel<1><<<dim3((n + 255) / 256, B), 256, 0, st>>>(content, stylized, out, n);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

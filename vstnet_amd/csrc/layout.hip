// Layout glue between the reference's NCHW tensors and the ZC state layout, and weight packing.
//
// Reference semantics restated here (paths relative to the reference root):
//   split/merge            models/RevResNet.py:8-16    -> the two state halves s1 / s2
//   injective_pad          models/RevResNet.py:19-31   -> zero channels 3..15 of s1, s2 = 0
//   squeeze/unsqueeze      models/RevResNet.py:34-43   -> identity on the ZC layout
//   channel_reduction "spread" :139-144 / its inverse :148-154 -> vst_spread / vst_gather
#include "common.h"

// ------------------------------------------------------------------------------------------------
// x[B,C,H,W] -> s1 (view level 0, 16 channels, C..15 zero)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, float* __restrict__ s1,
                                                         int C, int H, int W, const float* __restrict__ addk) {
    const int xg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (xg >= W || yg >= H) return;
    const size_t plane = (size_t)H * W;
    const float* src = x + (size_t)b * C * plane + (size_t)yg * W + xg;
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = (c < C ? src[c * plane] : 0.f) + (addk ? addk[c] : 0.f);
    float* dst = s1 + (size_t)b * plane * 16 + zc_offset(0, yg, xg, W >> 2);
#pragma unroll
    for (int c = 0; c < 16; c += 4) *(float4*)(dst + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
}

__global__ __launch_bounds__(256) void unpack_output_kernel(const float* __restrict__ s1, float* __restrict__ x,
                                                            int C, int H, int W) {
    const int xg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (xg >= W || yg >= H) return;
    const size_t plane = (size_t)H * W;
    const float* src = s1 + (size_t)b * plane * 16 + zc_offset(0, yg, xg, W >> 2);
    float* dst = x + (size_t)b * C * plane + (size_t)yg * W + xg;
    for (int c = 0; c < C; ++c) dst[c * plane] = src[c];
}

// ------------------------------------------------------------------------------------------------
// uint8 HWC frame edge.  In: transforms.ToTensor (image_transfer.py:167, video_transfer.py:188):
// x = u8 / 255 as fp32, HWC -> channel planes.  Out: grid.mul(255).clamp(0, 255).byte() with truncation + CHW -> HWC
// (image_transfer.py:217-218, video_transfer.py:212).  Frames cross PCIe as 3 bytes per pixel instead of 12.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_input_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ s1,
                                                            int H, int W, const float* __restrict__ addk) {
    const int xg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (xg >= W || yg >= H) return;
    const uint8_t* src = img + ((size_t)b * H * W + (size_t)yg * W + xg) * 3;
    float* dst = s1 + (size_t)b * H * W * 16 + zc_offset(0, yg, xg, W >> 2);
    float k[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) k[c] = addk ? addk[c] : 0.f;
    *(float4*)(dst) = make_float4((float)src[0] / 255.f + k[0], (float)src[1] / 255.f + k[1], (float)src[2] / 255.f + k[2], k[3]);
    *(float4*)(dst + 4) = make_float4(k[4], k[5], k[6], k[7]);
    *(float4*)(dst + 8) = make_float4(k[8], k[9], k[10], k[11]);
    *(float4*)(dst + 12) = make_float4(k[12], k[13], k[14], k[15]);
}

__global__ __launch_bounds__(256) void unpack_output_u8_kernel(const float* __restrict__ s1, uint8_t* __restrict__ img,
                                                               int H, int W) {
    const int xg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (xg >= W || yg >= H) return;
    const float4 v = *(const float4*)(s1 + (size_t)b * H * W * 16 + zc_offset(0, yg, xg, W >> 2));
    uint8_t* dst = img + ((size_t)b * H * W + (size_t)yg * W + xg) * 3;
    const float c[3] = {v.x, v.y, v.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = c[k] * 255.f;
        t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);       // NaN falls through both compares like torch.clamp keeps NaN; byte() of it is 0
        dst[k] = (uint8_t)t;                               // truncation toward zero
    }
}

// ------------------------------------------------------------------------------------------------
// Forward block 0 sees x2 = 0 (injective_pad, RevResNet.py:212-214), so y1 = x1 + F(0) and F(0) is one constant per
// channel: conv of zeros = bias, reflection padding of a constant map is the same constant, hence
//   h1 = relu(b1),  h2 = relu(b2 + sum_taps W2 h1),  k = b3 + sum_taps W3 h2.
// One tiny workgroup evaluates k (fp32, taps-major fp32 weight sections); the pack kernels add it, and the forward
// pass skips block 0's three convolutions.
// ------------------------------------------------------------------------------------------------
__global__ void block0_const_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                    const float* __restrict__ b1, const float* __restrict__ b2,
                                    const float* __restrict__ b3, float* __restrict__ k) {
    __shared__ float h1[4], h2[4];
    const int t = threadIdx.x;
    if (t < 4) h1[t] = b1[t] > 0.f ? b1[t] : 0.f;
    __syncthreads();
    if (t < 4) {
        float acc = b2[t];
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < 4; ++ci) acc = fmaf(w2[(tap * 4 + ci) * 4 + t], h1[ci], acc);      // [tap][ci][co], co = 4
        h2[t] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    if (t < 16) {
        float acc = b3[t];
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < 4; ++ci) acc = fmaf(w3[(tap * 4 + ci) * 16 + t], h2[ci], acc);     // co = 16
        k[t] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// spread / gather.  m = cat(s1, s2) has 512 channels per quarter-res cell.
//   SP == 2:  z[d32 , 4h+2i+i', 4w+2j+j'] = m[(2i+j)*128 + (2i'+j')*32 + d32 ][h][w]
//   SP == 1:  z[d128, 2h+i    , 2w+j    ] = m[(2i+j)*128 + d128             ][h][w]
// i selects the half (i=0 -> s1, i=1 -> s2).  One workgroup moves 16 cells x both halves through a
// skewed LDS image so that both the state side (float4 per lane) and the z side (one row segment per
// wave) are coalesced and the z-side LDS accesses hit 32 distinct banks per half-wave.
// ------------------------------------------------------------------------------------------------
template <int SP>
struct SpreadGeom {
    static constexpr int CELL_STRIDE = SP == 2 ? 324 : 258;  // == 4 (resp. 2) mod 32
    __device__ static __forceinline__ int lds_index(int slot, int c) {
        if (SP == 2) {
            const int g = c >> 5, d = c & 31;                  // g = 4j + 2i' + j'
            return slot * 324 + (g >> 2) * 162 + ((g >> 1) & 1) * 66 + (g & 1) * 33 + d;   // disjoint 32-float runs
        } else {
            return slot * 258 + (c >> 7) * 129 + (c & 127);
        }
    }
};

// PLANES (gather only): the first half is written as fp16 hi / lo split planes (common.h, SP layout) into `sp1` instead of
// fp32 into s1 - the inverse pass's first 256-channel block reads its src only through them (conv3.hip).
template <int SP, bool TO_Z, bool PLANES = false>
__global__ __launch_bounds__(256) void spread_gather_kernel(float* __restrict__ s1, float* __restrict__ s2,
                                                            float* __restrict__ z, int H, int W,
                                                            unsigned char* __restrict__ sp1 = nullptr) {
    __shared__ float lds[32 * SpreadGeom<SP>::CELL_STRIDE];
    const int Hq = H >> 2, Wq = W >> 2;
    const int w0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int t = threadIdx.x;
    const size_t img = (size_t)Hq * Wq * 256;
    float* half[2] = {s1 + (size_t)b * img, s2 + (size_t)b * img};

    auto state_phase = [&]() {
        if (PLANES) {                                          // half 0: 16 cells x 32 eight-channel groups, both planes
            unsigned char* const sp_img = sp1 + (size_t)b * img * 4;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = it * 256 + t;
                const int cell = idx & 15, cig = idx >> 4;
                const int w = w0 + cell;
                if (w >= Wq) continue;
                const int cot = cig >> 3, j = (cig >> 2) & 1, kg = cig & 3;
                const int li = SpreadGeom<SP>::lds_index(cell, cot * 64 + j * 32 + kg * 4);
                const float f8[8] = {lds[li], lds[li + 1], lds[li + 2], lds[li + 3], lds[li + 16], lds[li + 17], lds[li + 18], lds[li + 19]};
                u32x4 hi, lo;
                split8_sp(f8, hi, lo);
                *(u32x4*)(sp_img + sp_offset(cig, 0, h, w, Hq, Wq)) = hi;
                *(u32x4*)(sp_img + sp_offset(cig, 1, h, w, Hq, Wq)) = lo;
            }
        }
#pragma unroll
        for (int it = PLANES ? 4 : 0; it < 8; ++it) {
            const int idx = it * 256 + t;
            const int slot = idx >> 6, q = idx & 63;
            const int w = w0 + (slot & 15);
            if (w >= Wq) continue;
            float* p = half[slot >> 4] + ((size_t)h * Wq + w) * 256 + q * 4;
            const int li = SpreadGeom<SP>::lds_index(slot, q * 4);
            if (TO_Z) {
                const float4 v = *(const float4*)p;
                lds[li] = v.x; lds[li + 1] = v.y; lds[li + 2] = v.z; lds[li + 3] = v.w;
            } else {
                *(float4*)p = make_float4(lds[li], lds[li + 1], lds[li + 2], lds[li + 3]);
            }
        }
    };
    auto z_phase = [&]() {
        if (SP == 2) {
            const int X = t & 63, wave = t >> 6;
            const int wl = X >> 2, j = (X >> 1) & 1, jp = X & 1;
            const int xg = w0 * 4 + X;
#pragma unroll 8
            for (int it = 0; it < 32; ++it) {                 // (unrolled: keeps 8 row-segment loads in flight when gathering)
                const int combo = it * 4 + wave;          // (d32, Yl)
                const int d = combo >> 2, Yl = combo & 3;
                const int i = Yl >> 1, ip = Yl & 1;
                if (xg >= W) continue;
                float* p = z + (((size_t)b * 32 + d) * H + (4 * h + Yl)) * W + xg;
                const int li = (i * 16 + wl) * 324 + j * 162 + ip * 66 + jp * 33 + d;
                if (TO_Z) *p = lds[li]; else lds[li] = *p;
            }
        } else {
            const int H2 = H >> 1, W2 = W >> 1;
            const int X = t & 31, dsel = (t >> 5) & 1, wave = t >> 6;
            const int wl = X >> 1, j = X & 1;
            const int xg = w0 * 2 + X;
#pragma unroll 8
            for (int it = 0; it < 32; ++it) {
                const int combo = (it * 4 + wave) * 2 + dsel;   // (d128, i)
                const int d = combo >> 1, i = combo & 1;
                if (xg >= W2) continue;
                float* p = z + (((size_t)b * 128 + d) * H2 + (2 * h + i)) * W2 + xg;
                const int li = (i * 16 + wl) * 258 + j * 129 + d;
                if (TO_Z) *p = lds[li]; else lds[li] = *p;
            }
        }
    };
    if (TO_Z) { state_phase(); __syncthreads(); z_phase(); }
    else      { z_phase(); __syncthreads(); state_phase(); }
}

// ------------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> [fp32 [tap][ci][co] | bf16 hi [kstep][kg][coutp][8] | bf16 lo ...]
// K ordering of the MFMA kernel (conv_mfma.hip, k_to_tap_ci):
//   cin >= 32: kstep = chunk*9 + tap, k = kg*8+j -> ci = chunk*32 + k
//   cin == 16: kstep s covers taps 2s, 2s+1:   tap = 2s + (kg>>1), ci = (kg&1)*8 + j
//   cin ==  4: kstep s covers taps 8s..8s+7:   tap = 8s + 2kg + (j>>2), ci = j&3
// taps > 8 and co >= cout are zero.
// ------------------------------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* __restrict__ w, int cout, int cin, unsigned char* __restrict__ packed) {
    const PackedConvLayout L = packed_conv_layout(cout, cin);
    const size_t n_f32 = (size_t)9 * cin * cout;
    const size_t n_frag = (size_t)L.ksteps * 4 * L.coutp * 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_f32) {
        const int co = idx % cout;
        const int ci = (idx / cout) % cin;
        const int tap = idx / ((size_t)cout * cin);
        ((float*)packed)[idx] = w[((size_t)co * cin + ci) * 9 + tap];
    }
    if (idx < n_frag) {
        const int j = idx & 7;
        const int co = (idx >> 3) % L.coutp;
        const int kg = (idx >> 3) / L.coutp % 4;
        const int ks = (idx >> 3) / L.coutp / 4;
        int tap, ci;
        if (cin >= 32) { tap = ks % 9; ci = (ks / 9) * 32 + kg * 8 + j; }
        else if (cin == 16) { tap = 2 * ks + (kg >> 1); ci = (kg & 1) * 8 + j; }
        else { tap = 8 * ks + 2 * kg + (j >> 2); ci = j & 3; }
        float v = 0.f;
        if (tap < 9 && co < cout && ci < cin) v = w[((size_t)co * cin + ci) * 9 + tap];
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        ((__bf16*)(packed + L.f32_bytes))[idx] = hi;
        ((__bf16*)(packed + L.f32_bytes + L.frag_bytes))[idx] = lo;
        ((_Float16*)(packed + L.f16_offset))[idx] = (_Float16)v;      // the 2-term kernels: w rounded to fp16, once
        if (fabsf(v) > 65504.f) atomicOr(&vst_tu_range_flags, VST_RANGE_WEIGHT);
        if (L.sp_sections) {
            // conv3.hip: the 8 channels of k-group kg are the ones a producer lane holds, {4kg + r, 16 + 4kg + r} of the chunk
            const int cip = (ks / 9) * 32 + (j >> 2) * 16 + kg * 4 + (j & 3);
            const float vp = co < cout ? w[((size_t)co * cin + cip) * 9 + tap] : 0.f;
            unsigned char* sp = packed + L.f32_bytes + 2 * L.frag_bytes;
            ((_Float16*)sp)[idx] = (_Float16)vp;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// vst_normalize_block: per-channel power-of-two scales of a residual block's two intermediates (vstnet.h).  One workgroup;
// the tensors are small (at most 64 x 256 x 9 floats).  Phase 1: s1 from the row norms of W1, applied to W1 / b1 rows and
// W4 columns; phase 2: s2 from the row norms of the rescaled W4, applied to W4 / b4 rows and W7 columns.
// s = 2^-round(log2(norm)) through the norm's exponent bits (no transcendental: the same on every device), 1 for a zero row,
// exponents clamped to +-40.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_scale_of_norm(float sumsq) {
    const float nrm = sqrtf(sumsq);
    if (!(nrm > 0.f) || !(nrm < 3.0e38f)) return 1.f;
    int e;
    const float m = frexpf(nrm, &e);                 // nrm = m 2^e, m in [0.5, 1): round(log2 nrm) = e if m >= 1/sqrt(2) else e - 1
    int r = m >= 0.70710678f ? e : e - 1;
    r = r > 40 ? 40 : (r < -40 ? -40 : r);
    return ldexpf(1.f, -r);
}

__global__ __launch_bounds__(256) void normalize_block_kernel(float* __restrict__ w1, float* __restrict__ b1, float* __restrict__ w4,
                                                              float* __restrict__ b4, float* __restrict__ w7, int cin1, int mid,
                                                              int cout, float* __restrict__ scales) {
    __shared__ float sc[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto row_scales = [&](const float* w, int K) {
        for (int r = wave; r < mid; r += 4) {
            float acc = 0.f;
            for (int k = lane; k < K; k += 64) { const float v = w[(size_t)r * K + k]; acc += v * v; }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) sc[r] = pow2_scale_of_norm(acc);
        }
        __syncthreads();
    };
    const int K1 = cin1 * 9, K4 = mid * 9;
    row_scales(w1, K1);
    for (int i = tid; i < mid * K1; i += 256) w1[i] *= sc[i / K1];
    for (int i = tid; i < mid; i += 256) { b1[i] *= sc[i]; if (scales) scales[i] = sc[i]; }
    for (int i = tid; i < mid * K4; i += 256) w4[i] /= sc[(i / 9) % mid];          // OIHW: input channel = (i / 9) % mid
    __syncthreads();
    row_scales(w4, K4);
    for (int i = tid; i < mid * K4; i += 256) w4[i] *= sc[i / K4];
    for (int i = tid; i < mid; i += 256) { b4[i] *= sc[i]; if (scales) scales[mid + i] = sc[i]; }
    for (int i = tid; i < cout * K4; i += 256) w7[i] /= sc[(i / 9) % mid];
}

VST_DEFINE_TU_RANGE(vst_range_tu_layout)

// ------------------------------------------------------------------------------------------------
extern "C" {

int vst_version(void) { return 102; }

int vst_normalize_block(float* w1, float* b1, float* w4, float* b4, float* w7, int c_in1, int c_mid, int c_out, float* scales,
                        void* stream) {
    if (!w1 || !b1 || !w4 || !b4 || !w7) return VST_E_ARG;
    if (c_in1 <= 0 || c_mid <= 0 || c_mid > 64 || c_out <= 0) return VST_E_SHAPE;
    normalize_block_kernel<<<dim3(1), 256, 0, (hipStream_t)stream>>>(w1, b1, w4, b4, w7, c_in1, c_mid, c_out, scales);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

const char* vst_error_string(int code) {
    switch (code) {
        case VST_OK: return "ok";
        case VST_E_ARG: return "invalid argument (null pointer or non-positive size)";
        case VST_E_SHAPE: return "unsupported shape (H, W must be multiples of 4 and >= 8; channels in the documented sets)";
        case VST_E_MODE: return "unknown mode (precision / sp_steps / direction)";
        case VST_E_WORKSPACE: return "workspace too small or null";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown vstnet error";
    }
}

size_t vst_conv_packed_bytes(int cout, int cin) {
    const PackedConvLayout L = packed_conv_layout(cout, cin);
    return L.f16_offset + L.frag_bytes;
}

int vst_pack_conv(const float* w, int cout, int cin, void* packed, void* stream) {
    if (!w || !packed || cout <= 0) return VST_E_ARG;
    if (!(cin == 4 || cin == 16 || (cin >= 32 && cin % 32 == 0))) return VST_E_SHAPE;
    const PackedConvLayout L = packed_conv_layout(cout, cin);
    size_t n = (size_t)9 * cin * cout;
    const size_t n_frag = (size_t)L.ksteps * 4 * L.coutp * 8;
    if (n_frag > n) n = n_frag;
    pack_conv_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(w, cout, cin, (unsigned char*)packed);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_pack_input_k(const float* x, const uint8_t* x_u8, float* s1, float* s2, int B, int C, int H, int W,
                     const float* addk, void* stream) {
    if ((!x && !x_u8) || !s1 || !s2) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W) || C < 1 || C > 16) return VST_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(s2, 0, (size_t)B * H * W * 16 * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    vst_prof_scope prof(VST_KERNEL_PACK, st);
    if (x_u8) pack_input_u8_kernel<<<grid, 256, 0, st>>>(x_u8, s1, H, W, addk);
    else pack_input_kernel<<<grid, 256, 0, st>>>(x, s1, C, H, W, addk);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_block0_const(const vst_block_weights* w0, float* k16, void* stream) {
    if (!w0 || !k16) return VST_E_ARG;
    block0_const_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const float*)w0->conv[1].packed, (const float*)w0->conv[2].packed,
                                                           w0->conv[0].bias, w0->conv[1].bias, w0->conv[2].bias, k16);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_pack_input(const float* x, float* s1, float* s2, int B, int C, int H, int W, void* stream) {
    if (!x) return VST_E_ARG;
    return vst_pack_input_k(x, nullptr, s1, s2, B, C, H, W, nullptr, stream);
}

int vst_unpack_output(const float* s1, float* x, int B, int C, int H, int W, void* stream) {
    if (!x || !s1) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W) || C < 1 || C > 16) return VST_E_SHAPE;
    vst_prof_scope prof(VST_KERNEL_UNPACK, (hipStream_t)stream);
    unpack_output_kernel<<<dim3((W + 63) / 64, (H + 3) / 4, B), 256, 0, (hipStream_t)stream>>>(s1, x, C, H, W);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_pack_input_u8(const uint8_t* frames_hwc, float* s1, float* s2, int B, int H, int W, void* stream) {
    if (!frames_hwc) return VST_E_ARG;
    return vst_pack_input_k(nullptr, frames_hwc, s1, s2, B, 3, H, W, nullptr, stream);
}

int vst_unpack_output_u8(const float* s1, uint8_t* frames_hwc, int B, int H, int W, void* stream) {
    if (!frames_hwc || !s1) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W)) return VST_E_SHAPE;
    vst_prof_scope prof(VST_KERNEL_UNPACK, (hipStream_t)stream);
    unpack_output_u8_kernel<<<dim3((W + 63) / 64, (H + 3) / 4, B), 256, 0, (hipStream_t)stream>>>(s1, frames_hwc, H, W);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_spread(const float* s1, const float* s2, float* z, int B, int H, int W, int sp_steps, void* stream) {
    if (!s1 || !s2 || !z) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W)) return VST_E_SHAPE;
    const dim3 grid((W / 4 + 15) / 16, H / 4, B);
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_SPREAD, st);
    if (sp_steps == 2) spread_gather_kernel<2, true><<<grid, 256, 0, st>>>((float*)s1, (float*)s2, z, H, W);
    else if (sp_steps == 1) spread_gather_kernel<1, true><<<grid, 256, 0, st>>>((float*)s1, (float*)s2, z, H, W);
    else return VST_E_MODE;
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

// gather for the f16x2 inverse pass: s1 only as split planes (`s1_planes`, 1024 B per quarter-res pixel and image), s2 in fp32
int vst3_gather_planes(const float* z, unsigned char* s1_planes, float* s2, int B, int H, int W, int sp_steps, void* stream) {
    if (!s1_planes || !s2 || !z) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W)) return VST_E_SHAPE;
    const dim3 grid((W / 4 + 15) / 16, H / 4, B);
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_GATHER, st);
    if (sp_steps == 2) spread_gather_kernel<2, false, true><<<grid, 256, 0, st>>>(nullptr, s2, (float*)z, H, W, s1_planes);
    else if (sp_steps == 1) spread_gather_kernel<1, false, true><<<grid, 256, 0, st>>>(nullptr, s2, (float*)z, H, W, s1_planes);
    else return VST_E_MODE;
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_gather(const float* z, float* s1, float* s2, int B, int H, int W, int sp_steps, void* stream) {
    if (!s1 || !s2 || !z) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W)) return VST_E_SHAPE;
    const dim3 grid((W / 4 + 15) / 16, H / 4, B);
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_GATHER, st);
    if (sp_steps == 2) spread_gather_kernel<2, false><<<grid, 256, 0, st>>>(s1, s2, (float*)z, H, W);
    else if (sp_steps == 1) spread_gather_kernel<1, false><<<grid, 256, 0, st>>>(s1, s2, (float*)z, H, W);
    else return VST_E_MODE;
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

}  // extern "C"

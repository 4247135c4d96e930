// 3x3 reflect-padded convolutions of the coupling blocks and the block / whole-pass runners.
//
// Reference semantics (paths relative to the reference root):
//   residual_block.conv     models/RevResNet.py:79-88   ReflectionPad2d(1)+Conv2d(3x3,bias) x3, ReLU between
//   residual_block.forward  models/RevResNet.py:96-104  (x1,x2) -> (x2, F(x2)+x1)   [stride 2: squeeze both]
//   residual_block.inverse  models/RevResNet.py:106-116 (x2,y1) -> (y1-F(x2), x2)
//   RevResNet._forward/_inverse  models/RevResNet.py:210-239, channel_reduction :131-163
//
// Every conv is an implicit GEMM  D[pixel][co] = sum_{tap,ci} A[pixel+tap][ci] * W[tap][ci][co]
// on v_mfma_f32_16x16x32_bf16 with split operands: a = a_hi + a_lo, w = w_hi + w_lo (bf16 each),
// D += a_hi*w_hi + a_lo*w_hi + a_hi*w_lo, fp32 accumulate (error ~2^-17 per product; the state in
// HBM stays fp32).  A workgroup (4 waves) owns a 16-wide x (4*MR)-high tile of output pixels and
// NT output channels; activations are staged fp32 -> {hi,lo} bf16 into an LDS image
// [channel-group of 8][slot][8 x bf16] in which the 16 pixels of an MFMA row block are consecutive
// 16-byte slots (bank-conflict-free ds_read_b128 for every tap shift); weights are pre-packed in
// fragment order (layout.hip) and copied straight into LDS.
#include "common.h"

struct ConvArgs {
    const float* in;
    float* out;
    const unsigned char* packed;
    const float* bias;
    int Hin, Win, Hout, Wout;
    int Wq;                      // W/4 of the full image (state addressing)
    size_t in_img_stride;        // floats per image
    size_t out_img_stride;
    float sign;                  // OUT_STATE: out += sign * (conv + bias)
};

template <int CIN, int COUT, int STRIDE>
struct ConvCfg {
    static constexpr int NT = COUT >= 64 ? 64 : 16;      // output channels per workgroup
    static constexpr int NB = NT / 16;                   // 16-wide N blocks per wave
    static constexpr int COUTP = (COUT + 15) / 16 * 16;
    static constexpr int NCOT = COUTP / NT;              // co tiles (grid.z factor)
    static constexpr int MR = STRIDE == 2 ? 2 : 4;       // tile rows (16-pixel M blocks) per wave
    static constexpr int TH = 4 * MR, TW = 16;
    static constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
    static constexpr int NSLOT = (IH * IW + 15) / 16 * 16;
    static constexpr int CC = CIN >= 32 ? 32 : CIN;      // input channels staged per chunk
    static constexpr int NCHUNK = CIN / CC;
    static constexpr int CIG = CC >= 8 ? CC / 8 : 1;     // 8-channel groups per chunk
    static constexpr int KS = CIN >= 32 ? 9 : (CIN == 16 ? 5 : 2);   // 32-deep k steps per chunk
    static constexpr int A_PLANE = CIN == 4 ? NSLOT * 8 : CIG * NSLOT * 16;
    static constexpr int B_PLANE = KS * 4 * NT * 16;
    static constexpr int LDS_BYTES = 2 * A_PLANE + 2 * B_PLANE;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ void split8(const float4 v0, const float4 v1, uint4& hi, uint4& lo) {
    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (__bf16)f[i];
        l[i] = (__bf16)(f[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
}

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    bf16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (__bf16)f[i];
        l[i] = (__bf16)(f[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

template <int CIN, int COUT, int STRIDE, bool IN_STATE, bool OUT_STATE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    using C = ConvCfg<CIN, COUT, STRIDE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const a_hi = smem;
    unsigned char* const a_lo = smem + C::A_PLANE;
    unsigned char* const b_hi = smem + 2 * C::A_PLANE;
    unsigned char* const b_lo = b_hi + C::B_PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int tx0 = blockIdx.x * C::TW, ty0 = blockIdx.y * C::TH;
    const int b = blockIdx.z / C::NCOT, co0 = (blockIdx.z % C::NCOT) * C::NT;

    const float* const in_img = a.in + (size_t)b * a.in_img_stride;
    const PackedConvLayout PL = packed_conv_layout(COUT, CIN);
    const unsigned char* const w_hi = a.packed + PL.f32_bytes;
    const unsigned char* const w_lo = w_hi + PL.frag_bytes;

    f32x4 acc[C::MR][C::NB];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < C::NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane slot of tile row (wave*MR), pixel lrow, tap (0,0)
    const int slot_base = (wave * C::MR * STRIDE) * C::IW + lrow * STRIDE;

    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
        if (chunk > 0) __syncthreads();
        // ---- stage activations: fp32 -> bf16 hi/lo image ------------------------------------------
        if (CIN == 4) {
            constexpr int ITEMS = C::IH * C::IW;
#pragma unroll
            for (int it = 0; it < (ITEMS + 255) / 256; ++it) {
                const int slot = it * 256 + tid;
                if (slot < ITEMS) {
                    const int iy = slot / C::IW, ix = slot - iy * C::IW;
                    const int gy = reflect_clamp(ty0 * STRIDE - 1 + iy, a.Hin);
                    const int gx = reflect_clamp(tx0 * STRIDE - 1 + ix, a.Win);
                    const float4 v = *(const float4*)(in_img + ((size_t)gy * a.Win + gx) * 4);
                    uint2 h, l;
                    split4(v, h, l);
                    *(uint2*)(a_hi + slot * 8) = h;
                    *(uint2*)(a_lo + slot * 8) = l;
                }
            }
        } else {
            constexpr int ITEMS = C::IH * C::IW * C::CIG;
#pragma unroll
            for (int it = 0; it < (ITEMS + 255) / 256; ++it) {
                const int idx = it * 256 + tid;
                if (idx < ITEMS) {
                    const int cig = idx % C::CIG, slot = idx / C::CIG;
                    const int iy = slot / C::IW, ix = slot - iy * C::IW;
                    const int gy = reflect_clamp(ty0 * STRIDE - 1 + iy, a.Hin);
                    const int gx = reflect_clamp(tx0 * STRIDE - 1 + ix, a.Win);
                    const size_t off = IN_STATE ? zc_offset(vst_level_of_channels(CIN), gy, gx, a.Wq)
                                                : ((size_t)gy * a.Win + gx) * CIN;
                    const float* p = in_img + off + chunk * C::CC + cig * 8;
                    const float4 v0 = *(const float4*)p;
                    const float4 v1 = *(const float4*)(p + 4);
                    uint4 h, l;
                    split8(v0, v1, h, l);
                    *(uint4*)(a_hi + (cig * C::NSLOT + slot) * 16) = h;
                    *(uint4*)(a_lo + (cig * C::NSLOT + slot) * 16) = l;
                }
            }
        }
        // ---- stage weights: packed fragments -> LDS ------------------------------------------------
        {
            constexpr int ITEMS = C::KS * 4 * C::NT;
#pragma unroll
            for (int it = 0; it < (ITEMS + 255) / 256; ++it) {
                const int idx = it * 256 + tid;
                if (idx < ITEMS) {
                    const int co = idx % C::NT, r = idx / C::NT;
                    const size_t src = ((size_t)(chunk * C::KS * 4 + r) * C::COUTP + co0 + co) * 16;
                    *(uint4*)(b_hi + idx * 16) = *(const uint4*)(w_hi + src);
                    *(uint4*)(b_lo + idx * 16) = *(const uint4*)(w_lo + src);
                }
            }
        }
        __syncthreads();

        // ---- MFMA over the chunk's k steps ---------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            bf16x8 bh[C::NB], bl[C::NB];
#pragma unroll
            for (int n = 0; n < C::NB; ++n) {
                const int boff = ((ks * 4 + kg) * C::NT + n * 16 + lrow) * 16;
                bh[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(b_hi + boff));
                bl[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(b_lo + boff));
            }
#pragma unroll
            for (int m = 0; m < C::MR; ++m) {
                bf16x8 ah, al;
                if (CIN >= 32) {
                    const int dy = ks / 3, dx = ks - dy * 3;
                    const int slot = slot_base + (m * STRIDE + dy) * C::IW + dx;
                    const int aoff = (kg * C::NSLOT + slot) * 16;
                    ah = __builtin_bit_cast(bf16x8, *(const uint4*)(a_hi + aoff));
                    al = __builtin_bit_cast(bf16x8, *(const uint4*)(a_lo + aoff));
                } else if (CIN == 16) {
                    int tap = 2 * ks + (kg >> 1);
                    tap = tap > 8 ? 8 : tap;
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const int slot = slot_base + (m * STRIDE + dy) * C::IW + dx;
                    const int aoff = ((kg & 1) * C::NSLOT + slot) * 16;
                    ah = __builtin_bit_cast(bf16x8, *(const uint4*)(a_hi + aoff));
                    al = __builtin_bit_cast(bf16x8, *(const uint4*)(a_lo + aoff));
                } else {
                    int t0 = 8 * ks + 2 * kg;
                    t0 = t0 > 8 ? 8 : t0;
                    const int t1 = t0 + 1 > 8 ? 8 : t0 + 1;
                    const int s0 = slot_base + (m * STRIDE + t0 / 3) * C::IW + t0 % 3;
                    const int s1 = slot_base + (m * STRIDE + t1 / 3) * C::IW + t1 % 3;
                    uint4 h, l;
                    const uint2 h0 = *(const uint2*)(a_hi + s0 * 8), h1 = *(const uint2*)(a_hi + s1 * 8);
                    const uint2 l0 = *(const uint2*)(a_lo + s0 * 8), l1 = *(const uint2*)(a_lo + s1 * 8);
                    h = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    l = make_uint4(l0.x, l0.y, l1.x, l1.y);
                    ah = __builtin_bit_cast(bf16x8, h);
                    al = __builtin_bit_cast(bf16x8, l);
                }
#pragma unroll
                for (int n = 0; n < C::NB; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[m][n], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: D[row = pixel 4*kg + r][col = channel lrow] -----------------------------------------
    float* const out_img = a.out + (size_t)b * a.out_img_stride;
#pragma unroll
    for (int n = 0; n < C::NB; ++n) {
        const int co = co0 + n * 16 + lrow;
        if (co >= COUT) continue;
        const float bias = a.bias[co];
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
            const int oy = ty0 + wave * C::MR + m;
            if (oy >= a.Hout) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = tx0 + kg * 4 + r;
                if (ox >= a.Wout) continue;
                const float v = acc[m][n][r] + bias;
                if (OUT_STATE) {
                    float* p = out_img + zc_offset(vst_level_of_channels(COUT), oy, ox, a.Wq) + co;
                    *p = *p + a.sign * v;
                } else {
                    out_img[((size_t)oy * a.Wout + ox) * COUT + co] = v > 0.f ? v : 0.f;
                }
            }
        }
    }
}

// Diagnostic fp32 direct convolution (VST_PREC_FP32): one thread per (pixel, co), plain FMA chain.
template <bool IN_STATE, bool OUT_STATE>
__global__ __launch_bounds__(256) void conv_fp32_kernel(const ConvArgs a, int CIN, int COUT, int STRIDE, int B) {
    const size_t total = (size_t)B * a.Hout * a.Wout * COUT;
    const float* wf = (const float*)a.packed;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int co = idx % COUT;
        size_t rest = idx / COUT;
        const int ox = rest % a.Wout; rest /= a.Wout;
        const int oy = rest % a.Hout;
        const int b = rest / a.Hout;
        const float* in_img = a.in + (size_t)b * a.in_img_stride;
        float acc = a.bias[co];
        for (int tap = 0; tap < 9; ++tap) {
            const int gy = reflect_clamp(oy * STRIDE - 1 + tap / 3, a.Hin);
            const int gx = reflect_clamp(ox * STRIDE - 1 + tap % 3, a.Win);
            const float* p = in_img + (IN_STATE ? zc_offset(vst_level_of_channels(CIN), gy, gx, a.Wq)
                                                : ((size_t)gy * a.Win + gx) * CIN);
            const float* w = wf + (size_t)tap * CIN * COUT + co;
            for (int ci = 0; ci < CIN; ++ci) acc = fmaf(p[ci], w[(size_t)ci * COUT], acc);
        }
        float* out_img = a.out + (size_t)b * a.out_img_stride;
        if (OUT_STATE) {
            float* p = out_img + zc_offset(vst_level_of_channels(COUT), oy, ox, a.Wq) + co;
            *p = *p + a.sign * acc;
        } else {
            out_img[((size_t)oy * a.Wout + ox) * COUT + co] = acc > 0.f ? acc : 0.f;
        }
    }
}

// ---- optional per-kernel-class timing with HIP events (vst_profile_begin / vst_profile_end) ---------
#define VST_PROFILE_MAX_RECORDS 4096
static int g_prof_kernel = 0;               // 0 = off, else VST_KERNEL_ID(cin, cout, stride)
static int g_prof_count = 0, g_prof_cap = 0;
static hipEvent_t g_prof_ev[2 * VST_PROFILE_MAX_RECORDS];
static bool g_prof_ev_created = false;

template <int CIN, int COUT, int STRIDE, bool IN_STATE, bool OUT_STATE>
static int launch_conv(const ConvArgs& a, int B, int precision, hipStream_t st) {
    if (precision == VST_PREC_FP32) {
        const size_t total = (size_t)B * a.Hout * a.Wout * COUT;
        size_t blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        conv_fp32_kernel<IN_STATE, OUT_STATE><<<dim3((unsigned)blocks), 256, 0, st>>>(a, CIN, COUT, STRIDE, B);
        VST_RETURN_IF_LAUNCH_FAILED();
        return VST_OK;
    }
    if (precision != VST_PREC_BF16X3) return VST_E_MODE;
    using C = ConvCfg<CIN, COUT, STRIDE>;
    auto kern = conv_mfma_kernel<CIN, COUT, STRIDE, IN_STATE, OUT_STATE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid((a.Wout + C::TW - 1) / C::TW, (a.Hout + C::TH - 1) / C::TH, B * C::NCOT);
    const bool timed = g_prof_kernel == VST_KERNEL_ID(CIN, COUT, STRIDE) && g_prof_count < g_prof_cap;
    if (timed) (void)hipEventRecord(g_prof_ev[2 * g_prof_count], st);
    kern<<<grid, 256, C::LDS_BYTES, st>>>(a);
    if (timed) { (void)hipEventRecord(g_prof_ev[2 * g_prof_count + 1], st); ++g_prof_count; }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

// one coupling block: dst (+/-)= F(src), three launches (h1, h2 are fp32 channels-last intermediates)
template <int CH, int STRIDE>
static int run_block(const vst_block_weights* w, int direction, int precision, float* dst, const float* src,
                     float* tmp, int B, int H, int W, hipStream_t st) {
    constexpr int LV = CH == 16 ? 0 : (CH == 64 ? 1 : 2);
    constexpr int MID = CH / 4;
    constexpr int IN_CH = STRIDE == 1 ? CH : CH / 4;
    const int Ho = H >> LV, Wo = W >> LV;
    const size_t state_img = (size_t)H * W * 16;
    const size_t mid_img = (size_t)Ho * Wo * MID;
    float* h1 = tmp;
    float* h2 = tmp + (size_t)B * mid_img;
    ConvArgs a{};
    a.Wq = W >> 2;
    // conv.1: state -> h1 (ReLU)
    a.in = src; a.out = h1; a.packed = (const unsigned char*)w->conv[0].packed; a.bias = w->conv[0].bias;
    a.Hin = Ho * STRIDE; a.Win = Wo * STRIDE; a.Hout = Ho; a.Wout = Wo;
    a.in_img_stride = state_img; a.out_img_stride = mid_img; a.sign = 0.f;
    int rc = launch_conv<IN_CH, MID, STRIDE, true, false>(a, B, precision, st);
    if (rc) return rc;
    // conv.4: h1 -> h2 (ReLU)
    a.in = h1; a.out = h2; a.packed = (const unsigned char*)w->conv[1].packed; a.bias = w->conv[1].bias;
    a.Hin = Ho; a.Win = Wo; a.in_img_stride = mid_img;
    rc = launch_conv<MID, MID, 1, false, false>(a, B, precision, st);
    if (rc) return rc;
    // conv.7: h2 -> dst += sign * (.)
    a.in = h2; a.out = dst; a.packed = (const unsigned char*)w->conv[2].packed; a.bias = w->conv[2].bias;
    a.out_img_stride = state_img; a.sign = direction > 0 ? 1.f : -1.f;
    return launch_conv<MID, CH, 1, false, true>(a, B, precision, st);
}

static const int kBlockChannel[VST_NUM_BLOCKS] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 64, 64, 64, 64, 64, 64,
                                                  64, 64, 64, 64, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256,
                                                  256, 256};
static const int kBlockStride[VST_NUM_BLOCKS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1,
                                                 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

extern "C" {

int vst_profile_begin(int kernel_id, int max_records) {
    if (kernel_id <= 0 || max_records <= 0) return VST_E_ARG;
    if (max_records > VST_PROFILE_MAX_RECORDS) max_records = VST_PROFILE_MAX_RECORDS;
    if (!g_prof_ev_created) {
        for (int i = 0; i < 2 * VST_PROFILE_MAX_RECORDS; ++i) {
            hipError_t e = hipEventCreate(&g_prof_ev[i]);
            if (e != hipSuccess) return (int)e;
        }
        g_prof_ev_created = true;
    }
    g_prof_kernel = kernel_id; g_prof_count = 0; g_prof_cap = max_records;
    return VST_OK;
}

int vst_profile_end(double* total_ms, int* launches) {
    if (!total_ms || !launches) return VST_E_ARG;
    double tot = 0.0;
    for (int i = 0; i < g_prof_count; ++i) {
        hipError_t e = hipEventSynchronize(g_prof_ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        tot += ms;
    }
    *total_ms = tot; *launches = g_prof_count;
    g_prof_kernel = 0; g_prof_count = 0; g_prof_cap = 0;
    return VST_OK;
}

size_t vst_block_tmp_bytes(int B, int H, int W) { return (size_t)B * H * W * 8 * sizeof(float); }

int vst_block_apply(const vst_block_weights* w, int channel, int stride, int direction, int precision,
                    float* dst, const float* src, void* tmp, int B, int H, int W, void* stream) {
    if (!w || !dst || !src || !tmp) return VST_E_ARG;
    if (!vst_shape_ok(B, H, W)) return VST_E_SHAPE;
    if (direction != 1 && direction != -1) return VST_E_MODE;
    for (int i = 0; i < 3; ++i)
        if (!w->conv[i].packed || !w->conv[i].bias) return VST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* t = (float*)tmp;
    if (channel == 16 && stride == 1) return run_block<16, 1>(w, direction, precision, dst, src, t, B, H, W, st);
    if (channel == 64 && stride == 1) return run_block<64, 1>(w, direction, precision, dst, src, t, B, H, W, st);
    if (channel == 64 && stride == 2) return run_block<64, 2>(w, direction, precision, dst, src, t, B, H, W, st);
    if (channel == 256 && stride == 1) return run_block<256, 1>(w, direction, precision, dst, src, t, B, H, W, st);
    if (channel == 256 && stride == 2) return run_block<256, 2>(w, direction, precision, dst, src, t, B, H, W, st);
    return VST_E_SHAPE;
}

size_t vst_pass_workspace_bytes(int B, int H, int W) { return (size_t)B * H * W * (16 + 16 + 8) * sizeof(float); }

int vst_revnet_forward(const vst_net_weights* w, const float* x, float* z, void* workspace, int B, int C_in, int H,
                       int W, int sp_steps, int precision, void* stream) {
    if (!w || !x || !z) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(B, H, W) || C_in < 1 || C_in > 16) return VST_E_SHAPE;
    if (sp_steps != 1 && sp_steps != 2) return VST_E_MODE;
    float* s[2];
    s[0] = (float*)workspace;
    s[1] = s[0] + (size_t)B * H * W * 16;
    float* tmp = s[1] + (size_t)B * H * W * 16;
    int rc = vst_pack_input(x, s[0], s[1], B, C_in, H, W, stream);
    if (rc) return rc;
    for (int k = 0; k < VST_NUM_BLOCKS; ++k) {
        rc = vst_block_apply(&w->blocks[k], kBlockChannel[k], kBlockStride[k], +1, precision, s[k & 1], s[1 - (k & 1)],
                             tmp, B, H, W, stream);
        if (rc) return rc;
    }
    return vst_spread(s[0], s[1], z, B, H, W, sp_steps, stream);
}

int vst_revnet_inverse(const vst_net_weights* w, const float* z, float* x, void* workspace, int B, int C_out, int H,
                       int W, int sp_steps, int precision, void* stream) {
    if (!w || !x || !z) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(B, H, W) || C_out < 1 || C_out > 16) return VST_E_SHAPE;
    if (sp_steps != 1 && sp_steps != 2) return VST_E_MODE;
    float* s[2];
    s[0] = (float*)workspace;
    s[1] = s[0] + (size_t)B * H * W * 16;
    float* tmp = s[1] + (size_t)B * H * W * 16;
    int rc = vst_gather(z, s[0], s[1], B, H, W, sp_steps, stream);
    if (rc) return rc;
    for (int k = VST_NUM_BLOCKS - 1; k >= 0; --k) {
        rc = vst_block_apply(&w->blocks[k], kBlockChannel[k], kBlockStride[k], -1, precision, s[k & 1], s[1 - (k & 1)],
                             tmp, B, H, W, stream);
        if (rc) return rc;
    }
    return vst_unpack_output(s[0], x, B, C_out, H, W, stream);
}

}  // extern "C"

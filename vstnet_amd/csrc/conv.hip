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
#include <mutex>
#include <stdlib.h>
#include "common.h"

#ifndef VST_ABLATE
#define VST_ABLATE 0
#endif

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
    int tiles_x, tiles_y, tiles_total;   // 1-D grid of round_up(tiles_total, 8) workgroups, see xcd_tile()
    const unsigned char* packed1;        // conv_pair_kernel: conv.4's packed weights and bias (in = h1)
    const float* bias1;
    unsigned char* out_sp;               // OUT_SP: the output goes to split fp16 planes (common.h) instead of `out`
    size_t out_sp_img_bytes;
    unsigned trace_base;                 // -DVST_TRACE=3 builds: first record of this launch in the trace buffer
};

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Give XCD k the contiguous
// (row-major) range of tiles [k*per, (k+1)*per) so that tiles sharing a halo also share an L2.
#ifndef VST_XCD_REMAP
#define VST_XCD_REMAP 1
#endif
#ifndef VST_PIPE_OLD_SPREAD
#define VST_PIPE_OLD_SPREAD 1        // conv_pipe_kernel (multi-slice form): old state values fetched one unit per k-step (see there)
#endif
#ifndef VST_PIPE_NO_DEFER
#define VST_PIPE_NO_DEFER 0          // 1: conv_pipe_kernel stores every slice at its end (the form before round 3; A/B builds)
#endif
#ifdef VST_TRACE
// Diagnostic builds only (-DVST_TRACE=1: conv_pair_kernel, 2: conv_mfma_kernel): every workgroup of the LAST traced launch
// leaves {start, end (100 MHz ticks), HW_ID, XCC_ID} here; tools/trace_grid.py reads them back through vst_trace_dump.
// -DVST_TRACE=3: EVERY workgroup of EVERY conv launch (classes 1, 2, 4 = conv_pipe_kernel) appends
// {start, end, HW_ID | XCC_ID << 32, class | Cin << 8 | Cout << 20 | blockIdx << 32} to a caller-provided buffer
// (vst_trace_set; tools/trace_cu.py): which kernels of which streams were resident on which CU, when.
__device__ unsigned long long vst_trace_buf[4 * 16384];
__device__ unsigned long long* vst_trace_ptr;
__device__ unsigned vst_trace_cap;
static std::atomic<unsigned> vst_trace_host_n{0};     // next free record (launch sites reserve one per workgroup)
#define VST_TRACE_RESERVE(t_, grid_) (t_).trace_base = vst_trace_host_n.fetch_add((unsigned)(grid_));
#define VST_TRACE_BEGIN(which)                                                                                  \
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();   /* scalar: no VGPR is held for it */
#define VST_TRACE_END_(which, cin_, cout_)                                                                      \
    if (VST_TRACE == (which) || VST_TRACE == 3) {                                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
        if (threadIdx.x == 0 && (VST_TRACE == 3 || blockIdx.x < 16384)) {                                       \
            unsigned hw, xcc;                                                                                   \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                    \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                  \
            if (VST_TRACE == 3) {                                                                               \
                const unsigned i_ = a.trace_base + blockIdx.x;                                                  \
                if (vst_trace_ptr && i_ < vst_trace_cap) {                                                      \
                    unsigned long long* r_ = vst_trace_ptr + 4 * (size_t)i_;                                    \
                    r_[0] = trace_t0;                                                                           \
                    r_[1] = __builtin_amdgcn_s_memrealtime();                                                   \
                    r_[2] = hw | ((unsigned long long)xcc << 32);                                               \
                    r_[3] = (which) | ((cin_) << 8) | ((cout_) << 20) | ((unsigned long long)blockIdx.x << 32); \
                }                                                                                               \
            } else {                                                                                            \
                vst_trace_buf[4 * blockIdx.x + 0] = trace_t0;                                                   \
                vst_trace_buf[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();                           \
                vst_trace_buf[4 * blockIdx.x + 2] = hw;                                                         \
                vst_trace_buf[4 * blockIdx.x + 3] = xcc;                                                        \
            }                                                                                                   \
        }                                                                                                       \
    }
#define VST_TRACE_END(which) VST_TRACE_END_(which, 0, 0)
extern "C" __attribute__((visibility("default"))) int vst_trace_dump(unsigned long long* host, int n_wg) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vst_trace_buf), sizeof(unsigned long long) * 4 * (size_t)n_wg);
}
// append mode: records go to `dev_buf` (cap records of 4 x u64); returns the number of records written so far through *n
extern "C" __attribute__((visibility("default"))) int vst_trace_set(unsigned long long* dev_buf, unsigned cap) {
    const unsigned zero = 0;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(vst_trace_ptr), &dev_buf, sizeof(dev_buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(vst_trace_cap), &cap, sizeof(cap));
    (void)zero;
    vst_trace_host_n.store(0);
    return (int)e;
}
extern "C" __attribute__((visibility("default"))) int vst_trace_count(unsigned* n) { *n = vst_trace_host_n.load(); return 0; }
#else
#define VST_TRACE_BEGIN(which)
#define VST_TRACE_END(which)
#define VST_TRACE_END_(which, cin_, cout_)
#define VST_TRACE_RESERVE(t_, grid_)
#endif

__device__ __forceinline__ bool xcd_tile(const ConvArgs& a, int& bx, int& by, int& bz) {
    const int g = blockIdx.x, per = gridDim.x >> 3;
    const int n = VST_XCD_REMAP ? (g & 7) * per + (g >> 3) : g;
    if (n >= a.tiles_total) return false;
    bx = n % a.tiles_x;
    const int r = n / a.tiles_x;
    by = r % a.tiles_y;
    bz = r / a.tiles_y;
    return true;
}

template <int CIN, int COUT, int STRIDE, int MRO = 0>      // MRO: tile rows per wave if not the default
struct ConvCfg {
    static constexpr int NT = COUT >= 64 ? 64 : 16;      // output channels per workgroup
    static constexpr int NB = NT / 16;                   // 16-wide N blocks per wave
    static constexpr int COUTP = (COUT + 15) / 16 * 16;
    static constexpr int NCOT = COUTP / NT;              // co tiles (grid.z factor)
    static constexpr int MR = MRO ? MRO : (STRIDE == 2 ? 2 : 4);   // tile rows (16-pixel M blocks) per wave
    static constexpr int TH = 4 * MR, TW = 16;
    static constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
    static constexpr int NSLOT = (IH * IW + 15) / 16 * 16;
    static constexpr int CC = CIN >= 32 ? 32 : CIN;      // input channels staged per chunk
    static constexpr int NCHUNK = CIN / CC;
    static constexpr int CIG = CC >= 8 ? CC / 8 : 1;     // 8-channel groups per chunk
#ifndef VST_ABLATE_KS16
#define VST_ABLATE_KS16 5            // (timing-only builds: fewer k-steps for the 16-channel inputs = what folding the horizontal tap into N would issue)
#endif
    static constexpr int KS = CIN >= 32 ? 9 : (CIN == 16 ? VST_ABLATE_KS16 : 2);   // 32-deep k steps per chunk
    static constexpr int A_PLANE = CIN == 4 ? NSLOT * 8 : CIG * NSLOT * 16;
    static constexpr int B_PLANE = KS * 4 * NT * 16;
    static constexpr int LDS_BYTES = 2 * A_PLANE + 2 * B_PLANE;
    static constexpr int LDS_BYTES_T2 = 2 * A_PLANE + B_PLANE;     // TERMS == 2: one weight plane
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// TERMS == 3: bf16 hi / lo (3-term product with split weights); TERMS == 2: fp16 hi / lo (2-term product w_hi (x_hi + x_lo),
// the arithmetic of conv3.hip; values saturate at +-65504 in the hi part)
__device__ __forceinline__ void split8_f16(const float4 v0, const float4 v1, uint4& hi, uint4& lo, float& amax) {
    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    u32x4 h, l;
    split8_sp(f, h, l, amax);
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
}

__device__ __forceinline__ void split4_f16(const float4 v, uint2& hi, uint2& lo, float& amax) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    f16x4 h, l;
#ifndef VST_NO_RANGE_CHECK
    amax = fmaxf(fmaxf(amax, fabsf(f[0])), fmaxf(fmaxf(fabsf(f[1]), fabsf(f[2])), fabsf(f[3])));
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float c = __builtin_amdgcn_fmed3f(f[i], -65504.f, 65504.f);
        h[i] = (_Float16)c;
        l[i] = (_Float16)(f[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

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

// ---- register epilogue ---------------------------------------------------------------------------
// The MFMAs are issued as D = W * X^T (weights are the A operand, pixels the B operand), so lane
// (lrow, kg) ends up with D[co = 4*kg + r][pixel = lrow]: four CONSECUTIVE channels of one pixel per
// accumulator -> one float4 store (or float4 read-modify-write of the state) per 16x16 tile, no LDS.
template <int COUT, int NB>
__device__ __forceinline__ void load_bias(const ConvArgs& a, int co_lane, float4 (&bias)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
        bias[n] = co_lane + n * 16 < COUT ? *(const float4*)(a.bias + co_lane + n * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// address of the 4 channels [co_lane + 16 n, +4) of output pixel (oy, ox); nullptr when outside the image / channel range
template <int COUT, bool OUT_STATE, bool FULL = false>
__device__ __forceinline__ float4* out_ptr(const ConvArgs& a, float* out_img, int oy, int ox, int co) {
    if (!FULL && (oy >= a.Hout || ox >= a.Wout || co >= COUT)) return nullptr;
    if (OUT_STATE) return (float4*)(out_img + zc_offset(vst_level_of_channels(COUT), oy, ox, a.Wq) + co);
    return (float4*)(out_img + ((size_t)oy * a.Wout + ox) * COUT + co);
}

template <int COUT, int MR, int NB, bool FULL = false>
__device__ __forceinline__ void load_old(const ConvArgs& a, float* out_img, int oy0, int ox, int co_lane,
                                         float4 (&old)[MR][NB]) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const float4* p = out_ptr<COUT, true, FULL>(a, out_img, oy0 + m, ox, co_lane + n * 16);
            old[m][n] = (FULL || p) ? *p : make_float4(0.f, 0.f, 0.f, 0.f);
        }
}

// OUT_STATE: out = old + sign * (acc + bias);  else: out = relu(acc + bias)
template <int COUT, bool OUT_STATE, int MR, int NB, bool FULL = false>
__device__ __forceinline__ void store_tile(const ConvArgs& a, float* out_img, int oy0, int ox, int co_lane,
                                           const f32x4 (&acc)[MR][NB], const float4 (&bias)[NB],
                                           const float4 (&old)[MR][NB]) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            float4* p = out_ptr<COUT, OUT_STATE, FULL>(a, out_img, oy0 + m, ox, co_lane + n * 16);
            if (!FULL && !p) continue;
            float4 r = make_float4(acc[m][n][0] + bias[n].x, acc[m][n][1] + bias[n].y, acc[m][n][2] + bias[n].z,
                                   acc[m][n][3] + bias[n].w);
            if (OUT_STATE) {
                const float4 o = old[m][n];
                r = make_float4(o.x + a.sign * r.x, o.y + a.sign * r.y, o.z + a.sign * r.z, o.w + a.sign * r.w);
            } else {
                r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f;
                r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
            }
            *p = r;
        }
}

#define MFMA3(acc, wh, wl, xh, xl)                                              \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0)

// the same with the number of terms chosen at compile time: operands are 16-byte containers (bf16x8), read as fp16 when TERMS == 2
#define MFMA_T(acc, wh, wl, xh, xl)                                                                                          \
    if constexpr (TERMS == 3) {                                                                                               \
        MFMA3(acc, wh, wl, xh, xl);                                                                                           \
    } else {                                                                                                                  \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xl), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xh), acc, 0, 0, 0); \
    }
// (range_amax: the kernel's running max |x| of what it rounds to fp16, handed to vst_note_range once at its end)
#define SPLIT8_T(v0_, v1_, h_, l_) { if constexpr (TERMS == 3) split8(v0_, v1_, h_, l_); else split8_f16(v0_, v1_, h_, l_, range_amax); }
#define SPLIT4_T(v_, h_, l_) { if constexpr (TERMS == 3) split4(v_, h_, l_); else split4_f16(v_, h_, l_, range_amax); }

#ifndef VST_EARLY_OLD
#define VST_EARLY_OLD 1
#endif
// ---- generic kernel: one tile per workgroup, staging per input-channel chunk (all shapes) --------------
// OUT_H16 (an intermediate h1, not the state): ReLU(acc + bias) is written as fp16, channels-last like the fp32 form
// (VST_PREC_F16X2H: the pair kernel reads h1 as the fp16 operand it is, one MFMA per product in conv.4)
template <int CIN, int COUT, int STRIDE, bool IN_STATE, bool OUT_STATE, bool OUT_SP = false, int TERMS = 3, bool OUT_H16 = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    using C = ConvCfg<CIN, COUT, STRIDE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const a_hi = smem;
    unsigned char* const a_lo = smem + C::A_PLANE;
    unsigned char* const b_hi = smem + 2 * C::A_PLANE;
    unsigned char* const b_lo = TERMS == 3 ? b_hi + C::B_PLANE : b_hi;     // TERMS == 2: one weight plane

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    int bx, by, bz;
    if (!xcd_tile(a, bx, by, bz)) return;
    float range_amax = 0.f;
    VST_TRACE_BEGIN(2)
    const int tx0 = bx * C::TW, ty0 = by * C::TH;
    const int b = bz / C::NCOT, co0 = (bz % C::NCOT) * C::NT;

    const float* const in_img = a.in + (size_t)b * a.in_img_stride;
    const PackedConvLayout PL = packed_conv_layout(COUT, CIN);
    const unsigned char* const w_hi = a.packed + (TERMS == 3 ? PL.f32_bytes : PL.f16_offset);
    const unsigned char* const w_lo = w_hi + PL.frag_bytes;                 // (not read when TERMS == 2)

    f32x4 acc[C::MR][C::NB];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < C::NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane slot of tile row (wave*MR), pixel lrow, tap (0,0)
    const int slot_base = (wave * C::MR * STRIDE) * C::IW + lrow * STRIDE;

    // Staging is split into "fetch" (global -> registers) and "land" (bf16 split, registers -> LDS) so that the
    // latencies overlap: weights and activations of a chunk are fetched together, the next chunk is fetched before this
    // chunk's MFMAs, and the read-modify-write state values are fetched before the last chunk's MFMAs wherever they
    // fit in registers that are dead until the epilogue (s_memtime stamps of the synchronous form showed the weight
    // fetch, the second chunk's fetch and the old-state fetch as 2-2.5 us waits each in a 13 us workgroup).
    constexpr int A_ITEMS_T = CIN == 4 ? C::IH * C::IW : C::IH * C::IW * C::CIG;
    constexpr int AIT = (A_ITEMS_T + 255) / 256, NV = CIN == 4 ? 1 : 2;
    constexpr int W_ITEMS_T = C::KS * 4 * C::NT, WIT = (W_ITEMS_T + 255) / 256;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // native vectors: these arrays must stay in VGPRs
    f32x4 areg[AIT][NV];
    u32x4 wreg[WIT][2];
#define F4(v_) make_float4((v_)[0], (v_)[1], (v_)[2], (v_)[3])
#define GEN_FETCH(chunk_)                                                                           \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                           \
        int idx = it * 256 + tid;                                                                  \
        idx = idx < W_ITEMS_T ? idx : W_ITEMS_T - 1;                                               \
        const int co = idx % C::NT, r = idx / C::NT;                                               \
        const size_t src = ((size_t)((chunk_) * C::KS * 4 + r) * C::COUTP + co0 + co) * 16;        \
        wreg[it][0] = *(const u32x4*)(w_hi + src);                                                 \
        if (TERMS == 3) wreg[it][1] = *(const u32x4*)(w_lo + src);                                 \
    }                                                                                              \
    _Pragma("unroll") for (int it = 0; it < AIT; ++it) {                                           \
        int idx = it * 256 + tid;                                                                  \
        idx = idx < A_ITEMS_T ? idx : A_ITEMS_T - 1;                                               \
        const int cig = CIN == 4 ? 0 : idx % C::CIG, slot = CIN == 4 ? idx : idx / C::CIG;         \
        const int iy = slot / C::IW, ix = slot - iy * C::IW;                                       \
        const int gy = reflect_clamp(ty0 * STRIDE - 1 + iy, a.Hin);                                \
        const int gx = reflect_clamp(tx0 * STRIDE - 1 + ix, a.Win);                                \
        const size_t off = IN_STATE ? zc_offset(vst_level_of_channels(CIN), gy, gx, a.Wq)          \
                                    : ((size_t)gy * a.Win + gx) * CIN;                             \
        const float* p = in_img + off + (chunk_) * C::CC + cig * 8;                                \
        areg[it][0] = *(const f32x4*)p;                                                            \
        if (NV == 2) areg[it][NV - 1] = *(const f32x4*)(p + 4);                                    \
    }
    // threads past the end repeat the last item (same value to the same address): no predicates, so the compiler keeps
    // every fetch where it was issued
#define GEN_LAND()                                                                                 \
    _Pragma("unroll") for (int it = 0; it < AIT; ++it) {                                           \
        int idx = it * 256 + tid;                                                                  \
        idx = idx < A_ITEMS_T ? idx : A_ITEMS_T - 1;                                               \
        if (CIN == 4) {                                                                            \
            uint2 h, l;                                                                            \
            SPLIT4_T(F4(areg[it][0]), h, l);                                                       \
            *(uint2*)(a_hi + idx * 8) = h;                                                         \
            *(uint2*)(a_lo + idx * 8) = l;                                                         \
        } else {                                                                                   \
            const int cig = idx % C::CIG, slot = idx / C::CIG;                                     \
            uint4 h, l;                                                                            \
            SPLIT8_T(F4(areg[it][0]), F4(areg[it][NV - 1]), h, l);                                 \
            *(uint4*)(a_hi + (cig * C::NSLOT + slot) * 16) = h;                                    \
            *(uint4*)(a_lo + (cig * C::NSLOT + slot) * 16) = l;                                    \
        }                                                                                          \
    }                                                                                              \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                           \
        int idx = it * 256 + tid;                                                                  \
        idx = idx < W_ITEMS_T ? idx : W_ITEMS_T - 1;                                               \
        *(u32x4*)(b_hi + idx * 16) = wreg[it][0];                                                  \
        if (TERMS == 3) *(u32x4*)(b_lo + idx * 16) = wreg[it][1];                                  \
    }

    float* const out_img = a.out + (size_t)b * a.out_img_stride;
    float4 bias[C::NB], old[C::MR][C::NB];
    const bool interior = COUT % 16 == 0 && ty0 + C::TH <= a.Hout && tx0 + C::TW <= a.Wout;   // no predicates needed
    // fetch the old state before the MFMAs only where its registers are free anyway (big register tiles: the 16-value
    // tiles of the 4->16 conv lose more to occupancy than they gain)
    constexpr bool EARLY_OLD = OUT_STATE && VST_EARLY_OLD && C::MR * C::NB >= 16;
    auto fetch_old = [&]() __attribute__((always_inline)) {
        if (interior) load_old<COUT, C::MR, C::NB, true>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, co0 + 4 * kg, old);
        else load_old<COUT, C::MR, C::NB>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, co0 + 4 * kg, old);
    };

    GEN_FETCH(0);
    GEN_LAND();
#pragma unroll
    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
        __syncthreads();
        if (chunk + 1 < C::NCHUNK) { GEN_FETCH(chunk + 1); }
        if (EARLY_OLD && chunk == C::NCHUNK - 1) fetch_old();

        // ---- MFMA over the chunk's k steps ---------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            bf16x8 wh[C::NB], wl[C::NB];
#pragma unroll
            for (int n = 0; n < C::NB; ++n) {
                const int boff = ((ks * 4 + kg) * C::NT + n * 16 + lrow) * 16;
                wh[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(b_hi + boff));
                if (TERMS == 3) wl[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(b_lo + boff)); else wl[n] = wh[n];
            }
#pragma unroll
            for (int m = 0; m < C::MR; ++m) {
                bf16x8 xh, xl;
                if (CIN >= 32) {
                    const int dy = ks / 3, dx = ks - dy * 3;
                    const int slot = slot_base + (m * STRIDE + dy) * C::IW + dx;
                    const int aoff = (kg * C::NSLOT + slot) * 16;
                    xh = __builtin_bit_cast(bf16x8, *(const uint4*)(a_hi + aoff));
                    xl = __builtin_bit_cast(bf16x8, *(const uint4*)(a_lo + aoff));
                } else if (CIN == 16) {
                    int tap = 2 * ks + (kg >> 1);
                    tap = tap > 8 ? 8 : tap;
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const int slot = slot_base + (m * STRIDE + dy) * C::IW + dx;
                    const int aoff = ((kg & 1) * C::NSLOT + slot) * 16;
                    xh = __builtin_bit_cast(bf16x8, *(const uint4*)(a_hi + aoff));
                    xl = __builtin_bit_cast(bf16x8, *(const uint4*)(a_lo + aoff));
                } else {
                    int t0 = 8 * ks + 2 * kg;
                    t0 = t0 > 8 ? 8 : t0;
                    const int t1 = t0 + 1 > 8 ? 8 : t0 + 1;
                    const int s0 = slot_base + (m * STRIDE + t0 / 3) * C::IW + t0 % 3;
                    const int s1 = slot_base + (m * STRIDE + t1 / 3) * C::IW + t1 % 3;
                    const uint2 h0 = *(const uint2*)(a_hi + s0 * 8), h1 = *(const uint2*)(a_hi + s1 * 8);
                    const uint2 l0 = *(const uint2*)(a_lo + s0 * 8), l1 = *(const uint2*)(a_lo + s1 * 8);
                    xh = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                    xl = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                }
#pragma unroll
                for (int n = 0; n < C::NB; ++n) { MFMA_T(acc[m][n], wh[n], wl[n], xh, xl); }
            }
        }
        if (chunk + 1 < C::NCHUNK) {
            __syncthreads();
            GEN_LAND();
        }
    }
#undef GEN_FETCH
#undef GEN_LAND
#undef F4

    load_bias<COUT, C::NB>(a, co0 + 4 * kg, bias);
    if constexpr (OUT_SP) {
        // ReLU(acc + bias) as split fp16 planes for conv3.hip's kernels: the lane's 8 channels {16(2j) + 4kg + r, 16(2j+1) + 4kg + r}
        static_assert(COUT == 64 && C::NB == 4 && !OUT_STATE, "plane output: the 64-channel intermediates only");
        unsigned char* const sp_img = a.out_sp + (size_t)b * a.out_sp_img_bytes;
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
            const int oy = ty0 + wave * C::MR + m, ox = tx0 + lrow;
            if (oy >= a.Hout || ox >= a.Wout) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float f8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int n = 2 * j + (e >> 2);
                    const float bb = e & 2 ? (e & 1 ? bias[n].w : bias[n].z) : (e & 1 ? bias[n].y : bias[n].x);
                    const float v = acc[m][n][e & 3] + bb;
                    f8[e] = v > 0.f ? v : 0.f;
                }
                u32x4 hi, lo;
                split8_sp(f8, hi, lo, range_amax);
                *(u32x4*)(sp_img + sp_offset(j * 4 + kg, 0, oy, ox, a.Hout, a.Wout)) = hi;
                *(u32x4*)(sp_img + sp_offset(j * 4 + kg, 1, oy, ox, a.Hout, a.Wout)) = lo;
            }
        }
        vst_note_range(range_amax);
        return;
    }
    if constexpr (OUT_H16) {
        static_assert(!OUT_STATE && !OUT_SP, "fp16 output: the h1 intermediates only");
        _Float16* const o16 = (_Float16*)a.out + (size_t)b * a.out_img_stride;
#pragma unroll
        for (int m = 0; m < C::MR; ++m)
#pragma unroll
            for (int n = 0; n < C::NB; ++n) {
                const int oy = ty0 + wave * C::MR + m, ox = tx0 + lrow, co = co0 + 4 * kg + n * 16;
                if (oy >= a.Hout || ox >= a.Wout || co >= COUT) continue;
                const float r[4] = {acc[m][n][0] + bias[n].x, acc[m][n][1] + bias[n].y, acc[m][n][2] + bias[n].z, acc[m][n][3] + bias[n].w};
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                f16x4 h;
                range_amax = fmaxf(fmaxf(range_amax, r[0]), fmaxf(fmaxf(r[1], r[2]), r[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (_Float16)__builtin_amdgcn_fmed3f(r[e], 0.f, 65504.f);     // ReLU, saturating
                *(f16x4*)(o16 + ((size_t)oy * a.Wout + ox) * COUT + co) = h;
            }
        vst_note_range(range_amax);
        VST_TRACE_END_(2, CIN, COUT)
        return;
    }
    if constexpr (TERMS == 2) vst_note_range(range_amax);
    if (OUT_STATE && !EARLY_OLD) fetch_old();
    if (interior) store_tile<COUT, OUT_STATE, C::MR, C::NB, true>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, co0 + 4 * kg, acc, bias, old);
    else store_tile<COUT, OUT_STATE, C::MR, C::NB>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, co0 + 4 * kg, acc, bias, old);
    VST_TRACE_END_(2, CIN, COUT)
}

// ---- conv.4 + conv.7 of one stage-1 / stage-2 coupling block in one launch -----------------------------------------
// h2 = ReLU(conv.4(h1)) never goes to HBM: the workgroup computes it for its 16x16 tile plus a one-pixel ring
// (18x18 positions, from a 20x20 region of h1) into the LDS image that conv.7's MFMAs read, then does
// dst += sign * (conv.7(h2) + bias).  ReflectionPad of h2 (RevResNet.py:85): an in-image pixel one step inside the
// border also writes the mirrored ring slot.  LDS: [h2 image][ union { h1 region + conv.4 weights ; conv.7 weights } ].
template <int CIN, bool NO_LO = false>
__device__ __forceinline__ void read_x_small(const unsigned char* img_hi, const unsigned char* img_lo, int nslot,
                                             int slot00, int iw, int ks, int kg, bf16x8& xh, bf16x8& xl) {
    if (CIN == 16) {
        int tap = 2 * ks + (kg >> 1);
        tap = tap > 8 ? 8 : tap;
        const int slot = slot00 + (tap / 3) * iw + tap % 3;
        const int aoff = ((kg & 1) * nslot + slot) * 16;
        xh = __builtin_bit_cast(bf16x8, *(const uint4*)(img_hi + aoff));
        if (!NO_LO) xl = __builtin_bit_cast(bf16x8, *(const uint4*)(img_lo + aoff)); else xl = xh;
    } else {
        int t0 = 8 * ks + 2 * kg;
        t0 = t0 > 8 ? 8 : t0;
        const int t1 = t0 + 1 > 8 ? 8 : t0 + 1;
        const int s0 = slot00 + (t0 / 3) * iw + t0 % 3, s1 = slot00 + (t1 / 3) * iw + t1 % 3;
        const uint2 h0 = *(const uint2*)(img_hi + s0 * 8), h1 = *(const uint2*)(img_hi + s1 * 8);
        xh = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        if (!NO_LO) {
            const uint2 l0 = *(const uint2*)(img_lo + s0 * 8), l1 = *(const uint2*)(img_lo + s1 * 8);
            xl = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
        } else {
            xl = xh;
        }
    }
}

template <int MID, int CH, int TERMS = 3, int MR = 4, bool H16 = false>
struct PairCfg {
    using C7 = ConvCfg<MID, CH, 1, MR>;
    // h1 region (TH + 4 rows x 20 columns), h2 ring region (TH + 2 rows x 18 columns) of a TH x 16 tile, TH = 4 MR
    static constexpr int R1 = 20, N1SLOT = (4 * MR + 4) * R1, RW = 18, RH = 4 * MR + 2, NPX = RH * RW, NBLK = (NPX + 15) / 16;
    static constexpr int H1_PLANE = MID == 4 ? N1SLOT * 8 : C7::CIG * N1SLOT * 16;
    static constexpr int W4_PLANE = C7::KS * 4 * 16 * 16;
    // MID == 16: conv.7's 40 KB of weights take over the h1 region + conv.4 weights once conv.4 is done (62 KB, two
    // workgroups per CU); MID == 4: everything is small, conv.7's weights get their own region (one barrier fewer)
    static constexpr bool ALIAS = MID == 16;
    static constexpr int WPL = TERMS == 3 ? 2 : 1;            // weight planes in LDS (hi + lo, or the fp16 plane alone)
    static constexpr int H1W4 = (H16 ? 1 : 2) * H1_PLANE + WPL * W4_PLANE;   // H16: h1 arrives as fp16, one plane
    static constexpr int U_BYTES = ALIAS ? (H1W4 > WPL * C7::B_PLANE ? H1W4 : WPL * C7::B_PLANE) : H1W4 + WPL * C7::B_PLANE;
    static constexpr int LDS_BYTES = 2 * C7::A_PLANE + U_BYTES;
};

// H16: h1 is fp16 in HBM (conv_mfma_kernel<..., OUT_H16>): it is copied into the LDS image as it is and conv.4 issues one
// MFMA per product
template <int MID, int CH, int TERMS = 3, int MR = 4, bool H16 = false>
__global__ __launch_bounds__(256, (MID == 16 && TERMS == 2) ? 3 : 1) void conv_pair_kernel(const ConvArgs a) {
    static_assert(!H16 || TERMS == 2, "fp16 h1: the 2-term kernels only");
    using P = PairCfg<MID, CH, TERMS, MR, H16>;
    using C = typename P::C7;
    static_assert(C::NCHUNK == 1 && C::NCOT == 1, "single-chunk shapes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const a_hi = smem;                        // h2 image (conv.7's activation image)
    unsigned char* const a_lo = smem + C::A_PLANE;
    unsigned char* const u0 = smem + 2 * C::A_PLANE;
    unsigned char* const b_hi = P::ALIAS ? u0 : u0 + P::H1W4;   // conv.7 weights (aliased: valid after conv.4 is done)
    unsigned char* const b_lo = TERMS == 3 ? b_hi + C::B_PLANE : b_hi;
    unsigned char* const h1_hi = u0;                         // h1 region + conv.4 weights (before)
    unsigned char* const h1_lo = H16 ? u0 : u0 + P::H1_PLANE;           // (H16: no lo plane)
    unsigned char* const w4s_hi = u0 + (H16 ? 1 : 2) * P::H1_PLANE;
    unsigned char* const w4s_lo = TERMS == 3 ? w4s_hi + P::W4_PLANE : w4s_hi;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    int bx, by, b;
    if (!xcd_tile(a, bx, by, b)) return;
    float range_amax = 0.f;
    VST_TRACE_BEGIN(1)
    const int tx0 = bx * C::TW, ty0 = by * C::TH;
    const int H = a.Hout, W = a.Wout;                        // h1, h2 and the output view share one resolution
    const float* const in_img = a.in + (size_t)b * a.in_img_stride;
    float* const out_img = a.out + (size_t)b * a.out_img_stride;
    const PackedConvLayout PL7 = packed_conv_layout(CH, MID), PL4 = packed_conv_layout(MID, MID);
    const unsigned char* const w7_hi = a.packed + (TERMS == 3 ? PL7.f32_bytes : PL7.f16_offset);
    const unsigned char* const w7_lo = w7_hi + PL7.frag_bytes;              // (the lo planes are not read when TERMS == 2)
    const unsigned char* const w4_hi = a.packed1 + (TERMS == 3 ? PL4.f32_bytes : PL4.f16_offset);
    const unsigned char* const w4_lo = w4_hi + PL4.frag_bytes;

    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr int H1_ITEMS = MID == 4 ? P::N1SLOT : P::N1SLOT * C::CIG, H1IT = (H1_ITEMS + 255) / 256, NV = MID == 4 ? 1 : 2;
    constexpr int W4_ITEMS = C::KS * 4 * 16, W4IT = (W4_ITEMS + 255) / 256;
    constexpr int W7_ITEMS = C::KS * 4 * C::NT, W7IT = (W7_ITEMS + 255) / 256;
    f32x4 hreg[H1IT][NV];
    u32x4 h16reg[H16 ? H1IT : 1];                            // (H16: raw fp16 words - never routed through float registers)
    u32x4 w4reg[W4IT][2], w7reg[W7IT][2];
    // ---- fetch everything this workgroup reads, back to back ---------------------------------------------------------
#pragma unroll
    for (int it = 0; it < W4IT; ++it) {
        int idx = it * 256 + tid;
        idx = idx < W4_ITEMS ? idx : W4_ITEMS - 1;
        w4reg[it][0] = *(const u32x4*)(w4_hi + (size_t)idx * 16);      // coutp == 16: fragment order is already [r][co]
        if (TERMS == 3) w4reg[it][1] = *(const u32x4*)(w4_lo + (size_t)idx * 16);
    }
#pragma unroll
    for (int it = 0; it < H1IT; ++it) {
        int idx = it * 256 + tid;
        idx = idx < H1_ITEMS ? idx : H1_ITEMS - 1;
        const int cig = MID == 4 ? 0 : idx % C::CIG, slot = MID == 4 ? idx : idx / C::CIG;
        const int iy = slot / P::R1, ix = slot - iy * P::R1;
        const int gy = reflect_clamp(ty0 - 2 + iy, H), gx = reflect_clamp(tx0 - 2 + ix, W);
        if constexpr (H16) {                                   // 8 (MID == 16) or 4 (MID == 4) fp16 values: the operand itself
            const _Float16* p16 = (const _Float16*)a.in + (size_t)b * a.in_img_stride + ((size_t)gy * W + gx) * MID + cig * 8;
            if (MID == 4) { const uint2 v = *(const uint2*)p16; h16reg[it] = u32x4{v.x, v.y, 0u, 0u}; }
            else h16reg[it] = *(const u32x4*)p16;
        } else {
            const float* p = in_img + ((size_t)gy * W + gx) * MID + cig * 8;
            hreg[it][0] = *(const f32x4*)p;
            if (NV == 2) hreg[it][NV - 1] = *(const f32x4*)(p + 4);
        }
    }
#pragma unroll
    for (int it = 0; it < W7IT; ++it) {
        int idx = it * 256 + tid;
        idx = idx < W7_ITEMS ? idx : W7_ITEMS - 1;
        const int co = idx % C::NT, r = idx / C::NT;
        const size_t src = ((size_t)r * C::COUTP + co) * 16;
        w7reg[it][0] = *(const u32x4*)(w7_hi + src);
        if (TERMS == 3) w7reg[it][1] = *(const u32x4*)(w7_lo + src);
    }
    float4 bias[C::NB], old[C::MR][C::NB];
    const bool interior = ty0 + C::TH <= H && tx0 + C::TW <= W;
    constexpr bool EARLY_OLD = C::MR * C::NB >= 16 && TERMS == 3;   // (2-term form: three workgroups per CU need <= 168 VGPRs)
#define PAIR_FETCH_OLD()                                                                                         \
    if (interior) load_old<CH, C::MR, C::NB, true>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, 4 * kg, old);   \
    else load_old<CH, C::MR, C::NB>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, 4 * kg, old);
    if (EARLY_OLD) { PAIR_FETCH_OLD(); }

    // ---- land h1 (bf16 split) and conv.4's weights --------------------------------------------------------------------
#define F4(v_) make_float4((v_)[0], (v_)[1], (v_)[2], (v_)[3])
#pragma unroll
    for (int it = 0; it < H1IT; ++it) {
        int idx = it * 256 + tid;
        idx = idx < H1_ITEMS ? idx : H1_ITEMS - 1;
        if constexpr (H16) {
            if (MID == 4) {
                *(uint2*)(h1_hi + idx * 8) = make_uint2(h16reg[it][0], h16reg[it][1]);
            } else {
                const int cig = idx % C::CIG, slot = idx / C::CIG;
                *(u32x4*)(h1_hi + (cig * P::N1SLOT + slot) * 16) = h16reg[it];
            }
        } else if (MID == 4) {
            uint2 h, l;
            SPLIT4_T(F4(hreg[it][0]), h, l);
            *(uint2*)(h1_hi + idx * 8) = h;
            *(uint2*)(h1_lo + idx * 8) = l;
        } else {
            const int cig = idx % C::CIG, slot = idx / C::CIG;
            uint4 h, l;
            SPLIT8_T(F4(hreg[it][0]), F4(hreg[it][NV - 1]), h, l);
            *(uint4*)(h1_hi + (cig * P::N1SLOT + slot) * 16) = h;
            *(uint4*)(h1_lo + (cig * P::N1SLOT + slot) * 16) = l;
        }
    }
#undef F4
#pragma unroll
    for (int it = 0; it < W4IT; ++it) {
        int idx = it * 256 + tid;
        idx = idx < W4_ITEMS ? idx : W4_ITEMS - 1;
        *(u32x4*)(w4s_hi + idx * 16) = w4reg[it][0];
        if (TERMS == 3) *(u32x4*)(w4s_lo + idx * 16) = w4reg[it][1];
    }
#define LAND_W7()                                                                                  \
    _Pragma("unroll") for (int it = 0; it < W7IT; ++it) {                                          \
        int idx = it * 256 + tid;                                                                  \
        idx = idx < W7_ITEMS ? idx : W7_ITEMS - 1;                                                 \
        *(u32x4*)(b_hi + idx * 16) = w7reg[it][0];                                                 \
        if (TERMS == 3) *(u32x4*)(b_lo + idx * 16) = w7reg[it][1];                                 \
    }
    if (!P::ALIAS) { LAND_W7(); }
    __syncthreads();

    // ---- conv.4 on the 18x18 ring region: 16-pixel blocks of the linearised region, one 16-channel block ---------------
    {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * kg < MID) b4 = *(const float4*)(a.bias1 + 4 * kg);
        const bool ring_inside = ty0 >= 1 && tx0 >= 1 && ty0 + C::TH + 1 <= H && tx0 + 17 <= W;
        constexpr bool HOIST = MID == 4;                        // conv.4's fragments are the same for every pixel block;
        bf16x8 w4h[HOIST ? C::KS : 1], w4l[HOIST ? C::KS : 1];   // keep them in registers where there is room
        if (HOIST) {
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const int boff = ((ks * 4 + kg) * 16 + lrow) * 16;
                w4h[HOIST ? ks : 0] = __builtin_bit_cast(bf16x8, *(const uint4*)(w4s_hi + boff));
                w4l[HOIST ? ks : 0] = __builtin_bit_cast(bf16x8, *(const uint4*)(w4s_lo + boff));
            }
        }
        // two pixel blocks per iteration (independent MFMA chains); the second may be past the end (clamped, not written)
        for (int blk0 = wave; blk0 < P::NBLK; blk0 += 8) {
            f32x4 acc4[2];
            int ryv[2], rxv[2];
            bool okv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int p = (blk0 + 4 * u) * 16 + lrow;
                okv[u] = p < P::NPX;
                p = okv[u] ? p : P::NPX - 1;
                ryv[u] = p / P::RW; rxv[u] = p - ryv[u] * P::RW;
                acc4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                bf16x8 wh, wl;
                if (HOIST) {
                    wh = w4h[HOIST ? ks : 0]; wl = w4l[HOIST ? ks : 0];
                } else {
                    const int boff = ((ks * 4 + kg) * 16 + lrow) * 16;
                    wh = __builtin_bit_cast(bf16x8, *(const uint4*)(w4s_hi + boff));
                    wl = __builtin_bit_cast(bf16x8, *(const uint4*)(w4s_lo + boff));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 xh, xl;
                    read_x_small<MID, H16>(h1_hi, h1_lo, P::N1SLOT, ryv[u] * P::R1 + rxv[u], P::R1, ks, kg, xh, xl);
                    if constexpr (H16) {
                        acc4[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, xh), acc4[u], 0, 0, 0);
                    } else {
                        MFMA_T(acc4[u], wh, wl, xh, xl);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // lane (lrow, kg) holds channels 4kg..4kg+3 of region pixel (ry, rx)
                const int ry = ryv[u], rx = rxv[u];
                float4 v = make_float4(acc4[u][0] + b4.x, acc4[u][1] + b4.y, acc4[u][2] + b4.z, acc4[u][3] + b4.w);
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                uint2 h, l;
                SPLIT4_T(v, h, l);
#define PAIR_PUT(slot_)                                                                            \
                if (MID == 4) {                                                                    \
                    *(uint2*)(a_hi + (slot_) * 8) = h;                                             \
                    *(uint2*)(a_lo + (slot_) * 8) = l;                                             \
                } else {                                                                           \
                    const int off_ = ((kg >> 1) * C::NSLOT + (slot_)) * 16 + (kg & 1) * 8;         \
                    *(uint2*)(a_hi + off_) = h;                                                    \
                    *(uint2*)(a_lo + off_) = l;                                                    \
                }
                if (ring_inside) {                           // uniform: the whole 18x18 ring region lies inside the image
                    if (okv[u] && 4 * kg < MID) { PAIR_PUT(ry * C::IW + rx); }
                } else {
                    const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
                    if (okv[u] && 4 * kg < MID && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                        // own slot + the mirrored ring slots (ReflectionPad2d(1) of h2): row -1 <- row 1, row H <- row H-2,
                        // same in x; where no mirror applies the same slot is written again
                        const int my = gy == 1 ? -2 : (gy == H - 2 ? 2 : 0), mx = gx == 1 ? -2 : (gx == W - 2 ? 2 : 0);
                        int sy = ry + my, sx = rx + mx;
                        sy = (sy < 0 || sy >= P::RH) ? ry : sy;
                        sx = (sx < 0 || sx >= P::RW) ? rx : sx;
                        PAIR_PUT(ry * C::IW + rx);
                        PAIR_PUT(sy * C::IW + rx);
                        PAIR_PUT(ry * C::IW + sx);
                        PAIR_PUT(sy * C::IW + sx);
                    }
                }
#undef PAIR_PUT
            }
        }
    }
    __syncthreads();                                         // h2 image complete; h1 region and conv.4 weights are dead
    if (P::ALIAS) {
        LAND_W7();
        __syncthreads();
    }
#undef LAND_W7

    // ---- conv.7 from the LDS image, state read-modify-write -------------------------------------------------------------
    f32x4 acc[C::MR][C::NB];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < C::NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int slot_base = (wave * C::MR) * C::IW + lrow;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        bf16x8 wh[C::NB], wl[C::NB];
#pragma unroll
        for (int n = 0; n < C::NB; ++n) {
            const int boff = ((ks * 4 + kg) * C::NT + n * 16 + lrow) * 16;
            wh[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(b_hi + boff));
            wl[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(b_lo + boff));
        }
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
            bf16x8 xh, xl;
            read_x_small<MID>(a_hi, a_lo, C::NSLOT, slot_base + m * C::IW, C::IW, ks, kg, xh, xl);
#pragma unroll
            for (int n = 0; n < C::NB; ++n) { MFMA_T(acc[m][n], wh[n], wl[n], xh, xl); }
        }
    }
    load_bias<CH, C::NB>(a, 4 * kg, bias);
    if constexpr (TERMS == 2) vst_note_range(range_amax);
    if (!EARLY_OLD) { PAIR_FETCH_OLD(); }
#undef PAIR_FETCH_OLD
    if (interior) store_tile<CH, true, C::MR, C::NB, true>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, 4 * kg, acc, bias, old);
    else store_tile<CH, true, C::MR, C::NB>(a, out_img, ty0 + wave * C::MR, tx0 + lrow, 4 * kg, acc, bias, old);
    VST_TRACE_END_(1, MID, CH)
}

#ifdef VST_PP_STAMP
// diagnostic build (-DVST_PP_STAMP=1, tools/pp_stamps.py): s_memtime stamps of one mid-grid workgroup's wave 0 (group X) and wave 4
// (group Y) around the parts of every stage; never in the shipped library
__device__ unsigned long long vst_pp_stamps[2 * 32 * 8];
extern "C" __attribute__((visibility("default"))) int vst_pp_stamps_dump(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vst_pp_stamps), sizeof(vst_pp_stamps));
}
#define PP_STAMP(grp_, st_, k_)                                                                                  \
    if (blockIdx.x == 128 && lane == 0 && (wave & 3) == 0 && (st_) < 32) {                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        vst_pp_stamps[((grp_) * 32 + (st_)) * 8 + (k_)] = __builtin_amdgcn_s_memtime();                          \
    }
// whole-kernel clock of the same workgroup: {s_memtime, s_memrealtime (100 MHz)} at its start and end
__device__ unsigned long long vst_pp_clk[4];
extern "C" __attribute__((visibility("default"))) int vst_pp_clk_dump(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vst_pp_clk), sizeof(vst_pp_clk));
}
#define PP_CLK(i_)                                                                                               \
    if (blockIdx.x == 128 && threadIdx.x == 0) {                                                                 \
        vst_pp_clk[2 * (i_)] = __builtin_amdgcn_s_memtime();                                                     \
        vst_pp_clk[2 * (i_) + 1] = __builtin_amdgcn_s_memrealtime();                                             \
    }
#else
#define PP_STAMP(grp_, st_, k_)
#define PP_CLK(i_)
#endif
// ---- pipelined kernel for the MFMA-bound shapes: CIN in {64,256}, COUT in {64,256}, stride 1 --------
// A stage = (32-channel chunk, tap row dy) = 3 k-steps = 144 MFMAs per wave.  Two activation buffers
// and two weight buffers in LDS; while stage s computes, the wave prefetches stage s+1's weights and a
// third of the next chunk's activations into registers and writes them to the other buffers after its
// MFMAs (one barrier per stage).  With COUT = 256 the four 64-channel output tiles are looped inside
// the workgroup: the activation image (both chunks) stays resident, only weights stream.
template <int CIN, int COUT, int NW_ = 8>
struct PipeCfg {
    // NW_ = 8: two waves per SIMD on a 16 x 16 tile, the workgroup owns its CU (135 KB of LDS).  NW_ = 4 (the "lean" form): one
    // wave per SIMD on an 8 x 16 tile, 96 KB of LDS and <= 256 VGPRs, so that a second workgroup - of this launch, or an
    // HBM-bound stage-1 / stage-2 workgroup of ANOTHER frame's stream (<= 61 KB, <= 256 VGPRs) - shares the CU
    static constexpr int NW = NW_, NTHR = 64 * NW;          // each wave owns MR = 2 tile rows (== launch bounds)
    static constexpr int NT = 64, NB = 4, MR = 2, TH = MR * NW, IW = 18, NPIX = (TH + 2) * IW, NSLOT = (NPIX + 15) / 16 * 16;
    static constexpr int NCHUNK = CIN / 32, NCOT = COUT / 64;
    static constexpr int A_PLANE = 4 * NSLOT * 16, A_BUF = 2 * A_PLANE;       // 43008 / 24576 (hi + lo planes of one 32-channel chunk)
    static constexpr int B_PLANE = 3 * 4 * 64 * 16, B_BUF = 2 * B_PLANE;      // 24576 (hi + lo, 3 k-steps x 64 channels)
    static constexpr int LDS_BYTES = 2 * A_BUF + 2 * B_BUF;                   // 135168 / 98304
    static constexpr int A_PART = NPIX * 4 / 3;                               // 432 / 240 (slot, cig) items per stage
    static constexpr int A_ITEMS = ((A_PART / 4 + 15) / 16 * 64 + NTHR - 1) / NTHR;   // per thread and part (lanes in 16-slot x 4-plane groups)
    static constexpr int B_ITEMS = 2 * 768 / NTHR;                            // uint4 per thread and stage
    static_assert(NPIX % 3 == 0, "three equal parts");
};

// Pipeline: a stage = (32-channel chunk, tap row dy) = 3 k-steps.  Two activation buffers and two weight
// buffers in LDS.  While stage s computes, the wave issues the global loads of stage s+2's weights and of a
// third of the next chunk's activations into registers (two register sets indexed by the compile-time
// stage parity: six stages are unrolled per loop body) and writes what it loaded during stage s-1 into the
// buffers of stage s+1; one barrier per stage.  Fragments are double-buffered in registers across k-steps.
// With COUT = 256 the four 64-channel output slices are looped inside the workgroup: the activation image
// (both chunks) stays resident, only weights stream.
template <int CIN, int COUT, bool IN_STATE, bool OUT_STATE, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_pipe_kernel(const ConvArgs a) {
    using C = PipeCfg<CIN, COUT, NW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + 2 * C::A_BUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    int bx, by, b;
    if (!xcd_tile(a, bx, by, b)) return;
    VST_TRACE_BEGIN(4)
    PP_CLK(0)
    const int tx0 = bx * 16, ty0 = by * C::TH;
    const float* const in_img = a.in + (size_t)b * a.in_img_stride;
    float* const out_img = a.out + (size_t)b * a.out_img_stride;
    const PackedConvLayout PL = packed_conv_layout(COUT, CIN);
    const unsigned char* const w_plane0 = a.packed + PL.f32_bytes;

    // activation staging: each third ("part") of the 18x18 x (4 channel-groups) image is 432 (slot, cig)
    // items, A_ITEMS per thread.  All uses index these arrays with compile-time constants (the tap-row
    // stages are unrolled), so they stay in registers.  Threads past the end repeat the last item.
    unsigned a_src[3][C::A_ITEMS], a_dst[3][C::A_ITEMS];
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int it = 0; it < C::A_ITEMS; ++it) {
            // 16 consecutive lanes take 16 consecutive slots of ONE channel-group plane (conflict-free ds_write_b128;
            // cig-fastest would put 4 lanes on the same banks) and together still read whole 128-byte pixel chunks
            const int j = it * C::NTHR + tid;
            int sl = (j >> 6) * 16 + (j & 15);                       // slot within this part (108 slots per part)
            sl = sl < C::A_PART / 4 ? sl : C::A_PART / 4 - 1;
            const int cig = (j >> 4) & 3, slot = part * (C::A_PART / 4) + sl;
            const int iy = slot / C::IW, ix = slot - iy * C::IW;
            const int gy = reflect_clamp(ty0 - 1 + iy, a.Hin), gx = reflect_clamp(tx0 - 1 + ix, a.Win);
            const size_t off = IN_STATE ? zc_offset(vst_level_of_channels(CIN), gy, gx, a.Wq)
                                        : ((size_t)gy * a.Win + gx) * CIN;
            a_src[part][it] = (unsigned)(off + cig * 8);
            a_dst[part][it] = (cig * C::NSLOT + slot) * 16;
        }
    // (native vector arrays filled by macros, every index a compile-time constant after unrolling: as structs returned from
    // lambdas the six-item stage of the 4-wave form was demoted to scratch)
    typedef f32x4 APart[C::A_ITEMS][2];
    typedef u32x4 BStage[C::B_ITEMS];
#define LOAD_A(r, chunk, part)                                                          \
    _Pragma("unroll") for (int it_ = 0; it_ < C::A_ITEMS; ++it_) {                      \
        const float* p_ = in_img + a_src[part][it_] + (chunk) * 32;                     \
        r[it_][0] = *(const f32x4*)p_; r[it_][1] = *(const f32x4*)(p_ + 4);             \
    }
#define F4_(v_) make_float4((v_)[0], (v_)[1], (v_)[2], (v_)[3])
#define STORE_A(buf, part, r)                                                           \
    _Pragma("unroll") for (int it_ = 0; it_ < C::A_ITEMS; ++it_) {                      \
        unsigned char* base_ = Abuf + (buf) * C::A_BUF;                                 \
        uint4 h_, l_;                                                                   \
        split8(F4_(r[it_][0]), F4_(r[it_][1]), h_, l_);                                 \
        *(uint4*)(base_ + a_dst[part][it_]) = h_;                                       \
        *(uint4*)(base_ + C::A_PLANE + a_dst[part][it_]) = l_;                          \
    }
    // weights of stage (q = cot*NCHUNK + chunk, dy): 2 planes (hi, lo) x 3 k-steps x 256 uint4 = 1536 items
    int b_off[C::B_ITEMS], b_dst[C::B_ITEMS];
#pragma unroll
    for (int it = 0; it < C::B_ITEMS; ++it) {
        if constexpr (NW == 4) {      // item it = (plane it / 3, k-step it % 3, thread = (kgi, co)): one base + constants
            const int plane = it / 3, k3 = it % 3, kgi = tid >> 6, co = tid & 63;
            b_off[it] = ((kgi * COUT + co) * 16) + (int)(plane * PL.frag_bytes) + k3 * 4 * COUT * 16;
            b_dst[it] = tid * 16 + plane * C::B_PLANE + k3 * 256 * 16;
        } else {
            const int idx = it * C::NTHR + tid;
            const int plane = idx >= 768, r = idx - plane * 768;
            const int k3 = r >> 8, kgi = (r >> 6) & 3, co = r & 63;
            b_off[it] = (int)(plane * PL.frag_bytes) + ((k3 * 4 + kgi) * COUT + co) * 16;
            b_dst[it] = plane * C::B_PLANE + r * 16;
        }
    }
#define LOAD_B(r, q_, dy_)                                                              \
    {                                                                                   \
        const int cot_ = (q_) / C::NCHUNK, chunk_ = (q_) - cot_ * C::NCHUNK;            \
        const unsigned char* src_ = w_plane0 + ((size_t)(chunk_ * 9 + (dy_) * 3) * 4 * COUT + cot_ * 64) * 16;   \
        _Pragma("unroll") for (int it_ = 0; it_ < C::B_ITEMS; ++it_) r[it_] = *(const u32x4*)(src_ + b_off[it_]); \
    }
#define STORE_B(buf, r)                                                                 \
    { _Pragma("unroll") for (int it_ = 0; it_ < C::B_ITEMS; ++it_) *(u32x4*)(Bbuf + (buf) * C::B_BUF + b_dst[it_]) = r[it_]; }
    struct Frags { bf16x8 wh[4], wl[4], xh[C::MR], xl[C::MR]; };
    auto read_frags = [&](Frags& f, const unsigned char* Ab, const unsigned char* Bb, int k3) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f.wh[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bb + (k3 * 256 + n * 16) * 16));
            f.wl[n] = __builtin_bit_cast(bf16x8, *(const uint4*)(Bb + C::B_PLANE + (k3 * 256 + n * 16) * 16));
        }
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
            f.xh[m] = __builtin_bit_cast(bf16x8, *(const uint4*)(Ab + (m * C::IW + k3) * 16));
            f.xl[m] = __builtin_bit_cast(bf16x8, *(const uint4*)(Ab + C::A_PLANE + (m * C::IW + k3) * 16));
        }
    };

    constexpr int Q = C::NCOT * C::NCHUNK;                  // (output slice, chunk) pairs; stage s = 3q + dy
    static_assert(Q % 2 == 0, "stage parity is static only for an even number of (slice, chunk) pairs");
    // ---- prologue: chunk 0 activations + stage 0 weights land in LDS; stage 1 weights and the first third of
    //      chunk 1 stay in registers (they are written to LDS during stage 0).  Register set[s&1] is loaded during
    //      stage s and stored during stage s+1.
    // (the 4-wave form stages twice as much per thread: it keeps ONE register set - a stage lands what the previous stage
    // loaded and then reuses the registers for its own loads; the 8-wave form issues its loads first, into a second set)
    constexpr bool ONE_SET = NW == 4;
#define RS(i_) (ONE_SET ? 0 : (i_))
    BStage rb[ONE_SET ? 1 : 2];
    APart ra[ONE_SET ? 1 : 2];
    {
        BStage rb0;
        LOAD_B(rb0, 0, 0);
        APart r0, r1, r2;
        LOAD_A(r0, 0, 0); LOAD_A(r1, 0, 1); LOAD_A(r2, 0, 2);
        LOAD_B(rb[RS(1)], 0, 1);
        if (C::NCHUNK > 1) LOAD_A(ra[RS(1)], 1, 0);
        STORE_A(0, 0, r0); STORE_A(0, 1, r1); STORE_A(0, 2, r2);
        STORE_B(0, rb0);
    }
    __syncthreads();

    f32x4 acc[C::MR][4];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int slot_base = (wave * C::MR) * C::IW + lrow;
    const int oy0 = ty0 + wave * C::MR;
    float4 bias[4], old[C::MR][4];
#if VST_ABLATE & 8
    f32x16 abl_big[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { abl_big[0][i] = 0.f; abl_big[1][i] = 0.f; }
#endif
    const bool full_tile = ty0 + C::TH <= a.Hout && tx0 + 16 <= a.Wout; // uniform: interior tiles skip all predicates
    // Kernels that loop several 64-channel output slices (the 64 -> 256 conv: four) DEFER a slice's stores: at the slice's end
    // the results are formed in place in `old` (old + sign * (acc + bias)), and the eight float4 stores per lane go out one per
    // k-step during the next slice's first chunk, between its MFMAs - all eight waves used to issue them together behind the
    // slice's last MFMA, in front of the stage barrier (a timing-only build without three of the four epilogues:
    // 59.5 -> 49.5 us).  `old` is free for that long: the next slice loads its own old values at its last stage.
    constexpr bool DEFER = OUT_STATE && C::NCOT > 1 && C::NCHUNK == 2 && !VST_PIPE_NO_DEFER;
    bool pending = false;                                    // uniform: `old` holds a finished slice whose stores are still due
    int pend_cot = 0;

#pragma unroll 1
    for (int q0 = 0; q0 < Q; q0 += 2) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = q0 + qq;
        const int cot = q / C::NCHUNK, chunk = q - cot * C::NCHUNK;
        const bool last_chunk = chunk == C::NCHUNK - 1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int s = q * 3 + dy;
            const int cur = (qq * 3 + dy) & 1;            // compile-time parity of s
            // ---- land what was loaded one stage ago in the buffers of stage s+1 (free since the last barrier) -------
#define PIPE_LAND()                                                                                 \
            if (!(VST_ABLATE & 2)) {                                                                \
                if (s + 1 < 3 * Q) STORE_B((s + 1) & 1, rb[RS(cur ^ 1)]);                            \
                if (q < C::NCHUNK - 1) STORE_A((chunk + 1) & 1, dy, ra[RS(cur ^ 1)]);                \
            }
            if constexpr (ONE_SET) { PIPE_LAND(); }
            // ---- issue the loads that are two stages ahead (consumed from registers during the NEXT stage) ----
            if (!(VST_ABLATE & 2)) {   // weights of stage s+2
                const int q2 = dy == 0 ? q : q + 1, dy2 = (dy + 2) % 3;
                if (q2 < Q) LOAD_B(rb[RS(cur)], q2, dy2);
                // activations stored during stage s+1 = (qn, dyn): part dyn of chunk(qn)+1 (first output slice only)
                const int qn = dy == 2 ? q + 1 : q, dyn = (dy + 1) % 3;
                if (qn < C::NCHUNK - 1) LOAD_A(ra[RS(cur)], qn + 1, dyn);
            }
            // (bias: a few cached bytes.  The multi-slice kernels fetch it at the slice's LAST stage top - live for one stage
            // instead of six - which pays for the longer-lived old values below)
            if (DEFER ? (dy == 2 && last_chunk) : (dy == 0 && chunk == 0)) load_bias<COUT, 4>(a, cot * 64 + 4 * kg, bias);
            // old state values of this output slice, used after its last MFMA.  (Fetching them one or two stages earlier - the
            // DEFER kernels' `old` registers are free by then - was measured and is WORSE, 60.1 / 60.4 against 58.4 us: vmcnt is
            // in-order, so the next stage's wait for its weights also waits for these older, slower loads.)
            // The DEFER kernels fetch them SPREAD instead: unit (m, n) right after the previous slice's unit (m, n) has been
            // stored from the same registers, one float4 per lane and k-step over the slice's first chunk - the launch's 256
            // workgroups run in lockstep, and a whole slice's old values at one stage top are a 16 MB burst (~3 us of HBM time
            // in front of a 1.3 us stage); a few small loads per stage, a stage or more old when used, stall nobody.
            if (OUT_STATE && !(DEFER && VST_PIPE_OLD_SPREAD) && dy == 2 && last_chunk && (!(VST_ABLATE & 4) || cot == C::NCOT - 1)) {
                if (full_tile) load_old<COUT, C::MR, 4, true>(a, out_img, oy0, tx0 + lrow, cot * 64 + 4 * kg, old);
                else load_old<COUT, C::MR, 4, false>(a, out_img, oy0, tx0 + lrow, cot * 64 + 4 * kg, old);
            }

            if constexpr (!ONE_SET) { PIPE_LAND(); }
#undef PIPE_LAND

            // ---- 3 k-steps of MFMAs on the current buffers; fragments are double-buffered in registers: the reads
            //      of k-step k+1 are issued before the MFMAs of k-step k --------------------------------------------
            const unsigned char* Ab = Abuf + (chunk & 1) * C::A_BUF + (kg * C::NSLOT + slot_base + dy * C::IW) * 16;
            const unsigned char* Bb = Bbuf + (s & 1) * C::B_BUF + (kg * 64 + lrow) * 16;
            Frags fr[2];
            read_frags(fr[0], Ab, Bb, 0);
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) {
                __builtin_amdgcn_sched_barrier(0);            // reads of k-step k3+1 stay ahead of the MFMAs of k3
                if (k3 < 2 && !(VST_ABLATE & 1)) read_frags(fr[(k3 + 1) & 1], Ab, Bb, k3 + 1);
                if constexpr (DEFER) {
                    if (qq == 0 && dy * 3 + k3 < C::MR * 4) {                  // unit dy*3+k3 = (m, n)
                        const int unit = dy * 3 + k3, m_ = unit >> 2, n_ = unit & 3;
                        if (pending) {                                         // ... of the previous slice: its deferred store
                            float4* p_ = full_tile ? out_ptr<COUT, OUT_STATE, true>(a, out_img, oy0 + m_, tx0 + lrow, pend_cot * 64 + 4 * kg + n_ * 16)
                                                   : out_ptr<COUT, OUT_STATE, false>(a, out_img, oy0 + m_, tx0 + lrow, pend_cot * 64 + 4 * kg + n_ * 16);
                            if (full_tile || p_) *p_ = old[m_][n_];
                        }
                        if (VST_PIPE_OLD_SPREAD) {                             // ... of this slice: its old state value, see below
                            const float4* q_ = full_tile ? out_ptr<COUT, OUT_STATE, true>(a, out_img, oy0 + m_, tx0 + lrow, cot * 64 + 4 * kg + n_ * 16)
                                                         : out_ptr<COUT, OUT_STATE, false>(a, out_img, oy0 + m_, tx0 + lrow, cot * 64 + 4 * kg + n_ * 16);
                            old[m_][n_] = (full_tile || q_) ? *q_ : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const Frags& f = fr[(VST_ABLATE & 1) ? 0 : (k3 & 1)];
#if VST_ABLATE & 8
                // timing-only (wrong results): the k-step's 24 v_mfma_f32_16x16x32 as 12 v_mfma_f32_32x32x16 on the same
                // fragment registers - the same matrix-pipe cycles with half the MFMA instructions (is the stage issue-bound?)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    abl_big[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[2 * jb], f.xl[0], abl_big[jb], 0, 0, 0);
                    abl_big[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wl[2 * jb], f.xh[0], abl_big[jb], 0, 0, 0);
                    abl_big[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[2 * jb], f.xh[0], abl_big[jb], 0, 0, 0);
                    abl_big[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[2 * jb + 1], f.xl[1], abl_big[jb], 0, 0, 0);
                    abl_big[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wl[2 * jb + 1], f.xh[1], abl_big[jb], 0, 0, 0);
                    abl_big[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[2 * jb + 1], f.xh[1], abl_big[jb], 0, 0, 0);
                }
#else
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) { MFMA3(acc[m][n], f.wh[n], f.wl[n], f.xh[m], f.xl[m]); }
#endif
            }

            // ---- output tile of this 64-channel slice ------------------------------------------------------------
            if (DEFER && dy == 2 && last_chunk && cot != C::NCOT - 1 && !(VST_ABLATE & 4)) {
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const float4 o = old[m][n];
                        old[m][n] = make_float4(o.x + a.sign * (acc[m][n][0] + bias[n].x), o.y + a.sign * (acc[m][n][1] + bias[n].y),
                                                o.z + a.sign * (acc[m][n][2] + bias[n].z), o.w + a.sign * (acc[m][n][3] + bias[n].w));
                        acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                pending = true;
                pend_cot = cot;
            } else
            if (dy == 2 && last_chunk && (!(VST_ABLATE & 4) || cot == C::NCOT - 1)) {   // (ablation 4: only the last slice's epilogue)
                if (full_tile) store_tile<COUT, OUT_STATE, C::MR, 4, true>(a, out_img, oy0, tx0 + lrow, cot * 64 + 4 * kg, acc, bias, old);
                else store_tile<COUT, OUT_STATE, C::MR, 4, false>(a, out_img, oy0, tx0 + lrow, cot * 64 + 4 * kg, acc, bias, old);
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (DEFER && qq == 0 && dy == 2) pending = false;       // the previous slice's eight units went out in this chunk
            if ((VST_ABLATE & 4) && dy == 2 && last_chunk && cot != C::NCOT - 1) {   // timing-only: a slice without its epilogue
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(acc[m][n]));
            }
            if (!(VST_ABLATE & 16)) __syncthreads();                // (ablation 16, timing-only: no stage barrier)
        }
      }
    }
#undef LOAD_A
#undef STORE_A
#undef LOAD_B
#undef STORE_B
#undef F4_
#undef RS
#if VST_ABLATE & 8
    asm volatile("" :: "v"(abl_big[0]), "v"(abl_big[1]));
#endif
    PP_CLK(1)
    VST_TRACE_END_(4, CIN, COUT)
}

#ifndef VST_WITH_PINGPONG
#define VST_WITH_PINGPONG 0          // 1 (tools/ab_build.py ... -DVST_WITH_PINGPONG=1): compile conv_pp_kernel, selectable by VST_OPT_STAGE3_PINGPONG
#endif
#if VST_WITH_PINGPONG
// ---- ping-pong form of the pipelined kernel (VST_OPT_STAGE3_PINGPONG) -------------------------------------------------------
// Same tile, LDS images, fragments and MFMA order per accumulator as conv_pipe_kernel (bit-identical results); what changes is
// WHEN the two waves of a SIMD do what.  In conv_pipe_kernel all eight waves pass every stage in lockstep: they land data,
// read their first fragments and wait together (matrix pipe idle: ~2 000 of a stage's ~4 300 cycles), then share the pipe.  Here
// the workgroup is two groups of four waves, one wave per SIMD each - X = waves 0-3 (tile rows 0-7), Y = waves 4-7 (rows 8-15;
// wave w and w + 4 share a SIMD) - that alternate: while X issues its 72 MFMAs of stage s back to back, Y does everything that
// is not matrix work (landing the staged weights / activations in LDS, issuing the next global loads, the read-modify-write
// traffic of its output rows, reading its first fragments of stage s); one barrier; then Y computes stage s while X reads its
// first fragments of stage s + 1 and moves its output rows.  Half-steps h = 0, 1, 2, ...: X computes stage s at h = 2s, Y at
// h = 2s + 1.  LDS hazards, one barrier per half-step:
//   * weights of stage s + 1 go to B[(s+1)&1] at h = 2s (by Y): last read at h = 2s - 1 (Y, stage s - 1), first read by X's
//     fragment prefetch at h = 2s + 1;
//   * activations of chunk c + 1 go to A[(c+1)&1] in thirds at h = 6c, 6c + 2, 6c + 4 (by Y): last read at h = 6c - 1 (Y, chunk
//     c - 1), first read at h = 6c + 5 (X's prefetch for stage 3(c+1)).
// All staging is done by Y (its segment runs beside X's MFMA burst; X's segment beside Y's burst holds only the fragment
// prefetch and X's share of the epilogue), so every fragment a burst starts with is in registers before the barrier.
// Measured (round 4, profiles/r04_pingpong.json): bit-identical, and NOT faster - 58.1 / 22.5 / 54.0 us against 56.9 / 20.9 / 52.2
// for 256->64 / 64->64 / 64->256.  In-kernel stamps: a burst of 72 MFMAs takes 1 450 - 1 600 cycles (1 152 at the pipe's rate) with
// the partner's segment beside it, a stage 3 300 - 3 600 cycles - what conv_pipe_kernel's lockstep stage takes as well: the SIMD's
// instruction issue (MFMA 8 of every 16 cycles, the fragment reads, the partner's ~150 staging instructions), not the order of
// the phases, is what a stage costs.  Kept out of the shipped library (VST_WITH_PINGPONG = 0).
template <int CIN, int COUT, bool IN_STATE, bool OUT_STATE>
__global__ __launch_bounds__(512) void conv_pp_kernel(const ConvArgs a) {
    using C = PipeCfg<CIN, COUT, 8>;
    constexpr int SY = 256;                                                  // staging threads (group Y)
    constexpr int A_ITEMS_Y = ((C::A_PART / 4 + 15) / 16 * 64 + SY - 1) / SY;   // (slot, cig) items per Y thread and part: 2
    constexpr int B_ITEMS_Y = 2 * 768 / SY;                                  // uint4 per Y thread and stage: 6
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + 2 * C::A_BUF;
    float* const bias_lds = (float*)(smem + C::LDS_BYTES);                  // (the sliced kernel: COUT floats behind the buffers)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, kg = lane >> 4;
    const bool grpY = wave >= 4;                                             // wave-uniform
    int bx, by, b;
    if (!xcd_tile(a, bx, by, b)) return;
    VST_TRACE_BEGIN(4)
    PP_CLK(0)
    const int tx0 = bx * 16, ty0 = by * C::TH;
    const float* const in_img = a.in + (size_t)b * a.in_img_stride;
    float* const out_img = a.out + (size_t)b * a.out_img_stride;
    const PackedConvLayout PL = packed_conv_layout(COUT, CIN);
    const unsigned char* const w_plane0 = a.packed + PL.f32_bytes;
    constexpr int Q = C::NCOT * C::NCHUNK, S = 3 * Q;                        // (slice, chunk) pairs; stages
    // 256-channel inputs stream their eight chunk images through the two buffers inside the loop (group Y stages them); the
    // two chunk images of a 64-channel input (h1 / h2) are both staged by the prologue and stay: no activation traffic, staging
    // registers or address tables in the loop of the 64 -> 64 and 64 -> 256 convs
    constexpr bool A_IN_LOOP = C::NCHUNK > 2;

#define PP_A_ADDR(j_, part_, src_, dst_)                                                                         \
    {                                                                                                            \
        int sl_ = ((j_) >> 6) * 16 + ((j_) & 15);                                                                \
        sl_ = sl_ < C::A_PART / 4 ? sl_ : C::A_PART / 4 - 1;                                                     \
        const int cig_ = ((j_) >> 4) & 3, slot_ = (part_) * (C::A_PART / 4) + sl_;                               \
        const int iy_ = slot_ / C::IW, ix_ = slot_ - iy_ * C::IW;                                                \
        const int gy_ = reflect_clamp(ty0 - 1 + iy_, a.Hin), gx_ = reflect_clamp(tx0 - 1 + ix_, a.Win);          \
        const size_t off_ = IN_STATE ? zc_offset(vst_level_of_channels(CIN), gy_, gx_, a.Wq)                     \
                                     : ((size_t)gy_ * a.Win + gx_) * CIN;                                        \
        src_ = (unsigned)(off_ + cig_ * 8);                                                                      \
        dst_ = (cig_ * C::NSLOT + slot_) * 16;                                                                   \
    }
#define PP_F4(v_) make_float4((v_)[0], (v_)[1], (v_)[2], (v_)[3])
#define PP_LAND_A(base_, r0_, r1_, dst_)                                                                         \
    {                                                                                                            \
        uint4 h_, l_;                                                                                            \
        split8(PP_F4(r0_), PP_F4(r1_), h_, l_);                                                                  \
        *(uint4*)((base_) + (dst_)) = h_;                                                                        \
        *(uint4*)((base_) + C::A_PLANE + (dst_)) = l_;                                                           \
    }
    // ---- prologue (all eight waves): chunk 0's image and stage 0's weights --------------------------------------------------
    if constexpr (OUT_STATE) {                   // the bias of all four slices waits in LDS (16 registers less in the loop)
        if (tid < COUT / 4) *(float4*)(bias_lds + 4 * tid) = *(const float4*)(a.bias + 4 * tid);
    }
    {
        u32x4 wb[3];
        f32x4 r[3][2];
        unsigned dsta[3];
        int dstb[3];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = it * 512 + tid;
            const int plane = idx >= 768, rr = idx - plane * 768;
            const int k3 = rr >> 8, kgi = (rr >> 6) & 3, co = rr & 63;
            wb[it] = *(const u32x4*)(w_plane0 + (size_t)(plane * PL.frag_bytes) + ((k3 * 4 + kgi) * COUT + co) * 16);
            dstb[it] = plane * C::B_PLANE + rr * 16;
        }
        // two-chunk inputs (h1 / h2): the second chunk's image as well - in the same burst of loads where the registers allow it
        // (the 64 -> 256 conv's loop is at the register limit: it lands chunk 0 first and reuses the registers)
        constexpr bool BURST2 = !A_IN_LOOP && !OUT_STATE;
        f32x4 r1[BURST2 ? 3 : 1][2];
        unsigned srca[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            PP_A_ADDR(tid, part, srca[part], dsta[part]);
            const float* p = in_img + srca[part];
            r[part][0] = *(const f32x4*)p;
            r[part][1] = *(const f32x4*)(p + 4);
            if constexpr (BURST2) {
                r1[part][0] = *(const f32x4*)(p + 32);
                r1[part][1] = *(const f32x4*)(p + 36);
            }
        }
#pragma unroll
        for (int part = 0; part < 3; ++part) PP_LAND_A(Abuf, r[part][0], r[part][1], dsta[part]);
#pragma unroll
        for (int it = 0; it < 3; ++it) *(u32x4*)(Bbuf + dstb[it]) = wb[it];
        if constexpr (BURST2) {
#pragma unroll
            for (int part = 0; part < 3; ++part) PP_LAND_A(Abuf + C::A_BUF, r1[part][0], r1[part][1], dsta[part]);
        } else if constexpr (!A_IN_LOOP) {
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const float* p = in_img + srca[part] + 32;
                r[part][0] = *(const f32x4*)p;
                r[part][1] = *(const f32x4*)(p + 4);
            }
#pragma unroll
            for (int part = 0; part < 3; ++part) PP_LAND_A(Abuf + C::A_BUF, r[part][0], r[part][1], dsta[part]);
        }
    }
    // ---- group Y's staging state: addresses of its items, one register set (landed, then reloaded, in every Y segment) ------
    const int ty = tid & (SY - 1);
    unsigned a_src[A_IN_LOOP ? 3 : 1][A_ITEMS_Y], a_dst[A_IN_LOOP ? 3 : 1][A_ITEMS_Y];
    if constexpr (A_IN_LOOP) {
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int it = 0; it < A_ITEMS_Y; ++it) PP_A_ADDR(it * SY + ty, part, a_src[part][it], a_dst[part][it]);
    }
    const int b_off0 = ((ty >> 6) * COUT + (ty & 63)) * 16, b_dst0 = ty * 16;     // item it = (plane it / 3, k-step it % 3)
    u32x4 rb[B_ITEMS_Y];
    f32x4 ra[A_ITEMS_Y][2];
#define PP_LOAD_B(st_)                                                                                           \
    {                                                                                                            \
        const int q_ = (st_) / 3, dy_ = (st_) - 3 * q_, cot_ = q_ / C::NCHUNK, chunk_ = q_ - cot_ * C::NCHUNK;   \
        const unsigned char* src_ = w_plane0 + ((size_t)(chunk_ * 9 + dy_ * 3) * 4 * COUT + cot_ * 64) * 16 + b_off0;   \
        _Pragma("unroll") for (int it_ = 0; it_ < B_ITEMS_Y; ++it_)                                              \
            rb[it_] = *(const u32x4*)(src_ + (size_t)(it_ / 3) * PL.frag_bytes + (it_ % 3) * 4 * COUT * 16);     \
    }
#define PP_STORE_B(buf_)                                                                                         \
    _Pragma("unroll") for (int it_ = 0; it_ < B_ITEMS_Y; ++it_)                                                  \
        *(u32x4*)(Bbuf + (buf_) * C::B_BUF + b_dst0 + (it_ / 3) * C::B_PLANE + (it_ % 3) * 256 * 16) = rb[it_];
#define PP_LOAD_A(chunk_, part_)                                                                                 \
    _Pragma("unroll") for (int it_ = 0; it_ < A_ITEMS_Y; ++it_) {                                                \
        const float* p_ = in_img + a_src[part_][it_] + (chunk_) * 32;                                            \
        ra[it_][0] = *(const f32x4*)p_; ra[it_][1] = *(const f32x4*)(p_ + 4);                                    \
    }
#define PP_STORE_A(buf_, part_)                                                                                  \
    _Pragma("unroll") for (int it_ = 0; it_ < A_ITEMS_Y; ++it_)                                                  \
        PP_LAND_A(Abuf + (buf_) * C::A_BUF, ra[it_][0], ra[it_][1], a_dst[part_][it_])
    if (grpY) {                                  // what Y lands in its first segment (h = 0): stage 1's weights, chunk 1's first third
        PP_LOAD_B(1);
        if constexpr (A_IN_LOOP) { PP_LOAD_A(1, 0); }
    }
#define PP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    PP_BARRIER();

    // fragments: the pixel operands of a k-step (xh, xl for the wave's two tile rows) are double-buffered per k-step, the weight
    // operands stream per 16-channel block n through a ring of three (step t = 4 k3 + n reads slot t % 3; the read of step t + 2
    // is issued before the six MFMAs of step t) - 56 registers where whole k-steps double-buffered take 96
    bf16x8 fxh[2][C::MR], fxl[2][C::MR], fwh[3], fwl[3];
#define PP_READ_X(buf_, Ab_, k3_)                                                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < C::MR; ++m_) {                                                       \
        fxh[buf_][m_] = __builtin_bit_cast(bf16x8, *(const uint4*)((Ab_) + (m_ * C::IW + (k3_)) * 16));          \
        fxl[buf_][m_] = __builtin_bit_cast(bf16x8, *(const uint4*)((Ab_) + C::A_PLANE + (m_ * C::IW + (k3_)) * 16));   \
    }
#define PP_READ_W(slot_, Bb_, t_)                                                                                \
    {                                                                                                            \
        fwh[slot_] = __builtin_bit_cast(bf16x8, *(const uint4*)((Bb_) + (((t_) >> 2) * 256 + ((t_) & 3) * 16) * 16));   \
        fwl[slot_] = __builtin_bit_cast(bf16x8, *(const uint4*)((Bb_) + C::B_PLANE + (((t_) >> 2) * 256 + ((t_) & 3) * 16) * 16));   \
    }
    // what a burst starts with: the pixel operands of k-step 0 and the weight operands of steps 0 and 1 (read in the segment
    // BEFORE the barrier that opens the burst)
#define PP_PREFETCH(st_)                                                                                         \
    {                                                                                                            \
        const unsigned char* Ab_ = PP_AB(st_);                                                                   \
        const unsigned char* Bb_ = PP_BB(st_);                                                                   \
        PP_READ_X(0, Ab_, 0);                                                                                    \
        PP_READ_W(0, Bb_, 0);                                                                                    \
        PP_READ_W(1, Bb_, 1);                                                                                    \
    }
    const int slot_base = (wave * C::MR) * C::IW + lrow;
    const int oy0 = ty0 + wave * C::MR;
    const unsigned a_lane = (kg * C::NSLOT + slot_base) * 16, b_lane = (kg * 64 + lrow) * 16;
    // LDS addresses of stage st's operands for this lane
#define PP_AB(st_) (Abuf + ((((st_) / 3) % C::NCHUNK) & 1) * C::A_BUF + a_lane + ((st_) % 3) * C::IW * 16)
#define PP_BB(st_) (Bbuf + ((st_) & 1) * C::B_BUF + b_lane)

    f32x4 acc[C::MR][4];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bias[4], old[C::MR][4];
    const bool full_tile = ty0 + C::TH <= a.Hout && tx0 + 16 <= a.Wout;     // uniform: interior tiles skip all predicates
    constexpr bool SLICED = OUT_STATE && C::NCOT > 1;                        // the 64 -> 256 conv: four 64-channel output slices
    static_assert(!OUT_STATE || (C::NCOT > 1 && C::NCHUNK == 2), "read-modify-write epilogue: the 64 -> 256 conv");
    bool pending = false;                                                    // `old` holds a finished slice whose stores are due
    int pend_cot = 0;
    // the matrix burst of stage st: 12 steps (k-step k3, channel block n) of 6 MFMAs
#define PP_COMPUTE(st_)                                                                                          \
    {                                                                                                            \
        const unsigned char* Ab_ = PP_AB(st_);                                                                   \
        const unsigned char* Bb_ = PP_BB(st_);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int t = 0; t < 12; ++t) {                                                         \
            const int k3 = t >> 2, n = t & 3;                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if (n == 0 && k3 < 2) PP_READ_X((k3 + 1) & 1, Ab_, k3 + 1);                                          \
            if (t + 2 < 12) PP_READ_W((t + 2) % 3, Bb_, t + 2);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            _Pragma("unroll") for (int m = 0; m < C::MR; ++m) { MFMA3(acc[m][n], fwh[t % 3], fwl[t % 3], fxh[k3 & 1][m], fxl[k3 & 1][m]); }   \
        }                                                                                                        \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    }
    // the 64 -> 256 conv's epilogue, spread over the group's own staging segments.  j = stage within the slice (0..5):
    //   a slice's results are formed in place in `old` (old + sign * (acc + bias)) once its last burst is done, and leave - two
    //   units (m, n) per segment - during the next slice's first four segments, each followed by the load of the same unit's
    //   old state value of the new slice (in flight for two segments or more before it is used)
#define PP_FINISH(cot_)                                                                                          \
    {                                                                                                            \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) bias[n] = *(const float4*)(bias_lds + (cot_) * 64 + 4 * kg + n * 16);   \
        _Pragma("unroll") for (int m = 0; m < C::MR; ++m)                                                        \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                      \
                const float4 o = old[m][n];                                                                      \
                old[m][n] = make_float4(o.x + a.sign * (acc[m][n][0] + bias[n].x), o.y + a.sign * (acc[m][n][1] + bias[n].y),   \
                                        o.z + a.sign * (acc[m][n][2] + bias[n].z), o.w + a.sign * (acc[m][n][3] + bias[n].w));  \
                acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};                                                           \
            }                                                                                                    \
        pending = true;                                                                                          \
        pend_cot = (cot_);                                                                                       \
    }
#define PP_UNITS(j_, cot_)                                                                                       \
    if ((j_) < 4) {                                                                                              \
        _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                                       \
            const int unit = 2 * (j_) + u_, m_ = unit >> 2, n_ = unit & 3;                                       \
            if (pending) {                                                                                       \
                float4* p_ = full_tile ? out_ptr<COUT, true, true>(a, out_img, oy0 + m_, tx0 + lrow, pend_cot * 64 + 4 * kg + n_ * 16)   \
                                       : out_ptr<COUT, true, false>(a, out_img, oy0 + m_, tx0 + lrow, pend_cot * 64 + 4 * kg + n_ * 16); \
                if (full_tile || p_) *p_ = old[m_][n_];                                                          \
            }                                                                                                    \
            const float4* q_ = full_tile ? out_ptr<COUT, true, true>(a, out_img, oy0 + m_, tx0 + lrow, (cot_) * 64 + 4 * kg + n_ * 16)   \
                                         : out_ptr<COUT, true, false>(a, out_img, oy0 + m_, tx0 + lrow, (cot_) * 64 + 4 * kg + n_ * 16); \
            old[m_][n_] = (full_tile || q_) ? *q_ : make_float4(0.f, 0.f, 0.f, 0.f);                             \
        }                                                                                                        \
        if ((j_) == 3) pending = false;                                                                          \
    }
    if (!grpY) { PP_PREFETCH(0); }                                             // X's first burst starts at once

    // (two (slice, chunk) pairs per loop body: the chunk's parity - the activation buffer, and for the sliced kernel the stage's
    // position j in its slice, which picks the epilogue units - is then a compile-time constant)
    static_assert(Q % 2 == 0 && C::NCHUNK % 2 == 0, "two (slice, chunk) pairs per loop body");
#pragma unroll 1
    for (int q0 = 0; q0 < Q; q0 += 2) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = q0 + qq;
        const int cot = q / C::NCHUNK, chunk = q - cot * C::NCHUNK;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int s = q * 3 + dy;
            const int j = (C::NCHUNK == 2 ? qq * 3 : 0) + dy;               // stage within the output slice (SLICED kernels)
            if (!grpY) {
                // ================= X: burst of stage s, then its segment beside Y's burst =================================
                PP_STAMP(0, s, 0)
                PP_COMPUTE(s);
                PP_STAMP(0, s, 1)
                PP_BARRIER();
                PP_STAMP(0, s, 2)
                if (s + 1 < S) { PP_PREFETCH(s + 1); }
                if constexpr (SLICED) {
                    PP_UNITS(j, cot);
                    if (j == 5 && cot < C::NCOT - 1) { PP_FINISH(cot); }
                }
                PP_STAMP(0, s, 3)
                PP_BARRIER();
            } else {
                // ================= Y: its segment beside X's burst of stage s, then its own burst =========================
                PP_STAMP(1, s, 0)
                PP_PREFETCH(s);
                if (s + 1 < S) { PP_STORE_B((s + 1) & 1); }
                if constexpr (A_IN_LOOP) {
                    if (q < C::NCHUNK - 1) { PP_STORE_A((chunk + 1) & 1, dy); }
                }
                PP_STAMP(1, s, 1)
                if (s + 2 < S) { PP_LOAD_B(s + 2); }
                if constexpr (A_IN_LOOP) {   // the activations landed in the NEXT Y segment (stage s + 1 = (q1, dy1)): part dy1 of chunk(q1) + 1
                    const int q1 = dy == 2 ? q + 1 : q, dy1 = (dy + 1) % 3;
                    if (q1 < C::NCHUNK - 1) { PP_LOAD_A(q1 + 1, dy1); }
                }
                if constexpr (SLICED) {
                    if (j == 0 && cot > 0) { PP_FINISH(cot - 1); }
                    PP_UNITS(j, cot);
                }
                PP_STAMP(1, s, 2)
                PP_BARRIER();
                PP_STAMP(1, s, 3)
                PP_COMPUTE(s);
                PP_STAMP(1, s, 4)
                PP_BARRIER();
            }
        }
      }
    }
    // ---- the last output slice (or the only one) ------------------------------------------------------------------------------
    if constexpr (SLICED) {
#pragma unroll
        for (int n = 0; n < 4; ++n) bias[n] = *(const float4*)(bias_lds + (C::NCOT - 1) * 64 + 4 * kg + n * 16);
    } else {
        load_bias<COUT, 4>(a, 4 * kg, bias);
    }
    if (full_tile) store_tile<COUT, OUT_STATE, C::MR, 4, true>(a, out_img, oy0, tx0 + lrow, (C::NCOT - 1) * 64 + 4 * kg, acc, bias, old);
    else store_tile<COUT, OUT_STATE, C::MR, 4, false>(a, out_img, oy0, tx0 + lrow, (C::NCOT - 1) * 64 + 4 * kg, acc, bias, old);
#undef PP_A_ADDR
#undef PP_F4
#undef PP_LAND_A
#undef PP_LOAD_B
#undef PP_STORE_B
#undef PP_LOAD_A
#undef PP_STORE_A
#undef PP_BARRIER
#undef PP_AB
#undef PP_BB
#undef PP_COMPUTE
#undef PP_READ_X
#undef PP_READ_W
#undef PP_PREFETCH
#undef PP_FINISH
#undef PP_UNITS
    PP_CLK(1)
    VST_TRACE_END_(4, CIN, COUT)
}

#endif  // VST_WITH_PINGPONG

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
// All of it is behind one lock: launch sites on any host thread may open records while a session is active.
#define VST_PROFILE_MAX_RECORDS 4096
static std::mutex g_prof_mu;
static std::atomic<int> g_prof_kernel{0};   // 0 = off, else VST_KERNEL_ID(cin, cout, stride)
static int g_prof_count = 0, g_prof_cap = 0;
static int g_prof_id[VST_PROFILE_MAX_RECORDS];
static hipEvent_t g_prof_ev[2 * VST_PROFILE_MAX_RECORDS];
static bool g_prof_ev_created = false;

int vst_prof_open(int kernel_id, hipStream_t st) {
    const int sel = g_prof_kernel.load(std::memory_order_relaxed);
    if (sel != kernel_id && sel != VST_KERNEL_ALL) return -1;                       // the common case: no lock taken
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((g_prof_kernel.load() != kernel_id && g_prof_kernel.load() != VST_KERNEL_ALL) || g_prof_count >= g_prof_cap) return -1;
    const int rec = g_prof_count++;
    g_prof_id[rec] = kernel_id;
    (void)hipEventRecord(g_prof_ev[2 * rec], st);
    return rec;
}

void vst_prof_close(int rec, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (rec < g_prof_count) (void)hipEventRecord(g_prof_ev[2 * rec + 1], st);
}

// VST_OPT_STAGE3_LEAN (vstnet.h): the stage-3 convs of the bf16x3 mode as half-CU workgroups, see PipeCfg
#ifndef VST_LEAN_DEFAULT
#define VST_LEAN_DEFAULT 0
#endif
static std::atomic<int> g_opt_lean{[] { const char* e = getenv("VST_LEAN"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : VST_LEAN_DEFAULT; }()};
static bool vst_lean_stage3() { return g_opt_lean.load(std::memory_order_relaxed) != 0; }
// VST_OPT_STAGE3_PINGPONG (vstnet.h): conv_pp_kernel instead of conv_pipe_kernel
#ifndef VST_PINGPONG_DEFAULT
#define VST_PINGPONG_DEFAULT 0
#endif
static std::atomic<int> g_opt_pp{[] { const char* e = getenv("VST_PINGPONG"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : VST_PINGPONG_DEFAULT; }()};

template <int CIN, int COUT, int STRIDE, bool IN_STATE, bool OUT_STATE>
static int launch_conv(const ConvArgs& a, int B, int precision, hipStream_t st, bool out_h16 = false) {
    if (precision == VST_PREC_FP32) {
        const size_t total = (size_t)B * a.Hout * a.Wout * COUT;
        size_t blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        conv_fp32_kernel<IN_STATE, OUT_STATE><<<dim3((unsigned)blocks), 256, 0, st>>>(a, CIN, COUT, STRIDE, B);
        VST_RETURN_IF_LAUNCH_FAILED();
        return VST_OK;
    }
    if (precision != VST_PREC_BF16X3 && !vst_is_f16(precision)) return VST_E_MODE;
    vst_prof_scope prof(VST_KERNEL_ID(CIN, COUT, STRIDE), st);
    if constexpr (CIN >= 64 && COUT >= 64 && STRIDE == 1) {
        ConvArgs t = a;
        t.tiles_x = (a.Wout + 15) / 16;
        if (vst_lean_stage3()) {                 // half-CU workgroups: see PipeCfg
            using C = PipeCfg<CIN, COUT, 4>;
            auto kern = conv_pipe_kernel<CIN, COUT, IN_STATE, OUT_STATE, 4>;
            static std::atomic<unsigned> attr_done{0};
            if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)(C::LDS_BYTES), &attr_done)) return rc_;
            t.tiles_y = (a.Hout + C::TH - 1) / C::TH; t.tiles_total = t.tiles_x * t.tiles_y * B;
            VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
            kern<<<dim3((t.tiles_total + 7) / 8 * 8), C::NTHR, C::LDS_BYTES, st>>>(t);
#if VST_WITH_PINGPONG
        } else if (g_opt_pp.load(std::memory_order_relaxed)) {
            using C = PipeCfg<CIN, COUT>;
            auto kern = conv_pp_kernel<CIN, COUT, IN_STATE, OUT_STATE>;
            constexpr int lds = C::LDS_BYTES + (OUT_STATE ? COUT * 4 : 0);       // (+ the bias of the sliced kernel)
            static std::atomic<unsigned> attr_done{0};
            if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, lds, &attr_done)) return rc_;
            t.tiles_y = (a.Hout + C::TH - 1) / C::TH; t.tiles_total = t.tiles_x * t.tiles_y * B;
            VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
            kern<<<dim3((t.tiles_total + 7) / 8 * 8), C::NTHR, lds, st>>>(t);
#endif
        } else {
            using C = PipeCfg<CIN, COUT>;
            auto kern = conv_pipe_kernel<CIN, COUT, IN_STATE, OUT_STATE>;
            static std::atomic<unsigned> attr_done{0};
            if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)(C::LDS_BYTES), &attr_done)) return rc_;
            t.tiles_y = (a.Hout + C::TH - 1) / C::TH; t.tiles_total = t.tiles_x * t.tiles_y * B;
            VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
            kern<<<dim3((t.tiles_total + 7) / 8 * 8), C::NTHR, C::LDS_BYTES, st>>>(t);
        }
    } else {
        using C = ConvCfg<CIN, COUT, STRIDE>;
        // f16x2: the convs of the 16- and 64-channel blocks run the 2-term fp16 product as well (one weight plane in LDS: one
        // more workgroup per CU, a third fewer MFMAs)
        constexpr bool T2_SHAPE = (CIN == 64 && COUT == 16) || (CIN == 16 && COUT == 16 && STRIDE == 2) || (CIN == 16 && COUT == 4);
        const bool t2 = T2_SHAPE && vst_is_f16(precision);
        constexpr bool H16_SHAPE = T2_SHAPE && !OUT_STATE;      // an h1 intermediate of a 16- / 64-channel block
        const bool h16 = t2 && out_h16 && H16_SHAPE;
        auto kern = h16 ? conv_mfma_kernel<CIN, COUT, STRIDE, IN_STATE, OUT_STATE, false, T2_SHAPE ? 2 : 3, H16_SHAPE>
                  : t2 ? conv_mfma_kernel<CIN, COUT, STRIDE, IN_STATE, OUT_STATE, false, T2_SHAPE ? 2 : 3>
                       : conv_mfma_kernel<CIN, COUT, STRIDE, IN_STATE, OUT_STATE, false, 3>;
        const int lds = t2 ? C::LDS_BYTES_T2 : C::LDS_BYTES;
        static std::atomic<unsigned> attr_done[3];
        if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, lds, &attr_done[h16 ? 2 : (t2 ? 1 : 0)])) return rc_;
        ConvArgs t = a;
        t.tiles_x = (a.Wout + C::TW - 1) / C::TW; t.tiles_y = (a.Hout + C::TH - 1) / C::TH;
        t.tiles_total = t.tiles_x * t.tiles_y * B * C::NCOT;
        VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
        kern<<<dim3((t.tiles_total + 7) / 8 * 8), 256, lds, st>>>(t);
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

// conv.1 of the stride-2 256-channel block with its output written as split planes (the F16X2 path)
static int launch_conv_s2_planes(const ConvArgs& a, int B, hipStream_t st) {
    using C = ConvCfg<64, 64, 2>;
    vst_prof_scope prof(VST_KERNEL_ID(64, 64, 2), st);
    auto kern = conv_mfma_kernel<64, 64, 2, true, false, true, 2>;      // (the f16x2 path: 2-term fp16 like the rest of the block)
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)(C::LDS_BYTES_T2), &attr_done)) return rc_;
    ConvArgs t = a;
    t.tiles_x = (a.Wout + C::TW - 1) / C::TW; t.tiles_y = (a.Hout + C::TH - 1) / C::TH;
    t.tiles_total = t.tiles_x * t.tiles_y * B * C::NCOT;
    VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
    kern<<<dim3((t.tiles_total + 7) / 8 * 8), 256, C::LDS_BYTES_T2, st>>>(t);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

#ifndef VST_PAIR
#define VST_PAIR 1
#endif
#ifndef VST_PAIR_MR2_T3
#define VST_PAIR_MR2_T3 0
#endif
#ifndef VST_PAIR_MR2
#define VST_PAIR_MR2 0       // 1: the 64-channel 2-term pair on 8 x 16 tiles (measured, not kept: DESIGN.md)
#endif
template <int MID, int CH>
static int launch_pair(const ConvArgs& a, int B, int precision, hipStream_t st) {
    constexpr int T2 = 2;                                    // f16x2: the pair runs the 2-term fp16 product
    const bool t2 = T2 == 2 && vst_is_f16(precision);
    if (precision == VST_PREC_F16X2H) {                      // h1 arrives as fp16 (launch_conv(..., out_h16 = true) wrote it)
        using P = PairCfg<MID, CH, 2, 4, true>;
        vst_prof_scope prof(VST_KERNEL_ID(MID, CH, 1), st);
        auto kern = conv_pair_kernel<MID, CH, 2, 4, true>;
        static std::atomic<unsigned> attr_h16{0};
        if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, P::LDS_BYTES, &attr_h16)) return rc_;
        ConvArgs t = a;
        t.tiles_x = (a.Wout + 15) / 16; t.tiles_y = (a.Hout + 15) / 16; t.tiles_total = t.tiles_x * t.tiles_y * B;
        VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
        kern<<<dim3((t.tiles_total + 7) / 8 * 8), 256, P::LDS_BYTES, st>>>(t);
        VST_RETURN_IF_LAUNCH_FAILED();
        return VST_OK;
    }
    vst_prof_scope prof(VST_KERNEL_ID(MID, CH, 1), st);
    // (-DVST_PAIR_MR2=1: the 64-channel blocks' 2-term pair on 8 x 16 tiles - 33 KB of LDS, 96 VGPRs, four workgroups per CU,
    // eight per CU and launch at 1024 x 1024 in two even rounds instead of 3 + 1: 7 % faster alone, 0.6 % slower in the frame)
    constexpr int MRT = (MID == 16 && VST_PAIR_MR2) ? 2 : 4;
    constexpr int MRT3 = (MID == 16 && VST_PAIR_MR2_T3) ? 2 : 4;        // the same experiment for the bf16 3-term pair
    auto kern = t2 ? conv_pair_kernel<MID, CH, T2, MRT> : conv_pair_kernel<MID, CH, 3, MRT3>;
    const int lds = t2 ? PairCfg<MID, CH, T2, MRT>::LDS_BYTES : PairCfg<MID, CH, 3, MRT3>::LDS_BYTES;
    const int th = t2 ? 4 * MRT : 4 * MRT3;
    static std::atomic<unsigned> attr_done[2];
    if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, lds, &attr_done[t2 ? 1 : 0])) return rc_;
    ConvArgs t = a;
    t.tiles_x = (a.Wout + 15) / 16; t.tiles_y = (a.Hout + th - 1) / th; t.tiles_total = t.tiles_x * t.tiles_y * B;
    VST_TRACE_RESERVE(t, (t.tiles_total + 7) / 8 * 8)
    kern<<<dim3((t.tiles_total + 7) / 8 * 8), 256, lds, st>>>(t);
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
    if constexpr (CH == 256 && STRIDE == 2) {
        if (vst_is_f16(precision)) {
            // the F16X2 path: h1 / h2 as split fp16 planes, conv.4 and conv.7 on conv3.hip's kernels.  In a forward pass the
            // new state half also goes out as planes: it is the src of the run of stride-1 blocks that follows.
            const size_t mid_bytes = (size_t)Ho * Wo * 64 * 4, state_bytes = (size_t)Ho * Wo * 256 * 4;
            unsigned char* const h1p = (unsigned char*)tmp;
            unsigned char* const h2p = h1p + (size_t)B * mid_bytes;
            unsigned char* const planes_a = h2p + (size_t)B * mid_bytes;
            (void)state_bytes;
            a.out_sp = h1p; a.out_sp_img_bytes = mid_bytes;
            int rc2 = launch_conv_s2_planes(a, B, st);
            if (rc2) return rc2;
            const int h2_single = precision == VST_PREC_F16X2H;
            rc2 = vst3_conv_mid(&w->conv[1], h1p, h2p, h2_single, B, H, W, st);
            if (rc2) return rc2;
            return vst3_conv_out(&w->conv[2], h2p, h2_single, dst, direction > 0 ? planes_a : nullptr, direction > 0 ? 1.f : -1.f,
                                 B, H, W, st);
        }
    }
    // f16x2h: h1 of the 16- / 64-channel blocks goes through HBM as fp16 (the pair kernel that reads it is chosen by the same test)
    const bool h16 = CH <= 64 && VST_PAIR && precision == VST_PREC_F16X2H;
    int rc = launch_conv<IN_CH, MID, STRIDE, true, false>(a, B, precision, st, h16);
    if (rc) return rc;
    if constexpr (CH <= 64 && VST_PAIR) {
        if (precision != VST_PREC_FP32) {         // conv.4 + conv.7 in one launch, h2 stays in LDS
            a.in = h1; a.out = dst; a.Hin = Ho; a.Win = Wo; a.in_img_stride = mid_img; a.out_img_stride = state_img;
            a.packed = (const unsigned char*)w->conv[2].packed; a.bias = w->conv[2].bias;
            a.packed1 = (const unsigned char*)w->conv[1].packed; a.bias1 = w->conv[1].bias;
            a.sign = direction > 0 ? 1.f : -1.f;
            return launch_pair<MID, CH>(a, B, precision, st);
        }
    }
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

VST_DEFINE_TU_RANGE(vst_range_tu_conv)

extern "C" {

int vst_set_option(int option, int value) {
    if (option == VST_OPT_STAGE3_LEAN) g_opt_lean.store(value != 0, std::memory_order_relaxed);
    else if (option == VST_OPT_STAGE3_PINGPONG && VST_WITH_PINGPONG) g_opt_pp.store(value != 0, std::memory_order_relaxed);
    else return VST_E_ARG;
    return VST_OK;
}

int vst_get_option(int option) {
    if (option == VST_OPT_STAGE3_LEAN) return g_opt_lean.load(std::memory_order_relaxed);
    if (option == VST_OPT_STAGE3_PINGPONG && VST_WITH_PINGPONG) return g_opt_pp.load(std::memory_order_relaxed);
    return VST_E_ARG;
}

int vst_range_flags(unsigned* flags_host, int reset) {
    if (!flags_host) return VST_E_ARG;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    unsigned v = 0;
    if (int rc = vst_range_tu_conv(&v, reset)) return rc;
    if (int rc = vst_range_tu_conv3(&v, reset)) return rc;
    if (int rc = vst_range_tu_layout(&v, reset)) return rc;
    if (int rc = vst_range_tu_cwct(&v, reset)) return rc;
    *flags_host = v;
    return VST_OK;
}

int vst_range_flags_async(unsigned* flags4_dev, void* stream) {
    if (!flags4_dev) return VST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = vst_range_tu_conv_async(flags4_dev + 0, st)) return rc;
    if (int rc = vst_range_tu_conv3_async(flags4_dev + 1, st)) return rc;
    if (int rc = vst_range_tu_layout_async(flags4_dev + 2, st)) return rc;
    return vst_range_tu_cwct_async(flags4_dev + 3, st);
}

int vst_profile_begin(int kernel_id, int max_records) {
    if ((kernel_id <= 0 && kernel_id != VST_KERNEL_ALL) || max_records <= 0) return VST_E_ARG;
    if (max_records > VST_PROFILE_MAX_RECORDS) max_records = VST_PROFILE_MAX_RECORDS;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_ev_created) {
        for (int i = 0; i < 2 * VST_PROFILE_MAX_RECORDS; ++i) {
            hipError_t e = hipEventCreate(&g_prof_ev[i]);
            if (e != hipSuccess) return (int)e;
        }
        g_prof_ev_created = true;
    }
    g_prof_count = 0; g_prof_cap = max_records;
    g_prof_kernel.store(kernel_id);
    return VST_OK;
}

int vst_profile_end(double* total_ms, int* launches) {
    if (!total_ms || !launches) return VST_E_ARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_kernel.store(0);
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
    g_prof_count = 0; g_prof_cap = 0;
    return VST_OK;
}

// h1 + h2 (8 floats per pixel) + the split planes of both state halves for the stage-3 kernels of conv3.hip (2 x 16)
int vst_profile_end_table(int* ids, double* ms, int* launches, int cap, int* n_ids) {
    if (!ids || !ms || !launches || !n_ids || cap <= 0) return VST_E_ARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_kernel.store(0);
    int n = 0;
    for (int i = 0; i < g_prof_count; ++i) {
        hipError_t e = hipEventSynchronize(g_prof_ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        float t = 0.f;
        e = hipEventElapsedTime(&t, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        int k = 0;
        while (k < n && ids[k] != g_prof_id[i]) ++k;
        if (k == n) {
            if (n == cap) continue;
            ids[n] = g_prof_id[i]; ms[n] = 0.0; launches[n] = 0; ++n;
        }
        ms[k] += t; launches[k] += 1;
    }
    *n_ids = n;
    g_prof_count = 0; g_prof_cap = 0;
    return VST_OK;
}

size_t vst_block_tmp_bytes(int B, int H, int W) { return (size_t)B * H * W * 40 * sizeof(float); }

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
    if (channel == 256 && stride == 1) {
        if (vst_is_f16(precision))
#if defined(VST_SP_ABLATE) && (VST_SP_ABLATE & 8)
            return vst3_block256(w, direction, precision, dst, src, tmp, 5, 0, B, H, W, stream);   // diagnostic build: a mid-run block
#else
            return vst3_block256(w, direction, precision, dst, src, tmp, -1, 0, B, H, W, stream);
#endif
        return run_block<256, 1>(w, direction, precision, dst, src, t, B, H, W, st);
    }
    if (channel == 256 && stride == 2) return run_block<256, 2>(w, direction, precision, dst, src, t, B, H, W, st);
    return VST_E_SHAPE;
}

size_t vst_pass_workspace_bytes(int B, int H, int W) { return (size_t)B * H * W * (16 + 16 + 40) * sizeof(float); }

// Images per internal sub-batch: the reversible state + intermediates of a sub-batch (160 B/pixel) should stay in
// the 256 MiB Infinity Cache between the 96 conv launches of a pass (measured: 8 frames of 1024x1024 in one batch
// run 27 % slower per frame than one at a time); small images are still batched to fill the chip.
static int pass_sub_batch(int B, int H, int W) {
    const size_t per_img = (size_t)H * W * 288;
    size_t nb = ((size_t)192 << 20) / per_img;
    if (nb < 1) nb = 1;
    return nb > (size_t)B ? B : (int)nb;
}

// the 32 coupling blocks of a forward pass on the state halves s[0], s[1] (n images each), input packing included
static int forward_blocks(const vst_net_weights* w, const float* x, const uint8_t* x_u8, float* const s[2], float* tmp, int B,
                          int C_in, int H, int W, int precision, void* stream) {
    // forward block 0 has x2 = 0: F(0) is a per-channel constant that the pack kernel adds (fp32 diagnostic mode keeps
    // the literal three convolutions)
    const bool fold0 = precision != VST_PREC_FP32;
    float* k16 = tmp;                                        // 16 floats at the head of the intermediates' scratch
    int rc = fold0 ? vst_block0_const(&w->blocks[0], k16, stream) : VST_OK;
    if (rc) return rc;
    rc = vst_pack_input_k(x, x_u8, s[0], s[1], B, x_u8 ? 3 : C_in, H, W, fold0 ? k16 : nullptr, stream);
    if (rc) return rc;
    const bool sp = vst_is_f16(precision);
    for (int k = fold0 ? 1 : 0; k < VST_NUM_BLOCKS; ++k) {
        if (sp && k >= 21)      // block k's conv.7 leaves the split planes of its dst = block k+1's src
            rc = vst3_block256(&w->blocks[k], +1, precision, s[k & 1], s[1 - (k & 1)], tmp, k - 21, 1, B, H, W, stream);
        else
            rc = vst_block_apply(&w->blocks[k], kBlockChannel[k], kBlockStride[k], +1, precision, s[k & 1], s[1 - (k & 1)],
                                 tmp, B, H, W, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

static int revnet_forward_chunk(const vst_net_weights* w, const float* x, const uint8_t* x_u8, float* z, void* workspace,
                                int B, int C_in, int H, int W, int sp_steps, int precision, void* stream) {
    float* s[2];
    s[0] = (float*)workspace;
    s[1] = s[0] + (size_t)B * H * W * 16;
    float* tmp = s[1] + (size_t)B * H * W * 16;
    const int rc = forward_blocks(w, x, x_u8, s, tmp, B, C_in, H, W, precision, stream);
    if (rc) return rc;
    return vst_spread(s[0], s[1], z, B, H, W, sp_steps, stream);
}

// the 32 coupling blocks of an inverse pass on the state halves (f16x2: s[0] is given as split planes in plane buffer 0 of
// tmp - block 31 reads its src only through them and block 30 takes its old values from them too), output unpacking included
static int inverse_blocks(const vst_net_weights* w, float* x, uint8_t* x_u8, float* const s[2], float* tmp, int B, int C_out,
                          int H, int W, int precision, void* stream) {
    const bool sp = vst_is_f16(precision);
    int rc = VST_OK;
    for (int k = VST_NUM_BLOCKS - 1; k >= 0; --k) {
        if (sp && k >= 21)
            rc = vst3_block256(&w->blocks[k], -1, precision, s[k & 1], s[1 - (k & 1)], tmp, VST_NUM_BLOCKS - 1 - k, 1, B, H, W,
                               stream);
        else
            rc = vst_block_apply(&w->blocks[k], kBlockChannel[k], kBlockStride[k], -1, precision, s[k & 1], s[1 - (k & 1)],
                                 tmp, B, H, W, stream);
        if (rc) return rc;
    }
    return x_u8 ? vst_unpack_output_u8(s[0], x_u8, B, H, W, stream) : vst_unpack_output(s[0], x, B, C_out, H, W, stream);
}

static int revnet_inverse_chunk(const vst_net_weights* w, const float* z, float* x, uint8_t* x_u8, void* workspace, int B,
                                int C_out, int H, int W, int sp_steps, int precision, void* stream) {
    float* s[2];
    s[0] = (float*)workspace;
    s[1] = s[0] + (size_t)B * H * W * 16;
    float* tmp = s[1] + (size_t)B * H * W * 16;
    // f16x2: the gather writes s[0] straight into plane buffer 0 (no fp32 copy, no pre-split pass)
    const bool sp = vst_is_f16(precision);
    const int rc = sp ? vst3_gather_planes(z, vst3_plane_buffer(tmp, 0, B, H, W), s[1], B, H, W, sp_steps, stream)
                      : vst_gather(z, s[0], s[1], B, H, W, sp_steps, stream);
    if (rc) return rc;
    return inverse_blocks(w, x, x_u8, s, tmp, B, C_out, H, W, precision, stream);
}

// Packed code (photorealistic mode): the state halves themselves, per image [2][H/4][W/4][256] floats = one 32-float row per
// full-resolution pixel.  encode = forward pass without the spread, one image at a time (its halves are the pass's state
// buffers); decode = [affine map of an unmasked cWCT on the rows ->] inverse pass without the gather.
static int revnet_encode_any(const vst_net_weights* w, const float* x, const uint8_t* x_u8, float* code, void* workspace, int B,
                             int C_in, int H, int W, int precision, void* stream) {
    if (!w || (!x && !x_u8) || !code) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(B, H, W) || C_in < 1 || C_in > 16) return VST_E_SHAPE;
    const size_t img = (size_t)32 * H * W;
    float* tmp = (float*)workspace + img;                    // the scratch part of a one-image pass workspace
    for (int b = 0; b < B; ++b) {
        float* s[2] = {code + b * img, code + b * img + img / 2};
        const int rc = forward_blocks(w, x ? x + (size_t)b * C_in * H * W : nullptr, x_u8 ? x_u8 + (size_t)b * H * W * 3 : nullptr,
                                      s, tmp, 1, C_in, H, W, precision, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

// one image whose cWCT is a masked one: a map per row (label slot), cwct.hip: vst3_apply_labels_code
static int revnet_decode_labels_any(const vst_net_weights* w, const float* code, const float* affines, const uint8_t* mask_rows,
                                    const void* plan, int max_slots, float* x, uint8_t* x_u8, void* workspace, int C_out, int H,
                                    int W, int precision, void* stream) {
    if (!w || (!x && !x_u8) || !code || !affines || !mask_rows || !plan) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(1, H, W) || C_out < 1 || C_out > 16 || max_slots < 1 || max_slots > 8) return VST_E_SHAPE;
    const size_t img = (size_t)32 * H * W;
    float* s[2] = {(float*)workspace, (float*)workspace + img / 2};
    float* tmp = (float*)workspace + img;
    unsigned char* planes0 = vst_is_f16(precision) ? vst3_plane_buffer(tmp, 0, 1, H, W) : nullptr;
    int rc = vst3_apply_labels_code(code, s[0], s[1], planes0, H, W, affines, mask_rows, plan, max_slots, stream);
    if (rc) return rc;
    return inverse_blocks(w, x, x_u8, s, tmp, 1, C_out, H, W, precision, stream);
}

static int revnet_decode_any(const vst_net_weights* w, const float* code, const float* affines, float* x, uint8_t* x_u8,
                             void* workspace, int B, int C_out, int H, int W, int sp_steps, int precision, void* stream) {
    if (!w || (!x && !x_u8) || !code) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(B, H, W) || C_out < 1 || C_out > 16) return VST_E_SHAPE;
    if (sp_steps != 1 && sp_steps != 2) return VST_E_MODE;
    const int N = sp_steps == 2 ? 32 : 128;
    const size_t img = (size_t)32 * H * W;
    float* s[2] = {(float*)workspace, (float*)workspace + img / 2};
    float* tmp = (float*)workspace + img;
    const bool sp = vst_is_f16(precision);
    unsigned char* planes0 = sp ? vst3_plane_buffer(tmp, 0, 1, H, W) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < B; ++b) {
        const float* c = code + b * img;
        int rc;
        if (affines) {
            rc = vst3_apply_code(c, s[0], s[1], planes0, H, W, sp_steps, affines + (size_t)b * ((size_t)N * N + N), stream);
        } else {                                             // plain copy into the pass's state (it is updated in place)
            rc = sp ? vst3_presplit(c, planes0, 1, H, W, stream)
                    : (int)hipMemcpyAsync(s[0], c, img / 2 * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (!rc) rc = (int)hipMemcpyAsync(s[1], c + img / 2, img / 2 * sizeof(float), hipMemcpyDeviceToDevice, st);
        }
        if (rc) return rc;
        rc = inverse_blocks(w, x ? x + (size_t)b * C_out * H * W : nullptr, x_u8 ? x_u8 + (size_t)b * H * W * 3 : nullptr, s, tmp,
                            1, C_out, H, W, precision, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

extern "C" int vst_pass_sub_batch(int B, int H, int W) {
    if (!vst_shape_ok(B, H, W)) return VST_E_SHAPE;
    return pass_sub_batch(B, H, W);
}

static int revnet_forward_any(const vst_net_weights* w, const float* x, const uint8_t* x_u8, float* z, void* workspace,
                              int B, int C_in, int H, int W, int sp_steps, int precision, void* stream) {
    if (!w || (!x && !x_u8) || !z) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(B, H, W) || C_in < 1 || C_in > 16) return VST_E_SHAPE;
    if (sp_steps != 1 && sp_steps != 2) return VST_E_MODE;
    const int nb = pass_sub_batch(B, H, W);
    const size_t zimg = (size_t)32 * H * W;                 // floats per image of z in both modes
    for (int b0 = 0; b0 < B; b0 += nb) {
        const int n = B - b0 < nb ? B - b0 : nb;
        const int rc = revnet_forward_chunk(w, x ? x + (size_t)b0 * C_in * H * W : nullptr,
                                            x_u8 ? x_u8 + (size_t)b0 * H * W * 3 : nullptr, z + (size_t)b0 * zimg,
                                            workspace, n, C_in, H, W, sp_steps, precision, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

static int revnet_inverse_any(const vst_net_weights* w, const float* z, float* x, uint8_t* x_u8, void* workspace, int B,
                              int C_out, int H, int W, int sp_steps, int precision, void* stream) {
    if (!w || (!x && !x_u8) || !z) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!vst_shape_ok(B, H, W) || C_out < 1 || C_out > 16) return VST_E_SHAPE;
    if (sp_steps != 1 && sp_steps != 2) return VST_E_MODE;
    const int nb = pass_sub_batch(B, H, W);
    const size_t zimg = (size_t)32 * H * W;
    for (int b0 = 0; b0 < B; b0 += nb) {
        const int n = B - b0 < nb ? B - b0 : nb;
        const int rc = revnet_inverse_chunk(w, z + (size_t)b0 * zimg, x ? x + (size_t)b0 * C_out * H * W : nullptr,
                                            x_u8 ? x_u8 + (size_t)b0 * H * W * 3 : nullptr, workspace, n, C_out, H, W,
                                            sp_steps, precision, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

int vst_revnet_forward(const vst_net_weights* w, const float* x, float* z, void* workspace, int B, int C_in, int H,
                       int W, int sp_steps, int precision, void* stream) {
    if (!x) return VST_E_ARG;
    return revnet_forward_any(w, x, nullptr, z, workspace, B, C_in, H, W, sp_steps, precision, stream);
}

int vst_revnet_inverse(const vst_net_weights* w, const float* z, float* x, void* workspace, int B, int C_out, int H,
                       int W, int sp_steps, int precision, void* stream) {
    if (!x) return VST_E_ARG;
    return revnet_inverse_any(w, z, x, nullptr, workspace, B, C_out, H, W, sp_steps, precision, stream);
}

int vst_revnet_forward_u8(const vst_net_weights* w, const uint8_t* frames_hwc, float* z, void* workspace, int B, int H,
                          int W, int sp_steps, int precision, void* stream) {
    if (!frames_hwc) return VST_E_ARG;
    return revnet_forward_any(w, nullptr, frames_hwc, z, workspace, B, 3, H, W, sp_steps, precision, stream);
}

int vst_revnet_inverse_u8(const vst_net_weights* w, const float* z, uint8_t* frames_hwc, void* workspace, int B, int H,
                          int W, int sp_steps, int precision, void* stream) {
    if (!frames_hwc) return VST_E_ARG;
    return revnet_inverse_any(w, z, nullptr, frames_hwc, workspace, B, 3, H, W, sp_steps, precision, stream);
}

int vst_revnet_encode(const vst_net_weights* w, const float* x, float* code, void* workspace, int B, int C_in, int H, int W,
                      int precision, void* stream) {
    if (!x) return VST_E_ARG;
    return revnet_encode_any(w, x, nullptr, code, workspace, B, C_in, H, W, precision, stream);
}

int vst_revnet_encode_u8(const vst_net_weights* w, const uint8_t* frames_hwc, float* code, void* workspace, int B, int H, int W,
                         int precision, void* stream) {
    if (!frames_hwc) return VST_E_ARG;
    return revnet_encode_any(w, nullptr, frames_hwc, code, workspace, B, 3, H, W, precision, stream);
}

int vst_revnet_decode(const vst_net_weights* w, const float* code, const float* affines, float* x, void* workspace, int B,
                      int C_out, int H, int W, int sp_steps, int precision, void* stream) {
    if (!x) return VST_E_ARG;
    return revnet_decode_any(w, code, affines, x, nullptr, workspace, B, C_out, H, W, sp_steps, precision, stream);
}

int vst_revnet_decode_u8(const vst_net_weights* w, const float* code, const float* affines, uint8_t* frames_hwc, void* workspace,
                         int B, int H, int W, int sp_steps, int precision, void* stream) {
    if (!frames_hwc) return VST_E_ARG;
    return revnet_decode_any(w, code, affines, nullptr, frames_hwc, workspace, B, 3, H, W, sp_steps, precision, stream);
}

int vst_revnet_decode_labels(const vst_net_weights* w, const float* code, const float* affines, const uint8_t* mask_rows,
                             const void* plan, int max_slots, float* x, void* workspace, int C_out, int H, int W, int precision,
                             void* stream) {
    if (!x) return VST_E_ARG;
    return revnet_decode_labels_any(w, code, affines, mask_rows, plan, max_slots, x, nullptr, workspace, C_out, H, W, precision,
                                    stream);
}

int vst_revnet_decode_labels_u8(const vst_net_weights* w, const float* code, const float* affines, const uint8_t* mask_rows,
                                const void* plan, int max_slots, uint8_t* frame_hwc, void* workspace, int H, int W, int precision,
                                void* stream) {
    if (!frame_hwc) return VST_E_ARG;
    return revnet_decode_labels_any(w, code, affines, mask_rows, plan, max_slots, nullptr, frame_hwc, workspace, 3, H, W,
                                    precision, stream);
}

int vst_code_to_z(const float* code, float* z, int B, int H, int W, int sp_steps, void* stream) {
    if (!code || !z) return VST_E_ARG;
    const size_t img = (size_t)32 * H * W;
    for (int b = 0; b < B; ++b) {
        const int rc = vst_spread(code + b * img, code + b * img + img / 2, z + b * img, 1, H, W, sp_steps, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

int vst_z_to_code(const float* z, float* code, int B, int H, int W, int sp_steps, void* stream) {
    if (!code || !z) return VST_E_ARG;
    const size_t img = (size_t)32 * H * W;
    for (int b = 0; b < B; ++b) {
        const int rc = vst_gather(z + b * img, code + b * img, code + b * img + img / 2, 1, H, W, sp_steps, stream);
        if (rc) return rc;
    }
    return VST_OK;
}

}  // extern "C"

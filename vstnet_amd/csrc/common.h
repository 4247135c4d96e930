// Shared device/host helpers for libvstnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "vstnet.h"

#define VST_RETURN_IF_LAUNCH_FAILED()                      \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---------------------------------------------------------------------------------------------
// ZC state layout.  One half of the reversible state of an HxW image is [H/4][W/4][256] floats.
// A "view level" l in {0,1,2} sees it as an (H>>l)x(W>>l) image with 16*4^l channels, channels
// last, such that the reference's squeeze (models/RevResNet.py:34-37,
//   out[(i*2+j)*D+d, h, w] = in[d, 2h+i, 2w+j]) is the identity on memory.
// zc_offset returns the float offset of channel 0 of view pixel (y,x) inside one image.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int vst_level_of_channels(int c) { return c == 16 ? 0 : (c == 64 ? 1 : 2); }

__device__ __forceinline__ size_t zc_offset(int level, int y, int x, int Wq) {
    if (level == 0) {
        const int cell = (y >> 2) * Wq + (x >> 2);
        const int sub = ((((y >> 1) & 1) * 2 + ((x >> 1) & 1)) << 6) + (((y & 1) * 2 + (x & 1)) << 4);
        return (size_t)cell * 256 + sub;
    } else if (level == 1) {
        const int cell = (y >> 1) * Wq + (x >> 1);
        return (size_t)cell * 256 + (((y & 1) * 2 + (x & 1)) << 6);
    }
    return (size_t)(y * Wq + x) * 256;
}

// ReflectionPad2d(1) index for a coordinate that may be one past either end, clamped for
// coordinates further out (those only feed outputs that are never stored).
__device__ __forceinline__ int reflect_clamp(int v, int n) {
    if (v < 0) v = -v;
    if (v >= n) v = 2 * n - 2 - v;
    v = v < 0 ? 0 : v;
    return v >= n ? n - 1 : v;
}

// ---------------------------------------------------------------------------------------------
// Split fp16 planes ("SP") of the 256-channel blocks (conv3.hip): [8-channel group][plane][y][x][8 x fp16] at quarter
// resolution, hi = round(x) (saturating at the largest finite fp16), lo = round(x - hi).
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// fp16 range flags (VST_PREC_F16X2 / VST_PREC_F16X2H): every place that rounds an ACTIVATION to fp16 clamps it to the largest
// finite fp16 and, if anything was clamped (|x| > 65504, Inf), raises VST_RANGE_SATURATED in this word; vst_pack_conv raises
// VST_RANGE_WEIGHT for a weight that does not fit.  One word per translation unit (the library is built without relocatable
// device code), OR-ed together by vst_range_flags (conv.hip).  Read by calibration (RevResNet.check_range), bench.py and tests.
static __device__ unsigned vst_tu_range_flags;
__device__ __forceinline__ void vst_note_range(float amax) {
#ifndef VST_NO_RANGE_CHECK           // (timing-only A/B builds of tools/ab_build.py; the shipped library always checks)
    if (amax > 65504.f) atomicOr(&vst_tu_range_flags, VST_RANGE_SATURATED);
#endif
}
#define VST_DEFINE_TU_RANGE(name)                                                                                    \
    __attribute__((visibility("hidden"))) int name(unsigned* acc, int reset) {                                       \
        unsigned v = 0;                                                                                              \
        hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(vst_tu_range_flags), sizeof(v));                           \
        if (e != hipSuccess) return (int)e;                                                                          \
        *acc |= v;                                                                                                   \
        if (reset && v) {                                                                                            \
            const unsigned zero = 0;                                                                                 \
            e = hipMemcpyToSymbol(HIP_SYMBOL(vst_tu_range_flags), &zero, sizeof(zero));                              \
            if (e != hipSuccess) return (int)e;                                                                      \
        }                                                                                                            \
        return VST_OK;                                                                                               \
    }                                                                                                                \
    __attribute__((visibility("hidden"))) int name##_async(unsigned* dst_dev, hipStream_t st) {                      \
        return (int)hipMemcpyFromSymbolAsync(dst_dev, HIP_SYMBOL(vst_tu_range_flags), sizeof(unsigned), 0,           \
                                             hipMemcpyDeviceToDevice, st);                                           \
    }

// `amax`: the caller's running max |x| over everything it rounds to fp16 (v_max3 only: no compare, no branch in the hot loops);
// the caller hands it to vst_note_range ONCE, at its end.  The overload without it checks on the spot.
__device__ __forceinline__ void split8_sp(const float (&f)[8], u32x4& hi, u32x4& lo, float& amax) {
    f16x8 h, l;
#ifndef VST_NO_RANGE_CHECK
    amax = fmaxf(fmaxf(fmaxf(amax, fabsf(f[0])), fmaxf(fabsf(f[1]), fabsf(f[2]))),
                 fmaxf(fmaxf(fabsf(f[3]), fabsf(f[4])), fmaxf(fmaxf(fabsf(f[5]), fabsf(f[6])), fabsf(f[7]))));
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (_Float16)__builtin_amdgcn_fmed3f(f[i], -65504.f, 65504.f);
        l[i] = (_Float16)(f[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}
__device__ __forceinline__ void split8_sp(const float (&f)[8], u32x4& hi, u32x4& lo) {
    float amax = 0.f;
    split8_sp(f, hi, lo, amax);
    vst_note_range(amax);
}

// byte offset of (8-channel group, plane, y, x) in an SP tensor of an HxW image (32 bits: one image's planes are at most
// 4 * 256 * H * W bytes = 1 GiB at the 4096 x 4096 frame's quarter resolution; the image index goes into the 64-bit base)
__device__ __forceinline__ unsigned sp_offset(int cig, int plane, int y, int x, int H, int W) {
    return ((((unsigned)cig * 2 + plane) * H + y) * W + x) * 16u;
}
// the same for a tensor kept as ONE fp16 plane (VST_PREC_F16X2H: h2)
__device__ __forceinline__ unsigned sp_offset1(int cig, int y, int x, int H, int W) {
    return (((unsigned)cig * H + y) * W + x) * 16u;
}
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: set it once per (kernel, device), not once per process
// (the mask is atomic: entry points may be called from several host threads, one stream each; setting the attribute
// twice is harmless)
static inline int vst_ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned>* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 32 && (done_mask->load(std::memory_order_acquire) >> dev) & 1u) return VST_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev < 32) done_mask->fetch_or(1u << dev, std::memory_order_release);
    return VST_OK;
}

static inline bool vst_shape_ok(int B, int H, int W) {
    return B > 0 && H >= 8 && W >= 8 && (H % 4) == 0 && (W % 4) == 0;
}

// packed conv weights: [fp32 taps-major | bf16 hi frags | bf16 lo frags | (stage-3 shapes, cin, cout >= 64: the fp16 fragments
// of conv3.hip in its permuted K order) | fp16 frags in the bf16 sections' order]
struct PackedConvLayout {
    int ksteps;       // number of 32-deep K steps (all chunks)
    int coutp;        // cout padded to a multiple of 16
    size_t f32_bytes; // 9*cin*cout*4 rounded up to 256
    size_t frag_bytes;// ksteps*4*coutp*16 (per section)
    int sp_sections;  // 1 for the stage-3 shapes, else 0
    size_t f16_offset;// byte offset of the fp16 fragments in the bf16 sections' order (the 2-term kernels of conv.hip)
};

__host__ __device__ inline PackedConvLayout packed_conv_layout(int cout, int cin) {
    PackedConvLayout p;
    p.ksteps = cin >= 32 ? 9 * (cin / 32) : (cin == 16 ? 5 : 2);
    p.coutp = (cout + 15) / 16 * 16;
    p.f32_bytes = ((size_t)9 * cin * cout * 4 + 255) / 256 * 256;
    p.frag_bytes = (size_t)p.ksteps * 4 * p.coutp * 16;
    p.sp_sections = (cin >= 64 && cout >= 64) ? 1 : 0;
    p.f16_offset = p.f32_bytes + (size_t)(2 + p.sp_sections) * p.frag_bytes;
    return p;
}

// HIP-event timing of one kernel class (vst_profile_begin / vst_profile_end): a launch site opens a scope around its
// launch; sessions and records are serialised by a lock inside conv.hip
int vst_prof_open(int kernel_id, hipStream_t st);
void vst_prof_close(int rec, hipStream_t st);
struct vst_prof_scope {
    int rec;
    hipStream_t st;
    vst_prof_scope(int kernel_id, hipStream_t s) : rec(vst_prof_open(kernel_id, s)), st(s) {}
    ~vst_prof_scope() { if (rec >= 0) vst_prof_close(rec, st); }
};

// conv3.hip: one 256-channel stride-1 coupling block on the LDS-DMA kernels.  tmp = [h1 | h2 | planes A | planes B]
// (vst_block_tmp_bytes).  pos = position 0..10 of the block in a pass's run of eleven such blocks (the state then lives in the
// split planes between the run's ends), or -1 for a block on its own (fp32 state in, fp32 state out).  src_planes_ready: the
// planes of the run's first src are already in planes A (written by block 20's conv.7 in a forward pass).
int vst3_block256(const vst_block_weights* w, int direction, int precision, float* dst, const float* src, void* tmp,
                  int pos, int src_planes_ready, int B, int H, int W, void* stream);
// the stride-2 256-channel block's conv.4 (h1 planes -> h2 planes) and conv.7 (h2 planes -> fp32 state read-modify-write,
// optionally also the new state's planes into planes A) on the same kernels; conv.1 is conv.hip's stride-2 kernel writing planes
// (out_single / in_single: h2 as one fp16 plane, VST_PREC_F16X2H)
int vst3_conv_mid(const vst_conv_weights* c, const void* in_sp, void* out_sp, int out_single, int B, int H, int W, void* stream);
int vst3_conv_out(const vst_conv_weights* c, const void* in_sp, int in_single, float* state, void* out_sp, float sign, int B,
                  int H, int W, void* stream);
static inline bool vst_is_f16(int precision) { return precision == VST_PREC_F16X2 || precision == VST_PREC_F16X2H; }

// the split-plane buffer idx (0 = A, 1 = B) inside tmp; layout.hip: gather whose first half goes straight into split planes
unsigned char* vst3_plane_buffer(void* tmp, int idx, int B, int H, int W);
extern "C" __attribute__((visibility("hidden"))) int vst3_gather_planes(const float* z, unsigned char* s1_planes, float* s2, int B, int H, int W, int sp_steps,
                                  void* stream);

int vst3_presplit(const float* state, unsigned char* planes, int B, int H, int W, void* stream);
// cwct.hip: y = T x + t0 on the rows of one image's packed code; half 0 to out0 or (planes0 != nullptr) to split planes
int vst3_apply_code(const float* code, float* out0, float* out1, unsigned char* planes0, int H, int W, int sp_steps,
                    const float* affine, void* stream);

int vst3_apply_labels_code(const float* code, float* out0, float* out1, unsigned char* planes0, int H, int W,
                           const float* affines, const uint8_t* mask_rows, const void* plan, int max_slots, void* stream);

// internal (not part of the C ABI, hidden in the shared library): input packing with the constant of forward block 0 folded in
#define VST_INTERNAL __attribute__((visibility("hidden")))
int vst_range_tu_conv(unsigned* acc, int reset);
int vst_range_tu_conv3(unsigned* acc, int reset);
int vst_range_tu_layout(unsigned* acc, int reset);
int vst_range_tu_cwct(unsigned* acc, int reset);
int vst_range_tu_conv_async(unsigned* dst_dev, hipStream_t st);
int vst_range_tu_conv3_async(unsigned* dst_dev, hipStream_t st);
int vst_range_tu_layout_async(unsigned* dst_dev, hipStream_t st);
int vst_range_tu_cwct_async(unsigned* dst_dev, hipStream_t st);
extern "C" VST_INTERNAL int vst_pack_input_k(const float* x, const uint8_t* x_u8, float* s1, float* s2, int B, int C, int H, int W,
                                const float* addk, void* stream);
extern "C" VST_INTERNAL int vst_block0_const(const vst_block_weights* w0, float* k16, void* stream);

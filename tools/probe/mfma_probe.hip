// Synthetic MFMA-bound workgroup for tools/overlap_probe.py: the resource footprint (waves, VGPRs, LDS) and the
// LDS-read : MFMA ratio of a stage-3 conv workgroup, nothing else.  Diagnostic only; not part of libvstnet_hip.so.
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(NT) void mfma_probe_kernel(int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384 / 4; i += NT) ((float*)smem)[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* base = smem + lane * 16 + wave * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f16x8 w[4], xh[2], xl[2];
            const unsigned char* p = base + ((it + k) & 1) * 8192;
#pragma unroll
            for (int n = 0; n < 4; ++n) w[n] = *(const f16x8*)(p + n * 1024 % 8192);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xh[m] = *(const f16x8*)(p + (4 + m) * 1024 % 8192);
                xl[m] = *(const f16x8*)(p + (6 + m) * 1024 % 8192);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[n], xl[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[n], xh[m], acc[m][n], 0, 0, 0);
                }
        }
        if ((it & 1) == 1) __builtin_amdgcn_s_barrier();      // a stage boundary every 18 k-steps ~ 2 chunks
    }
    asm volatile("v_mov_b32 v190, 0" ::: "v190");             // reserve the register footprint of the real kernel
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (s == 123.456f) sink[0] = s;
}

extern "C" int mfma_probe(int waves, int lds_bytes, int iters, int grid, float* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (waves == 4) {
        (void)hipFuncSetAttribute((const void*)mfma_probe_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        mfma_probe_kernel<256><<<grid, 256, lds_bytes, st>>>(iters, sink);
    } else {
        (void)hipFuncSetAttribute((const void*)mfma_probe_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        mfma_probe_kernel<512><<<grid, 512, lds_bytes, st>>>(iters, sink);
    }
    return (int)hipGetLastError();
}

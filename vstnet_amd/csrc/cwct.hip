// cWCT — Cholesky whitening / colouring transform (reference: models/cWCT.py).
//
//   whitening  cWCT.py:134-149   mu = mean_L(x); Xc = x - mu; C = Xc Xc^T/(L-1); Xw = inv(chol(C)) Xc
//   coloring   cWCT.py:152-164   mu_s, Cs likewise;  out = chol(Cs) Xw + mu_s
//   cholesky_dec cWCT.py:111-132 retry with cumulative jitter eps, 2eps, 3eps, ... on failure
//   interpolation cWCT.py:206-262 mixL = sum_i a_i chol(Cs_i) (+ content blend), out = mixL Xw + mix_mu
//   _transfer_seg cWCT.py:49-109 the same per label on the pixels carrying that label
//
// Here the two matrix products are fused into one affine map per (content, style[, label]):
//   out = T x + t0,   T = mixL * Lc^-1,   t0 = mix_mu - T mu_c
// so the content features are read once for the statistics and once for the apply.
#include "common.h"

#define CWCT_MAX_STYLES 8
#define CWCT_MAX_TRIES 4096

// ================================================================================================
// statistics: per-workgroup shifted sums, combined in fp64
// partial record (floats): [0]=n  [4..4+N)=shift  [4+N..4+2N)=sum(x-shift)  [4+2N..)=sum (x-shift)(x-shift)^T
// ================================================================================================
__host__ __device__ inline size_t cwct_partial_stride(int N) { return (size_t)N * N + 2 * N + 4; }

static inline int cwct_stats_groups(long L, int* px_per_wg) {
    long per = 2048;
    long g = (L + per - 1) / per;
    if (g > 1024) { g = 1024; per = ((L + g - 1) / g + 63) / 64 * 64; g = (L + per - 1) / per; }
    *px_per_wg = (int)per;
    return (int)g;
}

template <int RB>
__global__ __launch_bounds__(256) void cwct_stats_partial_kernel(const float* __restrict__ x, long L,
                                                                 const uint8_t* __restrict__ mask, int label,
                                                                 float* __restrict__ partial, int px_per_wg) {
    constexpr int N = 16 * RB, PT = 64, LD = PT + 4;
    __shared__ __attribute__((aligned(16))) float xs[N * LD];
    __shared__ __attribute__((aligned(16))) float vflag[PT];
    __shared__ float sh[N];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const long p_begin = (long)blockIdx.x * px_per_wg;
    long p_end = p_begin + px_per_wg;
    if (p_end > L) p_end = L;
    for (int c = tid; c < N; c += 256) sh[c] = p_begin < L ? x[(size_t)c * L + p_begin] : 0.f;

    float q[RB][RB], asum[RB], cnt = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        asum[r] = 0.f;
#pragma unroll
        for (int s = 0; s < RB; ++s) q[r][s] = 0.f;
    }
    for (long p0 = p_begin; p0 < p_end; p0 += PT) {
        __syncthreads();
        if (tid < PT) {
            const long p = p0 + tid;
            vflag[tid] = (p < p_end && (mask == nullptr || mask[p] == label)) ? 1.f : 0.f;
        }
        __syncthreads();
        for (int idx = tid; idx < N * PT; idx += 256) {
            const int c = idx >> 6, pl = idx & 63;
            const long p = p0 + pl;
            xs[c * LD + pl] = vflag[pl] != 0.f ? x[(size_t)c * L + p] - sh[c] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int pl = 0; pl < PT; pl += 4) {
            float4 av[RB], bv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                av[r] = *(const float4*)&xs[(ti + 16 * r) * LD + pl];
                bv[r] = *(const float4*)&xs[(tj + 16 * r) * LD + pl];
            }
            const float4 vf = *(const float4*)&vflag[pl];
            cnt += vf.x + vf.y + vf.z + vf.w;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                asum[r] += (av[r].x + av[r].y) + (av[r].z + av[r].w);
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    q[r][s] = fmaf(av[r].x, bv[s].x, q[r][s]);
                    q[r][s] = fmaf(av[r].y, bv[s].y, q[r][s]);
                    q[r][s] = fmaf(av[r].z, bv[s].z, q[r][s]);
                    q[r][s] = fmaf(av[r].w, bv[s].w, q[r][s]);
                }
            }
        }
    }
    float* rec = partial + (size_t)blockIdx.x * cwct_partial_stride(N);
    if (tid == 0) rec[0] = cnt;
    for (int c = tid; c < N; c += 256) rec[4 + c] = sh[c];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (tj == 0) rec[4 + N + ti + 16 * r] = asum[r];
#pragma unroll
        for (int s = 0; s < RB; ++s) rec[4 + 2 * N + (size_t)(ti + 16 * r) * N + tj + 16 * s] = q[r][s];
    }
}

// Combine the per-workgroup records in fp64 (Chan et al. pairwise update).  Both kernels give 16 threads
// to every output (a channel mean / a covariance entry), each summing G/16 records, then reduce in LDS.
__global__ __launch_bounds__(256) void cwct_stats_mean_kernel(const float* __restrict__ partial, int G, int N,
                                                              double* __restrict__ stats) {
    __shared__ double sacc[16][17], snt[16][17];
    const int cl = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const size_t PS = cwct_partial_stride(N);
    double acc = 0.0, nt = 0.0;
    for (int g = gl; g < G; g += 16) {
        const float* rec = partial + (size_t)g * PS;
        const double n = rec[0];
        nt += n;
        acc += n * (double)rec[4 + c] + (double)rec[4 + N + c];
    }
    sacc[gl][cl] = acc; snt[gl][cl] = nt;
    __syncthreads();
    if (gl == 0) {
        double a2 = 0.0, n2 = 0.0;
        for (int k = 0; k < 16; ++k) { a2 += sacc[k][cl]; n2 += snt[k][cl]; }
        stats[1 + c] = n2 > 0.0 ? a2 / n2 : 0.0;
        if (c == 0) stats[0] = n2;
    }
}

__global__ __launch_bounds__(256) void cwct_stats_cov_kernel(const float* __restrict__ partial, int G, int N,
                                                             double* __restrict__ stats) {
    __shared__ double sm2[16][17];
    const int el = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int i = e / N, j = e - i * N;
    const size_t PS = cwct_partial_stride(N);
    const double mu_i = stats[1 + i], mu_j = stats[1 + j];
    double m2 = 0.0;
    for (int g = gl; g < G; g += 16) {
        const float* rec = partial + (size_t)g * PS;
        const double n = rec[0];
        if (n > 0.0) {
            const double ai = rec[4 + N + i], aj = rec[4 + N + j];
            const double di = (double)rec[4 + i] + ai / n - mu_i;
            const double dj = (double)rec[4 + j] + aj / n - mu_j;
            m2 += (double)rec[4 + 2 * N + e] - ai * aj / n + n * di * dj;
        }
    }
    sm2[gl][el] = m2;
    __syncthreads();
    if (gl == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sm2[k][el];
        stats[1 + N + e] = t / (stats[0] - 1.0);
    }
}

// ================================================================================================
// factor: Cholesky (fp32, LAPACK-like failure rule) with jitter retries, mix, triangular solve
// ================================================================================================
struct FactorArgs {
    const double* content;
    const double* styles[CWCT_MAX_STYLES];
    float alphas[CWCT_MAX_STYLES];
    int n_styles;
    float alpha_c;
    float eps;
    int N;
    float* affine;
    int* info;
};

// in-place lower Cholesky of the N x N matrix A (row stride N) in LDS; returns true on failure
__device__ bool chol_lds(float* A, int N, int tid, int* flag) {
    for (int j = 0; j < N; ++j) {
        __syncthreads();
        if (tid == 0) {
            const float d = A[j * N + j];
            if (!(d > 0.f)) *flag = 1; else A[j * N + j] = sqrtf(d);
        }
        __syncthreads();
        if (*flag) return true;
        const float piv = A[j * N + j];
        for (int i = j + 1 + tid; i < N; i += 256) A[i * N + j] /= piv;
        __syncthreads();
        const int m = N - 1 - j;
        for (int idx = tid; idx < m * m; idx += 256) {
            const int ii = j + 1 + idx / m, kk = j + 1 + idx % m;
            if (kk <= ii) A[ii * N + kk] -= A[ii * N + j] * A[kk * N + j];
        }
    }
    __syncthreads();
    return false;
}

// Cholesky of the covariance in `stats` with the cumulative-jitter schedule of cWCT.py:115-128.
__device__ int chol_with_jitter(const double* stats, float* A, int N, float eps, int tid, int* flag) {
    const double* cov = stats + 1 + N;
    int tries = 0;
    while (true) {
        __syncthreads();
        for (int idx = tid; idx < N * N; idx += 256) {
            float v = (float)cov[idx];
            if (idx / N == idx % N)
                for (int t = 1; t <= tries; ++t) v = v + (float)((double)t * (double)eps);
            A[idx] = v;
        }
        if (tid == 0) *flag = 0;
        __syncthreads();
        const bool failed = chol_lds(A, N, tid, flag);
        if (!failed || tries >= CWCT_MAX_TRIES) break;
        ++tries;
    }
    return tries;
}

__global__ __launch_bounds__(256) void cwct_factor_kernel(const FactorArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int N = a.N, tid = threadIdx.x, LDT = N + 1;
    float* A = (float*)fsm;                  // N*N      Cholesky workspace, finally Lc
    float* T = A + N * N;                    // N*(N+1)  mixL, solved in place into T
    double* mixmu = (double*)(T + N * LDT);  // N doubles; N*N + N*(N+1) floats is even -> 8-byte aligned
    int* const flagp = (int*)(mixmu + N);    // all LDS lives in the dynamic region (16-byte aligned base)

    for (int idx = tid; idx < N * LDT; idx += 256) T[idx] = 0.f;
    if (tid < N) mixmu[tid] = 0.0;
    __syncthreads();
    for (int s = 0; s < a.n_styles; ++s) {
        const int tries = chol_with_jitter(a.styles[s], A, N, a.eps, tid, flagp);
        if (tid == 0) a.info[2 + s] = tries;
        const float al = a.alphas[s];
        for (int idx = tid; idx < N * N; idx += 256) {
            const int i = idx / N, j = idx - i * N;
            if (j <= i) T[i * LDT + j] += A[idx] * al;
        }
        if (tid < N) mixmu[tid] += (double)(float)a.styles[s][1 + tid] * (double)al;
        __syncthreads();
    }
    const int ctries = chol_with_jitter(a.content, A, N, a.eps, tid, flagp);
    if (tid == 0) { a.info[0] = ctries; a.info[1] = ctries >= CWCT_MAX_TRIES; }
    if (a.alpha_c != 0.f) {
        const float ac = a.alpha_c;
        for (int idx = tid; idx < N * N; idx += 256) {
            const int i = idx / N, j = idx - i * N;
            if (j <= i) T[i * LDT + j] = T[i * LDT + j] * (1.f - ac) + A[idx] * ac;
        }
        if (tid < N) mixmu[tid] = mixmu[tid] * (double)(1.f - ac) + (double)(float)a.content[1 + tid] * (double)ac;
    }
    __syncthreads();
    // solve T * Lc = mixL row-wise from the last column (T lower triangular)
    if (tid < N) {
        float* row = T + tid * LDT;
        for (int j = tid; j >= 0; --j) {
            double acc = row[j];
            for (int k = j + 1; k <= tid; ++k) acc -= (double)row[k] * (double)A[k * N + j];
            row[j] = (float)(acc / (double)A[j * N + j]);
        }
        double t0 = mixmu[tid];
        for (int j = 0; j <= tid; ++j) t0 -= (double)row[j] * a.content[1 + j];
        a.affine[N * N + tid] = (float)t0;
    }
    __syncthreads();
    for (int idx = tid; idx < N * N; idx += 256) a.affine[idx] = T[(idx / N) * LDT + idx % N];
}

// ================================================================================================
// apply: y[:,p] = T x[:,p] + t0   (T, t0 read through the scalar path: uniform addresses)
// ================================================================================================
template <int N, int PX>
__global__ __launch_bounds__(256) void cwct_apply_kernel(const float* x, float* y, long L,
                                                         const float* __restrict__ affine,
                                                         const uint8_t* __restrict__ mask, int label) {
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * PX;
    if (p >= L) return;
    bool on[PX];
    bool any = false;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        on[k] = (p + k < L) && (mask == nullptr || mask[p + k] == label);
        any |= on[k];
    }
    if (!any) return;
    float xr[N][PX];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        if (PX == 4) {
            const float4 v = *(const float4*)(x + (size_t)c * L + p);
            xr[c][0] = v.x; xr[c][1] = v.y; xr[c][2] = v.z; xr[c][3] = v.w;
        } else if (PX == 2) {
            const float2 v = *(const float2*)(x + (size_t)c * L + p);
            xr[c][0] = v.x; xr[c][1] = v.y;
        } else {
            xr[c][0] = x[(size_t)c * L + p];
        }
    }
    const float* t0 = affine + N * N;
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        float acc[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) acc[k] = t0[i];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float t = affine[i * N + j];
#pragma unroll
            for (int k = 0; k < PX; ++k) acc[k] = fmaf(t, xr[j][k], acc[k]);
        }
        float* dst = y + (size_t)i * L + p;
        bool all = true;
#pragma unroll
        for (int k = 0; k < PX; ++k) all &= on[k];
        if (PX == 4 && all) *(float4*)dst = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else if (PX == 2 && all) *(float2*)dst = make_float2(acc[0], acc[1]);
        else {
#pragma unroll
            for (int k = 0; k < PX; ++k) if (on[k]) dst[k] = acc[k];
        }
    }
}

template <int N>
static int launch_apply(const float* x, float* y, long L, const float* affine, const uint8_t* mask, int label,
                        hipStream_t st) {
    constexpr int PXV = N <= 32 ? 4 : (N <= 64 ? 2 : 1);
    const bool vec = (L % PXV) == 0 && (((uintptr_t)x | (uintptr_t)y) % (4 * PXV)) == 0;
    if (vec && PXV > 1) {
        const long nthreads = L / PXV;
        cwct_apply_kernel<N, PXV><<<dim3((unsigned)((nthreads + 255) / 256)), 256, 0, st>>>(x, y, L, affine, mask, label);
    } else {
        cwct_apply_kernel<N, 1><<<dim3((unsigned)((L + 255) / 256)), 256, 0, st>>>(x, y, L, affine, mask, label);
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

extern "C" {

size_t vst_cwct_stats_workspace_bytes(int N, long L) {
    int per;
    const int g = cwct_stats_groups(L, &per);
    return (size_t)g * cwct_partial_stride(N) * sizeof(float);
}

int vst_cwct_stats(const float* x, int N, long L, const uint8_t* mask, int label, double* stats, void* workspace,
                   void* stream) {
    if (!x || !stats || L <= 0) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (!(N == 16 || N == 32 || N == 64 || N == 128)) return VST_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    int per;
    const int G = cwct_stats_groups(L, &per);
    float* partial = (float*)workspace;
    switch (N) {
        case 16: cwct_stats_partial_kernel<1><<<G, 256, 0, st>>>(x, L, mask, label, partial, per); break;
        case 32: cwct_stats_partial_kernel<2><<<G, 256, 0, st>>>(x, L, mask, label, partial, per); break;
        case 64: cwct_stats_partial_kernel<4><<<G, 256, 0, st>>>(x, L, mask, label, partial, per); break;
        default: cwct_stats_partial_kernel<8><<<G, 256, 0, st>>>(x, L, mask, label, partial, per); break;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    cwct_stats_mean_kernel<<<N / 16, 256, 0, st>>>(partial, G, N, stats);
    VST_RETURN_IF_LAUNCH_FAILED();
    cwct_stats_cov_kernel<<<N * N / 16, 256, 0, st>>>(partial, G, N, stats);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_factor(const double* content_stats, const double* const* style_stats_host_array,
                    const float* alphas_host, int n_styles, float alpha_c, float eps, int N, float* affine, int* info,
                    void* stream) {
    if (!content_stats || !style_stats_host_array || !alphas_host || !affine || !info) return VST_E_ARG;
    if (n_styles < 1 || n_styles > CWCT_MAX_STYLES) return VST_E_ARG;
    if (!(N == 16 || N == 32 || N == 64 || N == 128)) return VST_E_SHAPE;
    FactorArgs a{};
    a.content = content_stats;
    for (int i = 0; i < n_styles; ++i) {
        if (!style_stats_host_array[i]) return VST_E_ARG;
        a.styles[i] = style_stats_host_array[i];
        a.alphas[i] = alphas_host[i];
    }
    a.n_styles = n_styles; a.alpha_c = alpha_c; a.eps = eps; a.N = N; a.affine = affine; a.info = info;
    const size_t lds = (size_t)N * N * 4 + (size_t)N * (N + 1) * 4 + (size_t)N * 8 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)cwct_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           150 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    cwct_factor_kernel<<<1, 256, lds, (hipStream_t)stream>>>(a);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_apply(const float* x, float* y, int N, long L, const float* affine, const uint8_t* mask, int label,
                   void* stream) {
    if (!x || !y || !affine || L <= 0) return VST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (N) {
        case 16: return launch_apply<16>(x, y, L, affine, mask, label, st);
        case 32: return launch_apply<32>(x, y, L, affine, mask, label, st);
        case 64: return launch_apply<64>(x, y, L, affine, mask, label, st);
        case 128: return launch_apply<128>(x, y, L, affine, mask, label, st);
        default: return VST_E_SHAPE;
    }
}

}  // extern "C"

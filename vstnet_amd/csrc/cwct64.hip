// fp64 cWCT: the device side of cWCT(use_double=True).
//
// Reference: models/cWCT.py:13-16 (the flag), :35-47 (_transfer), :66,:106 (_transfer_seg), :220,:238,:259 (interpolation):
// the features are converted to double, mean / covariance / Cholesky / inverse / both matrix products run in fp64, and the
// result is converted back to the input dtype.  Nobody on the reference's scripts' path sets the flag (a fidelity option, not a
// hot path), so these kernels are written for exactness and clarity, not for the roofline: true fp64 two-pass statistics
// (mean, then centred products: what `x - mean` followed by `x @ x.T` computes), an fp64 Cholesky with the reference's
// cumulative jitter schedule (:111-132), triangular inverse, T = mixL Lc^-1, and y = float(T double(x) + t0).
#include "common.h"

#define CWCT64_MAX_STYLES 8
#define CWCT64_MAX_TRIES 4096
#define CWCT64_CHUNK 2048          // pixels per workgroup of the statistics passes

// ---- statistics -------------------------------------------------------------------------------------------------------------
// pass 1: per chunk, the fp64 sum of every channel over the chunk's (selected) pixels and the pixel count
template <int N>
__global__ __launch_bounds__(256) void stats64_sum_kernel(const float* __restrict__ x, long L, const uint8_t* __restrict__ mask,
                                                          int label, double* __restrict__ part) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long p0 = (long)blockIdx.x * CWCT64_CHUNK, p1 = p0 + CWCT64_CHUNK < L ? p0 + CWCT64_CHUNK : L;
    double* out = part + (size_t)blockIdx.x * (N + 1);
    for (int c = wave; c < N + 1; c += 4) {               // "channel" N = the count
        double s = 0.0;
        for (long p = p0 + lane; p < p1; p += 64) {
            const bool on = mask == nullptr || mask[p] == label;
            if (on) s += c < N ? (double)x[(size_t)c * L + p] : 1.0;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[c] = s;
    }
}

// pass 2 (one workgroup): n and the means, chunks added in index order
template <int N>
__global__ __launch_bounds__(256) void stats64_mean_kernel(const double* __restrict__ part, int G, double* __restrict__ stats) {
    const int c = threadIdx.x;
    if (c > N) return;
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * (N + 1) + c];
    __shared__ double n_sh;
    if (c == N) { n_sh = s; stats[0] = s; }
    __syncthreads();
    if (c < N) stats[1 + c] = n_sh > 0.0 ? s / n_sh : 0.0;
}

// pass 3: per chunk, the centred co-moments sum_p (x_i - mu_i)(x_j - mu_j) in fp64.  A tile of 32 pixels is staged centred
// (masked-out pixels as zeros) in LDS; thread t owns the pairs q = t + 256 r, (i, j) = (q / N, q % N).
template <int N>
__global__ __launch_bounds__(256) void stats64_cov_kernel(const float* __restrict__ x, long L, const uint8_t* __restrict__ mask,
                                                          int label, const double* __restrict__ stats, double* __restrict__ part) {
    constexpr int PAIRS = N * N / 256 > 0 ? N * N / 256 : 1, TP = 32;
    __shared__ double tile[TP][N + 1];
    __shared__ double mu[N];
    const int tid = threadIdx.x;
    if (tid < N) mu[tid] = stats[1 + tid];
    const long p0 = (long)blockIdx.x * CWCT64_CHUNK, p1 = p0 + CWCT64_CHUNK < L ? p0 + CWCT64_CHUNK : L;
    double acc[PAIRS];
#pragma unroll
    for (int r = 0; r < PAIRS; ++r) acc[r] = 0.0;
    __syncthreads();
    for (long t0 = p0; t0 < p1; t0 += TP) {
        for (int e = tid; e < TP * N; e += 256) {
            const int c = e / TP, pp = e % TP;
            const long p = t0 + pp;
            const bool on = p < p1 && (mask == nullptr || mask[p] == label);
            tile[pp][c] = on ? (double)x[(size_t)c * L + p] - mu[c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PAIRS; ++r) {
            const int q = tid + 256 * r;
            if (q < N * N) {
                const int i = q / N, j = q % N;
                double s = acc[r];
#pragma unroll 8
                for (int pp = 0; pp < TP; ++pp) s = fma(tile[pp][i], tile[pp][j], s);
                acc[r] = s;
            }
        }
        __syncthreads();
    }
    double* out = part + (size_t)blockIdx.x * N * N;
#pragma unroll
    for (int r = 0; r < PAIRS; ++r) {
        const int q = tid + 256 * r;
        if (q < N * N) out[q] = acc[r];
    }
}

// pass 4: cov = sum of the chunks' co-moments (index order) / (n - 1)   (cWCT.py:144, :157)
template <int N>
__global__ __launch_bounds__(256) void stats64_cov_combine_kernel(const double* __restrict__ part, int G, double* __restrict__ stats) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= N * N) return;
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * N * N + q];
    const double n = stats[0];
    stats[1 + N + q] = s / (n - 1.0);
}

// ---- factor -----------------------------------------------------------------------------------------------------------------
struct Factor64Args {
    const double* content;
    const double* styles[CWCT64_MAX_STYLES];
    float alphas[CWCT64_MAX_STYLES];
    int n_styles;
    float alpha_c, eps;
    int N;
    double* affine;
    int* info;
    double* ws;            // 4 N^2 doubles
};

// lower Cholesky factor of the covariance of `stats` into A (row-major N x N, upper triangle zeroed); returns the retries.
// Failure rule as LAPACK's (pivot <= 0 or NaN); jitter eps, 2 eps, ... cumulative, each step rounded to fp32 like the
// reference's float32 identity times eps (cWCT.py:120-128).  A prefactored record (stats[0] < 0) holds the factor itself.
__device__ int chol64(double* A, const double* stats, int N, float eps, int min_tries, int* s_flag) {
    const int tid = threadIdx.x;
    const double* cov = stats + 1 + N;
    if (stats[0] < 0.0) {
        for (int e = tid; e < N * N; e += 256) A[e] = cov[e];
        __syncthreads();
        return 0;
    }
    int tries = min_tries < 0 ? 0 : (min_tries > CWCT64_MAX_TRIES ? CWCT64_MAX_TRIES : min_tries);
    while (true) {
        for (int e = tid; e < N * N; e += 256) {
            double v = cov[e];
            if (e / N == e % N)
                for (int t = 1; t <= tries; ++t) v += (double)(float)((double)t * (double)eps);
            A[e] = v;
        }
        __syncthreads();
        bool failed = false;
        for (int j = 0; j < N; ++j) {
            if (tid == 0) {
                const double d = A[j * N + j];
                const bool bad = !(d > 0.0);
                *s_flag = bad;
                if (!bad) A[j * N + j] = sqrt(d);
            }
            __syncthreads();
            if (*s_flag) { failed = true; break; }
            const double piv = A[j * N + j];
            for (int i = j + 1 + tid; i < N; i += 256) A[i * N + j] /= piv;
            __syncthreads();
            const int m = N - j - 1;
            for (int e = tid; e < m * m; e += 256) {
                const int i = j + 1 + e / m, k = j + 1 + e % m;
                if (k <= i) A[i * N + k] -= A[i * N + j] * A[k * N + j];
            }
            __syncthreads();
        }
        __syncthreads();
        if (!failed || tries >= CWCT64_MAX_TRIES) break;
        ++tries;
    }
    for (int e = tid; e < N * N; e += 256)
        if (e % N > e / N) A[e] = 0.0;
    __syncthreads();
    return tries;
}

__global__ __launch_bounds__(256) void cwct_factor64_kernel(const Factor64Args a) {
    const int N = a.N, tid = threadIdx.x;
    __shared__ int s_flag;
    double* Lc = a.ws;
    double* Li = a.ws + (size_t)N * N;
    double* Ls = a.ws + (size_t)2 * N * N;
    double* M = a.ws + (size_t)3 * N * N;
    for (int e = tid; e < N * N; e += 256) M[e] = 0.0;
    __syncthreads();
    for (int s = 0; s < a.n_styles; ++s) {
        const int min_tries = a.info[2 + s];
        __syncthreads();
        const int tries = chol64(Ls, a.styles[s], N, a.eps, min_tries, &s_flag);
        if (tid == 0) a.info[2 + s] = tries;
        const double al = (double)a.alphas[s];
        for (int e = tid; e < N * N; e += 256) M[e] += Ls[e] * al;
        __syncthreads();
    }
    const int cmin = a.info[0];
    __syncthreads();
    const int ctries = chol64(Lc, a.content, N, a.eps, cmin, &s_flag);
    if (tid == 0) { a.info[0] = ctries; a.info[1] = ctries >= CWCT64_MAX_TRIES; }
    const double ac = (double)a.alpha_c;
    if (a.alpha_c != 0.f) {
        for (int e = tid; e < N * N; e += 256) M[e] = M[e] * (1.0 - ac) + Lc[e] * ac;
    }
    // Li = Lc^-1 (lower triangular): column c by forward substitution, one thread per column
    for (int e = tid; e < N * N; e += 256) Li[e] = 0.0;
    __syncthreads();
    if (tid < N) {
        const int c = tid;
        Li[c * N + c] = 1.0 / Lc[c * N + c];
        for (int i = c + 1; i < N; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s = fma(Lc[i * N + k], Li[k * N + c], s);
            Li[i * N + c] = -s / Lc[i * N + i];
        }
    }
    __syncthreads();
    // T = M Li ;  t0 = mix_mean - T mean_c
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e % N;
        double s = 0.0;
        for (int k = j; k <= i; ++k) s = fma(M[i * N + k], Li[k * N + j], s);
        a.affine[e] = s;
    }
    __syncthreads();
    if (tid < N) {
        double mm = 0.0;
        for (int s = 0; s < a.n_styles; ++s) mm += a.styles[s][1 + tid] * (double)a.alphas[s];
        if (a.alpha_c != 0.f) mm = mm * (1.0 - ac) + a.content[1 + tid] * ac;
        double s = 0.0;
        for (int k = 0; k < N; ++k) s = fma(a.affine[tid * N + k], a.content[1 + k], s);
        a.affine[N * N + tid] = mm - s;
    }
}

// ---- apply ------------------------------------------------------------------------------------------------------------------
// y[:, p] = float(T double(x[:, p]) + t0); one pixel per thread, the pixel's N inputs in registers (so y may alias x), T / t0
// through the scalar path (uniform addresses).  With a mask only pixels whose label matches are written.
template <int N>
__global__ __launch_bounds__(256) void cwct_apply64_kernel(const float* x, float* y, long L, const double* __restrict__ affine,
                                                           const uint8_t* __restrict__ mask, int label) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= L || (mask != nullptr && mask[p] != label)) return;
    float xr[N];
#pragma unroll
    for (int c = 0; c < N; ++c) xr[c] = x[(size_t)c * L + p];
    const double* t0 = affine + N * N;
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        double acc = t0[i];
#pragma unroll
        for (int j = 0; j < N; ++j) acc = fma(affine[i * N + j], (double)xr[j], acc);
        y[(size_t)i * L + p] = (float)acc;
    }
}

template <int N>
static int stats64_launch(const float* x, long L, const uint8_t* mask, int label, double* stats, double* ws, hipStream_t st) {
    const int G = (int)((L + CWCT64_CHUNK - 1) / CWCT64_CHUNK);
    double* part_sum = ws;
    double* part_cov = ws + (size_t)G * (N + 1);
    stats64_sum_kernel<N><<<G, 256, 0, st>>>(x, L, mask, label, part_sum);
    stats64_mean_kernel<N><<<1, 256, 0, st>>>(part_sum, G, stats);
    stats64_cov_kernel<N><<<G, 256, 0, st>>>(x, L, mask, label, stats, part_cov);
    stats64_cov_combine_kernel<N><<<(N * N + 255) / 256, 256, 0, st>>>(part_cov, G, stats);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

extern "C" {

size_t vst_cwct_stats_f64_workspace_bytes(int N, long L) {
    if (N <= 0 || L <= 0) return 0;
    const size_t G = (size_t)((L + CWCT64_CHUNK - 1) / CWCT64_CHUNK);
    return G * ((size_t)N + 1 + (size_t)N * N) * sizeof(double);
}

int vst_cwct_stats_f64(const float* x, int N, long L, const uint8_t* mask, int label, double* stats, void* workspace,
                       void* stream) {
    if (!x || !stats || L <= 0) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    switch (N) {
        case 16: return stats64_launch<16>(x, L, mask, label, stats, (double*)workspace, st);
        case 32: return stats64_launch<32>(x, L, mask, label, stats, (double*)workspace, st);
        case 64: return stats64_launch<64>(x, L, mask, label, stats, (double*)workspace, st);
        case 128: return stats64_launch<128>(x, L, mask, label, stats, (double*)workspace, st);
        default: return VST_E_SHAPE;
    }
}

size_t vst_cwct_factor_f64_workspace_bytes(int N) { return N > 0 ? (size_t)4 * N * N * sizeof(double) : 0; }

int vst_cwct_factor_f64(const double* content_stats, const double* const* style_stats_host_array, const float* alphas_host,
                        int n_styles, float alpha_c, float eps, int N, double* affine, int* info, void* workspace,
                        void* stream) {
    if (!content_stats || !style_stats_host_array || !alphas_host || !affine || !info) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (n_styles < 1 || n_styles > CWCT64_MAX_STYLES) return VST_E_ARG;
    if (!(N == 16 || N == 32 || N == 64 || N == 128)) return VST_E_SHAPE;
    Factor64Args a{};
    a.content = content_stats;
    for (int i = 0; i < n_styles; ++i) {
        if (!style_stats_host_array[i]) return VST_E_ARG;
        a.styles[i] = style_stats_host_array[i];
        a.alphas[i] = alphas_host[i];
    }
    a.n_styles = n_styles; a.alpha_c = alpha_c; a.eps = eps; a.N = N; a.affine = affine; a.info = info;
    a.ws = (double*)workspace;
    cwct_factor64_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_apply_f64(const float* x, float* y, int N, long L, const double* affine, const uint8_t* mask, int label,
                       void* stream) {
    if (!x || !y || !affine || L <= 0) return VST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((L + 255) / 256);
    switch (N) {
        case 16: cwct_apply64_kernel<16><<<grid, 256, 0, st>>>(x, y, L, affine, mask, label); break;
        case 32: cwct_apply64_kernel<32><<<grid, 256, 0, st>>>(x, y, L, affine, mask, label); break;
        case 64: cwct_apply64_kernel<64><<<grid, 256, 0, st>>>(x, y, L, affine, mask, label); break;
        case 128: cwct_apply64_kernel<128><<<grid, 256, 0, st>>>(x, y, L, affine, mask, label); break;
        default: return VST_E_SHAPE;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

}  // extern "C"

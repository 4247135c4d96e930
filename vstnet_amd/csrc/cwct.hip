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
#ifndef VST_LBL_ABL
#define VST_LBL_ABL 0        // timing-only builds of cwct_stats_labels_kernel: 1 = no row sums, 2 = no MFMAs, 4 = no staging;
                             // 128 = cwct_stats_mfma_kernel without its MFMAs
#endif
#define CWCT_MAX_TRIES 4096

// ================================================================================================
// statistics: per-workgroup shifted sums, combined in fp64
// partial record (floats): [0]=n  [4..4+N)=shift  [4+N..4+2N)=sum(x-shift)  [4+2N..)=sum (x-shift)(x-shift)^T
// ================================================================================================
__host__ __device__ inline size_t cwct_partial_stride(int N) { return (size_t)N * N + 2 * N + 4; }

static inline int cwct_stats_groups(long L, int* px_per_wg) {
    long per = 2048;
    while (per > 512 && L / per < 256) per >>= 1;          // enough workgroups to fill the chip on small codes
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

// MFMA form of the partial kernel for N in {32, 64, 128}: Q = Xs Xs^T on v_mfma_f32_32x32x2_f32 (exact fp32
// products / accumulation, i.e. the same numerics as the FMA form, at the matrix-core rate).  A and B operand of a
// 32x32 block are the same data (lane l: xs[32*blk + (l&31)][pixel + (l>>5)]), so NBLK fragment loads feed NBLK
// MFMAs per wave and pixel pair.  Wave w owns row block w % NBLK and pixel group w / NBLK of the 64-pixel tile.
template <int NBLK, bool VEC>
__global__ __launch_bounds__(256) void cwct_stats_mfma_kernel(const float* __restrict__ x, long L,
                                                              const uint8_t* __restrict__ mask, int label,
                                                              float* __restrict__ partial, int px_per_wg) {
    constexpr int N = 32 * NBLK, PT = 64, LD = PT + 1, PG = 4 / NBLK, PPG = PT / PG;
    constexpr int NV = N * PT / 4 / 256;                 // float4 groups per thread and tile (channel c = e>>4, pixels 4*(e&15)..)
    __shared__ float xs[N * LD];
    __shared__ float sh[N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = wave % NBLK, pg = wave / NBLK;
    const long p_begin = (long)blockIdx.x * px_per_wg;
    long p_end = p_begin + px_per_wg;
    if (p_end > L) p_end = L;
    for (int c = tid; c < N; c += 256) sh[c] = p_begin < L ? x[(size_t)c * L + p_begin] : 0.f;

    // tile prefetch registers, TWO tiles deep (a workgroup keeps 2 x N x 256 B in flight; with one tile the kernel sat at
    // 2.4 TB/s): values and per-pixel validity bits of this thread's float4 groups
    float4 pv[2][NV];
    unsigned pm[2][NV];
#define STATS_PREFETCH(S, p0_)                                                                                   \
    {                                                                                                           \
        const long p0__ = (p0_);                                                                                \
        _Pragma("unroll") for (int it = 0; it < NV; ++it) {                                                     \
            const int e = it * 256 + tid, c = e >> 4;                                                           \
            const long p = p0__ + 4 * (e & 15);                                                                 \
            unsigned m = 0;                                                                                     \
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
            if (VEC) {                                                                                          \
                if (p < p_end) {                          /* p_end, L and p are multiples of 4 here */          \
                    m = 0xf;                                                                                    \
                    if (mask != nullptr) {                                                                      \
                        const uchar4 mk = *(const uchar4*)(mask + p);                                           \
                        m = (mk.x == label) | ((mk.y == label) << 1) | ((mk.z == label) << 2) | ((mk.w == label) << 3); \
                    }                                                                                           \
                    if (m) v = *(const float4*)(x + (size_t)c * L + p);     /* other labels' pixels: never fetched */ \
                }                                                                                               \
            } else {                                                                                            \
                float t[4] = {0.f, 0.f, 0.f, 0.f};                                                              \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                   \
                    if (p + q < p_end) {                                                                        \
                        t[q] = x[(size_t)c * L + p + q];                                                        \
                        m |= (mask == nullptr || mask[p + q] == label) << q;                                    \
                    }                                                                                           \
                v = make_float4(t[0], t[1], t[2], t[3]);                                                        \
            }                                                                                                   \
            pv[S][it] = v; pm[S][it] = m;                                                                       \
        }                                                                                                       \
    }

    f32x16 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float asum[NV], cnt = 0.f;                         // shifted row sums of this thread's (channel, 4-pixel column) groups;
#pragma unroll                                         // pixel count (channel-0 owners)
    for (int it = 0; it < NV; ++it) asum[it] = 0.f;
    // one 64-pixel tile from prefetch slot S; refills the slot with the tile two ahead once its values are in LDS
#define STATS_TILE(S, p0_)                                                                                       \
    {                                                                                                           \
        const long p0t = (p0_);                                                                                 \
        /* barrier: the previous tile's MFMAs are done with xs.  With a mask it also tells whether this 64-pixel tile */ \
        /* holds any pixel of the label at all: the per-label passes of a masked transfer skip everything else */  \
        bool live = true;                                                                                       \
        if (mask != nullptr) {                                                                                  \
            unsigned anym = 0;                                                                                  \
            _Pragma("unroll") for (int it = 0; it < NV; ++it) anym |= pm[S][it];                                \
            live = __syncthreads_or((int)anym);                                                                 \
        } else {                                                                                                \
            __syncthreads();                                                                                    \
        }                                                                                                       \
        if (live) {                                                                                             \
            _Pragma("unroll") for (int it = 0; it < NV; ++it) {                                                 \
                const int e = it * 256 + tid, c = e >> 4, pl = 4 * (e & 15);                                    \
                const float s0 = sh[c];                                                                         \
                const unsigned m = pm[S][it];                                                                   \
                float* d = xs + c * LD + pl;                                                                    \
                const float d0 = (m & 1) ? pv[S][it].x - s0 : 0.f, d1 = (m & 2) ? pv[S][it].y - s0 : 0.f;       \
                const float d2 = (m & 4) ? pv[S][it].z - s0 : 0.f, d3 = (m & 8) ? pv[S][it].w - s0 : 0.f;       \
                d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3;                                                     \
                asum[it] += (d0 + d1) + (d2 + d3);       /* row sums: per staging thread, reduced once at the end */ \
                if (c == 0) cnt += (float)__popc(m);                                                            \
            }                                                                                                   \
            __syncthreads();                                                                                    \
        }                                                                                                       \
        if (p0t + 2 * PT < p_end) STATS_PREFETCH(S, p0t + 2 * PT)   /* in flight during two tiles' row sums and MFMAs */ \
        if (live) {                                                                                             \
            const float* base = xs + (lane & 31) * LD + pg * PPG + (lane >> 5);                                 \
            _Pragma("unroll 4") for (int t = 0; t < PPG / 2; ++t) {                                             \
                float f[NBLK];                                                                                  \
                _Pragma("unroll") for (int b = 0; b < NBLK; ++b) f[b] = base[b * 32 * LD + 2 * t];              \
                float fa = f[0];                              /* f[rb] without a runtime register index */      \
                _Pragma("unroll") for (int b = 1; b < NBLK; ++b) fa = rb == b ? f[b] : fa;                      \
                if (!(VST_LBL_ABL & 128))                                                                       \
                _Pragma("unroll") for (int b = 0; b < NBLK; ++b)                                                \
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, f[b], acc[b], 0, 0, 0);                   \
            }                                                                                                   \
        }                                                                                                       \
    }
    STATS_PREFETCH(0, p_begin)
    if (p_begin + PT < p_end) STATS_PREFETCH(1, p_begin + PT)
    for (long p0 = p_begin; p0 < p_end; p0 += 2 * PT) {
        STATS_TILE(0, p0)
        if (p0 + PT < p_end) STATS_TILE(1, p0 + PT)
    }
#undef STATS_TILE
#undef STATS_PREFETCH
    // ---- combine the pixel groups (PG > 1) through LDS, then one record per workgroup ------------------------------
    float* rec = partial + (size_t)blockIdx.x * cwct_partial_stride(N);
    if (PG > 1) {
        // one pixel group per round adds its accumulators to group 0 through the (now idle) tile buffer:
        // NBLK*NBLK*16*64 floats <= N*LD for N = 32 (1024 <= 2080) and N = 64 (4096 <= 4160)
        for (int round = 1; round < PG; ++round) {
            __syncthreads();
            if (pg == round) {
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xs[((rb * NBLK + b) * 16 + r) * 64 + lane] = acc[b][r];
            }
            __syncthreads();
            if (pg == 0) {
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] += xs[((rb * NBLK + b) * 16 + r) * 64 + lane];
            }
        }
    }
    // pixel count: the 16 threads that stage channel 0 (e>>4 == 0 <=> it == 0, tid < 16) each counted their 4 pixels
    __syncthreads();
    if (tid < 16) xs[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        float c2 = 0.f;
        for (int k = 0; k < 16; ++k) c2 += xs[k];
        rec[0] = c2;
    }
    for (int c = tid; c < N; c += 256) rec[4 + c] = sh[c];
    // row sums: the 16 lanes e & 15 = 0..15 of a staging group hold the columns of one channel c = e >> 4
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        float v = asum[it];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 16);
        if ((tid & 15) == 0) rec[4 + N + ((it * 256 + tid) >> 4)] = v;
    }
    if (pg == 0) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = b * 32 + (lane & 31);
                rec[4 + 2 * N + (size_t)i * N + j] = acc[b][r];
            }
    }
}

// Combine the per-workgroup records in fp64 (Chan et al. pairwise update).  Both kernels give 16 threads
// to every output (a channel mean / a covariance entry), each summing G/16 records, then reduce in LDS.
// blockIdx.y = label slot of a multi-label pass (records of one workgroup are `rs` consecutive slots; slot0 + blockIdx.y
// past *n_slots: nothing to do); single-label callers pass rs = 1, slot0 = 0, n_slots = nullptr.
__global__ __launch_bounds__(256) void cwct_stats_mean_kernel(const float* __restrict__ partial, int G, int N,
                                                              double* __restrict__ stats, int rs, int slot0,
                                                              const int* __restrict__ n_slots) {
    __shared__ double sacc[16][17], snt[16][17];
    if (n_slots != nullptr && slot0 + (int)blockIdx.y >= *n_slots) return;
    const int cl = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const size_t PS = cwct_partial_stride(N) * rs;
    partial += (size_t)blockIdx.y * cwct_partial_stride(N);
    stats += (size_t)(slot0 + blockIdx.y) * (1 + N + (size_t)N * N);
    double acc = 0.0, nt = 0.0;
#pragma unroll 4                                         // independent records: keep several L2 round trips in flight per lane
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
                                                             double* __restrict__ stats, int rs, int slot0,
                                                             const int* __restrict__ n_slots) {
    __shared__ double sm2[16][17];
    if (n_slots != nullptr && slot0 + (int)blockIdx.y >= *n_slots) return;
    const int el = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int i = e / N, j = e - i * N;
    const size_t PS = cwct_partial_stride(N) * rs;
    partial += (size_t)blockIdx.y * cwct_partial_stride(N);
    stats += (size_t)(slot0 + blockIdx.y) * (1 + N + (size_t)N * N);
    const double mu_i = stats[1 + i], mu_j = stats[1 + j];
    double m2 = 0.0;
#pragma unroll 4                                         // independent records: loads first (no branch around them), then the arithmetic
    for (int g = gl; g < G; g += 16) {
        const float* rec = partial + (size_t)g * PS;
        const float nf = rec[0], aif = rec[4 + N + i], ajf = rec[4 + N + j], si = rec[4 + i], sj = rec[4 + j], q = rec[4 + 2 * N + e];
        const double n = nf, ai = aif, aj = ajf;
        const double rn = nf > 0.f ? 1.0 / n : 0.0;
        const double di = (double)si + ai * rn - mu_i;
        const double dj = (double)sj + aj * rn - mu_j;
        m2 += nf > 0.f ? (double)q - ai * aj * rn + n * di * dj : 0.0;
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
    // multi-label form: workgroup s = blockIdx.x factors slot s: content / styles[i] / affine / info advance by these strides
    // per slot; workgroups at or past *n_slots exit.  All zero / null for the single-pair launch.
    size_t content_stride, style_stride, affine_stride;
    int info_stride;
    const int* n_slots;
};

// The N x N matrices live in registers, distributed 2-D cyclically over the 16 x 16 threads: thread (ti, tj)
// owns elements (ti + 16 a, tj + 16 b), a, b < BLK = N / 16.  Column / row broadcasts go through LDS; two
// barriers per Cholesky column, one per solve column.  fp32 arithmetic with LAPACK's failure rule (pivot <= 0
// or NaN) so that the jitter retries of cholesky_dec (cWCT.py:111-132) trigger like the reference's.
template <int BLK>
__device__ __forceinline__ void fac_load(const double* stats, int tries, float eps, int ti, int tj, float (&r)[BLK][BLK]) {
    constexpr int N = 16 * BLK;
    const double* cov = stats + 1 + N;
#pragma unroll
    for (int a = 0; a < BLK; ++a)
#pragma unroll
        for (int b = 0; b < BLK; ++b) {
            const int i = ti + 16 * a, k = tj + 16 * b;
            float v = (float)cov[i * N + k];
            if (i == k)
                for (int t = 1; t <= tries; ++t) v = v + (float)((double)t * (double)eps);   // eps, 2 eps, ... cumulative
            r[a][b] = v;
        }
}

// One barrier per column ("look-ahead"): the phase that applies column j's rank-1 update also publishes the UNSCALED
// column j+1 (diagonal included) in the other half of `col` (2N floats); after the barrier every thread takes the pivot
// sqrt(col[j+1]) and the scaled entries it needs itself.  Same divisions and products as the two-barrier form.
template <int BLK>
__device__ bool fac_chol(float (&r)[BLK][BLK], int ti, int tj, float* col, float* s_piv, int* s_flag) {
    constexpr int N = 16 * BLK;
    (void)s_piv; (void)s_flag;
    __syncthreads();                                        // a previous factorisation may still be reading col
    if (tj == 0) {
#pragma unroll
        for (int a = 0; a < BLK; ++a) col[ti + 16 * a] = r[a][0];
    }
    __syncthreads();
    int p = 0;
#pragma unroll
    for (int jb = 0; jb < BLK; ++jb) {
#pragma unroll 1
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * jb + jj;
            const float* cu = col + p * N;
            const float d = cu[j];
            if (!(d > 0.f)) return true;                    // uniform: every thread reads the same word
            const float piv = sqrtf(d);
            const float rpiv = 1.0f / piv;                  // LAPACK's potf2 scales the column by 1/ajj as well
            float li[BLK], lk[BLK];
#pragma unroll
            for (int a = jb; a < BLK; ++a) li[a] = cu[ti + 16 * a] * rpiv;
#pragma unroll
            for (int b = jb; b < BLK; ++b) lk[b] = cu[tj + 16 * b] * rpiv;
            if (tj == jj) {
#pragma unroll
                for (int a = jb; a < BLK; ++a) {
                    const int i = ti + 16 * a;
                    if (i > j) r[a][jb] = li[a];
                    else if (i == j) r[a][jb] = piv;
                }
            }
#pragma unroll
            for (int a = jb; a < BLK; ++a)
#pragma unroll
                for (int b = jb; b < BLK; ++b) {
                    const int i = ti + 16 * a, k = tj + 16 * b;
                    if (i > j && k > j && k <= i) r[a][b] -= li[a] * lk[b];
                }
            // publish the next column, unscaled (its owner threads have just finished updating it)
            const int jn = j + 1;
            if (jn < N && tj == (jn & 15)) {
                float* cn = col + (p ^ 1) * N;
#pragma unroll
                for (int a = jb; a < BLK; ++a) {
                    const int i = ti + 16 * a;
                    const float v = jj < 15 ? r[a][jb] : r[a][jb + 1 < BLK ? jb + 1 : jb];
                    if (i >= jn) cn[i] = v;
                }
            }
            __syncthreads();
            p ^= 1;
        }
    }
#pragma unroll
    for (int a = 0; a < BLK; ++a)
#pragma unroll
        for (int b = 0; b < BLK; ++b)
            if (tj + 16 * b > ti + 16 * a) r[a][b] = 0.f;
    return false;
}

// lower Cholesky factor of the covariance in `stats` (or the stored factor if the record is prefactored:
// stats[0] < 0) -> r; returns the number of jitter retries.  min_tries > 0 starts the schedule there (the reference
// factors a [B,N,N] stack at once, so a sample inherits the retries another sample of its batch needed)
template <int BLK>
__device__ int fac_factor(const double* stats, float eps, int ti, int tj, float (&r)[BLK][BLK], float* col, float* s_piv,
                          int* s_flag, int min_tries = 0) {
    constexpr int N = 16 * BLK;
    if (stats[0] < 0.0) {
#pragma unroll
        for (int a = 0; a < BLK; ++a)
#pragma unroll
            for (int b = 0; b < BLK; ++b) r[a][b] = (float)stats[1 + N + (ti + 16 * a) * N + tj + 16 * b];
        return 0;
    }
    int tries = min_tries < 0 ? 0 : (min_tries > CWCT_MAX_TRIES ? CWCT_MAX_TRIES : min_tries);
    while (true) {
        fac_load<BLK>(stats, tries, eps, ti, tj, r);
        const bool failed = fac_chol<BLK>(r, ti, tj, col, s_piv, s_flag);
        if (!failed || tries >= CWCT_MAX_TRIES) break;
        ++tries;
    }
    return tries;
}

template <int BLK>
__global__ __launch_bounds__(256) void cwct_factor_kernel(const FactorArgs args) {
    constexpr int N = 16 * BLK;
    // (the per-slot view is built from scalars: indexing / modifying the kernel-argument struct itself would move it to scratch)
    size_t slot = 0;
    if (args.n_slots != nullptr) {
        if ((int)blockIdx.x >= *args.n_slots) return;
        slot = blockIdx.x;
    }
    struct {
        const double* content;
        const double* const* styles_base;
        size_t style_off;
        const float* alphas;
        int n_styles;
        float alpha_c, eps;
        float* affine;
        int* info;
        __device__ const double* style(int s) const { return styles_base[s] + style_off; }
    } a = {args.content + slot * args.content_stride, args.styles, slot * args.style_stride, args.alphas, args.n_styles,
           args.alpha_c, args.eps, args.affine + slot * args.affine_stride, args.info + slot * args.info_stride};
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    float* Lmat = (float*)fsm;             // N*N   Lc, row-major
    float* col = Lmat + N * N;             // 2*N   column broadcast of the Cholesky (look-ahead double buffer)
    float* s_piv = col + 3 * N;            // (the launch reserves N*N + 3N floats + 16 bytes)
    int* s_flag = (int*)(s_piv + 1);
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;

    float r[BLK][BLK], m[BLK][BLK];
    double mixmu[BLK];
#pragma unroll
    for (int a2 = 0; a2 < BLK; ++a2) {
        mixmu[a2] = 0.0;
#pragma unroll
        for (int b = 0; b < BLK; ++b) m[a2][b] = 0.f;
    }
    for (int s = 0; s < a.n_styles; ++s) {
        const int tries = fac_factor<BLK>(a.style(s), a.eps, ti, tj, r, col, s_piv, s_flag, a.info[2 + s]);
        if (tid == 0) a.info[2 + s] = tries;      // (every thread read its minimum before the first barrier of fac_chol)
        const float al = a.alphas[s];
#pragma unroll
        for (int a2 = 0; a2 < BLK; ++a2) {
            mixmu[a2] += (double)(float)a.style(s)[1 + ti + 16 * a2] * (double)al;
#pragma unroll
            for (int b = 0; b < BLK; ++b) m[a2][b] += r[a2][b] * al;
        }
    }
    const int ctries = fac_factor<BLK>(a.content, a.eps, ti, tj, r, col, s_piv, s_flag, a.info[0]);
    if (tid == 0) { a.info[0] = ctries; a.info[1] = ctries >= CWCT_MAX_TRIES; }
    if (a.alpha_c != 0.f) {
        const float ac = a.alpha_c;
#pragma unroll
        for (int a2 = 0; a2 < BLK; ++a2) {
            mixmu[a2] = mixmu[a2] * (double)(1.f - ac) + (double)(float)a.content[1 + ti + 16 * a2] * (double)ac;
#pragma unroll
            for (int b = 0; b < BLK; ++b) m[a2][b] = m[a2][b] * (1.f - ac) + r[a2][b] * ac;
        }
    }
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < BLK; ++a2)
#pragma unroll
        for (int b = 0; b < BLK; ++b) Lmat[(ti + 16 * a2) * N + tj + 16 * b] = r[a2][b];
    __syncthreads();
    // ---- solve T * Lc = mixL in place (m := T), last column first; T stays lower triangular ---------------------------
    // Row i of T depends on row i of mixL only, and a row's 16 column owners (tj = 0..15) are 16 consecutive lanes of
    // one wave: the column value travels by a 16-lane shuffle and Lc is read-only in LDS, so the whole solve needs no
    // barrier (it was one barrier per column: 140 of the kernel's 240 us at N = 128).
#pragma unroll
    for (int jb = BLK - 1; jb >= 0; --jb) {
#pragma unroll 1
        for (int jj = 15; jj >= 0; --jj) {
            const int j = 16 * jb + jj;
            const float rljj = 1.0f / Lmat[j * N + j];
            float lrow[BLK];
#pragma unroll
            for (int b = 0; b <= jb; ++b) lrow[b] = Lmat[j * N + tj + 16 * b];
#pragma unroll
            for (int a2 = 0; a2 < BLK; ++a2) {
                const float mine = m[a2][jb] * rljj;             // meaningful in lane tj == jj
                const float t = __shfl(mine, jj, 16);
                if (tj == jj) m[a2][jb] = t;
#pragma unroll
                for (int b = 0; b <= jb; ++b)
                    if (tj + 16 * b < j) m[a2][b] -= t * lrow[b];
            }
        }
    }
    // ---- t0 = mix_mean - T * mean_c (reduce over the 16 threads of a row: contiguous lanes), write {T, t0} ----------
#pragma unroll
    for (int a2 = 0; a2 < BLK; ++a2) {
        double part = 0.0;
#pragma unroll
        for (int b = 0; b < BLK; ++b) part += (double)m[a2][b] * a.content[1 + tj + 16 * b];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) part += __shfl_xor(part, off, 16);
        if (tj == 0) a.affine[N * N + ti + 16 * a2] = (float)(mixmu[a2] - part);
#pragma unroll
        for (int b = 0; b < BLK; ++b) a.affine[(ti + 16 * a2) * N + tj + 16 * b] = m[a2][b];
    }
}

// stats {n, mean, cov} -> prefactored record {-(n+1), mean, L} so that later factor calls skip this Cholesky
template <int BLK>
__global__ __launch_bounds__(256) void cwct_prefactor_kernel(const double* stats, float eps, double* out, int* info) {
    constexpr int N = 16 * BLK;
    __shared__ float col[2 * N];
    __shared__ float s_piv;
    __shared__ int s_flag;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    float r[BLK][BLK];
    const int tries = fac_factor<BLK>(stats, eps, ti, tj, r, col, &s_piv, &s_flag);
    const double n = stats[0];
    const double mean = tid < N ? stats[1 + tid] : 0.0;
    __syncthreads();                                        // out may alias stats: every read is done
    if (tid == 0) { out[0] = n < 0.0 ? n : -(n + 1.0); info[0] = tries; }
    if (tid < N) out[1 + tid] = mean;
#pragma unroll
    for (int a = 0; a < BLK; ++a)
#pragma unroll
        for (int b = 0; b < BLK; ++b) out[1 + N + (ti + 16 * a) * N + tj + 16 * b] = (double)r[a][b];
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

// MFMA form for N in {32, 64, 128}: D[32 channels x 32 pixels] += T[32 x 2] * X[2 x 32] on v_mfma_f32_32x32x2_f32
// (exact fp32).  T sits in LDS (row stride N+1: conflict-free column reads); the X operand is read straight from
// global memory, coalesced along pixels: lane l holds channel 2t + (l>>5) of pixels (l&31)*PXV .. +PXV-1, i.e.
// PXV interleaved 32-pixel sets per wave, so loads and stores are PXV floats wide.
template <int NBLK, int PXV>
__global__ __launch_bounds__(256) void cwct_apply_mfma_kernel(const float* x, float* y, long L,
                                                              const float* __restrict__ affine,
                                                              const uint8_t* __restrict__ mask, int label, long ngroups) {
    constexpr int N = 32 * NBLK, LDT = N + 1, GP = 32 * PXV;
    extern __shared__ __attribute__((aligned(16))) float asm_[];
    float* Tl = asm_;
    float* t0 = asm_ + N * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < N * N; idx += 256) Tl[(idx / N) * LDT + idx % N] = affine[idx];
    for (int idx = tid; idx < N; idx += 256) t0[idx] = affine[N * N + idx];
    __syncthreads();
    const int col = lane & 31, kh = lane >> 5;
    for (long g = (long)blockIdx.x * 4 + wave; g < ngroups; g += (long)gridDim.x * 4) {
        const long pl = g * GP + (long)col * PXV;
        const bool inside = pl < L;                          // L % PXV == 0: the lane's PXV pixels are all in or all out
        const long plc = inside ? pl : L - PXV;
        bool on[PXV];
        bool any = false, all = true;
#pragma unroll
        for (int q = 0; q < PXV; ++q) {
            on[q] = inside && (mask == nullptr || mask[plc + q] == label);
            any |= on[q]; all &= on[q];
        }
        if (!__any(any)) continue;                           // no pixel of this label in the wave's 32*PXV pixels
        f32x16 acc[PXV][NBLK];
#pragma unroll
        for (int q = 0; q < PXV; ++q)
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][b][r] = 0.f;
        const float* xp = x + plc + (size_t)kh * L;
        const float* tp = Tl + col * LDT + kh;
#pragma unroll 4
        for (int t = 0; t < N / 2; ++t) {
            float v[PXV];
            if (PXV == 4) { const float4 w = *(const float4*)(xp + (size_t)2 * t * L); v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w; }
            else if (PXV == 2) { const float2 w = *(const float2*)(xp + (size_t)2 * t * L); v[0] = w.x; v[1] = w.y; }
            else v[0] = xp[(size_t)2 * t * L];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const float av = tp[b * 32 * LDT + 2 * t];
#pragma unroll
                for (int q = 0; q < PXV; ++q) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v[q], acc[q][b], 0, 0, 0);
            }
        }
        if (!any) continue;
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float bias = t0[i];
                float* dst = y + (size_t)i * L + pl;
                if (PXV == 4 && all) *(float4*)dst = make_float4(acc[0][b][r] + bias, acc[1][b][r] + bias, acc[2][b][r] + bias, acc[3][b][r] + bias);
                else if (PXV == 2 && all) *(float2*)dst = make_float2(acc[0][b][r] + bias, acc[1][b][r] + bias);
                else {
#pragma unroll
                    for (int q = 0; q < PXV; ++q) if (on[q]) dst[q] = acc[q][b][r] + bias;
                }
            }
    }
}

// Unmasked apply for N >= 64 on the bf16 matrix cores with split operands (the same "bf16x3" arithmetic as the
// convs: T = Th + Tl, x = xh + xl, y += Th*xl + Tl*xh + Th*xh, fp32 accumulate): the fp32 MFMA form is bound by the
// 32x32x2 instruction's throughput at N = 128 (8.6 GFLOP per 512x512 code -> 63 us at best, 152 measured), this one by
// HBM.  T's fragments are pre-split once per workgroup into LDS; a wave takes 64 pixels per iteration (16 lanes x 4
// consecutive pixels, float4 loads and stores along pixels), four interleaved 16-pixel B operands.
template <int N>
__global__ __launch_bounds__(256, 2) void cwct_apply_split_kernel(const float* x, float* y, long L,
                                                                  const float* __restrict__ affine, long niter) {
    constexpr int MB = N / 16, KS = N / 32;
    constexpr int PQ = N >= 128 ? 2 : 4;                     // interleaved 16-pixel sets per wave (accumulators: PQ*MB*4 VGPRs)
    typedef __attribute__((ext_vector_type(PQ))) float fvec;
    extern __shared__ __attribute__((aligned(16))) unsigned char asm2_[];
    uint4* const th = (uint4*)asm2_;                        // [MB][KS][4 kg][16 lrow] fragments of 8 bf16
    uint4* const tl = th + MB * KS * 64;
    float* const t0 = (float*)(tl + MB * KS * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int f = tid; f < MB * KS * 64; f += 256) {
        const int lrow = f & 15, kg = (f >> 4) & 3, ks = (f >> 6) % KS, m = (f >> 6) / KS;
        const float* src = affine + (size_t)(16 * m + lrow) * N + 32 * ks + 8 * kg;
        bf16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = src[i];
            h[i] = (__bf16)v;
            l[i] = (__bf16)(v - (float)h[i]);
        }
        th[f] = __builtin_bit_cast(uint4, h);
        tl[f] = __builtin_bit_cast(uint4, l);
    }
    for (int i = tid; i < N; i += 256) t0[i] = affine[N * N + i];
    __syncthreads();
    const int n16 = lane & 15, kg = lane >> 4;
    for (long it = (long)blockIdx.x * 4 + wave; it < niter; it += (long)gridDim.x * 4) {
        const long p = it * (16 * PQ) + PQ * n16;            // L % (16 PQ) == 0 (launch condition)
        f32x4 acc[PQ][MB];
#pragma unroll
        for (int q = 0; q < PQ; ++q)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[q][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1                                             // keep one k-step's loads in flight, not all of them
        for (int ks = 0; ks < KS; ++ks) {
            fvec v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = *(const fvec*)(x + (size_t)(32 * ks + 8 * kg + c) * L + p);
            bf16x8 xh[PQ], xl[PQ];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                    xh[q][c] = (__bf16)v[c][q];
                    xl[q][c] = (__bf16)(v[c][q] - (float)xh[q][c]);
                }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const bf16x8 wh = __builtin_bit_cast(bf16x8, th[(m * KS + ks) * 64 + kg * 16 + n16]);
                const bf16x8 wl = __builtin_bit_cast(bf16x8, tl[(m * KS + ks) * 64 + kg * 16 + n16]);
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                    acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[q], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[q], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[q], acc[q][m], 0, 0, 0);
                }
            }
        }
        // lane (n16, kg) holds output channels 16m + 4kg + r of pixels p + q
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * m + 4 * kg + r;
                const float b = t0[co];
                fvec o;
#pragma unroll
                for (int q = 0; q < PQ; ++q) o[q] = acc[q][m][r] + b;
                *(fvec*)(y + (size_t)co * L + p) = o;
            }
    }
}

template <int N>
static int launch_apply_split(const float* x, float* y, long L, const float* affine, hipStream_t st) {
    constexpr int MB = N / 16, KS = N / 32;
    const long niter = L / (N >= 128 ? 32 : 64);
    long wgs = (niter + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    const size_t lds = (size_t)2 * MB * KS * 64 * 16 + N * sizeof(float);
    auto kern = cwct_apply_split_kernel<N>;
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)lds, &attr_done)) return rc_;
    kern<<<dim3((unsigned)wgs), 256, lds, st>>>(x, y, L, affine, niter);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

template <int NBLK, int PXV>
static int launch_apply_mfma(const float* x, float* y, long L, const float* affine, const uint8_t* mask, int label,
                             hipStream_t st) {
    constexpr int N = 32 * NBLK;
    const long ngroups = (L + 32 * PXV - 1) / (32 * PXV);
    long wgs = (ngroups + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    const size_t lds = ((size_t)N * (N + 1) + N) * sizeof(float);
    auto kern = cwct_apply_mfma_kernel<NBLK, PXV>;
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)((int)lds), &attr_done)) return rc_;
    kern<<<dim3((unsigned)wgs), 256, lds, st>>>(x, y, L, affine, mask, label, ngroups);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

#ifndef VST_APPLY_SPLIT_MIN_N
#define VST_APPLY_SPLIT_MIN_N 64
#endif
template <int N>
static int launch_apply(const float* x, float* y, long L, const float* affine, const uint8_t* mask, int label,
                        bool exact, hipStream_t st) {
    if (!exact && N >= VST_APPLY_SPLIT_MIN_N && mask == nullptr && (L % 64) == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0)
        return launch_apply_split<(N >= 32 ? N : 32)>(x, y, L, affine, st);
    if (N >= 32) {                                           // matrix-core path (needs vector-aligned rows)
        constexpr int PXM = N == 32 ? 4 : (N == 64 ? 2 : 1);    // (N = 128 with two pixel sets needs 512 VGPRs and spills)
        if ((L % PXM) == 0 && L >= PXM && (((uintptr_t)x | (uintptr_t)y) % (4 * PXM)) == 0)
            return launch_apply_mfma<(N >= 32 ? N / 32 : 1), PXM>(x, y, L, affine, mask, label, st);
    }
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

// ================================================================================================
// single-pass masked transfer (models/cWCT.py:49-109,166-189): every label in ONE statistics pass and ONE apply pass
// ================================================================================================
// Plan record, built on the device from the two label maps (no host histogram, no sync): the labels that pass the
// reference's validity rule (count_c > 10, count_s > 10, count ratio < 100 both ways, cWCT.py:178) get consecutive slots
// in increasing label order; lut[label] = slot, 255 = "keep the content feature".
#define CWCT_MAX_SLOTS 32
struct LabelPlan {
    int n_slots, overflow;
    int hist_c[256], hist_s[256];
    unsigned char lut[256];
    unsigned char slot_label[CWCT_MAX_SLOTS];
};
static_assert(sizeof(LabelPlan) == VST_LABEL_PLAN_BYTES, "vstnet.h: VST_LABEL_PLAN_BYTES");

__global__ __launch_bounds__(256) void label_hist_kernel(const uint8_t* __restrict__ mask, long L, int* __restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < L; p += (long)gridDim.x * 256) atomicAdd(&h[mask[p]], 1);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void label_plan_kernel(LabelPlan* plan) {
    __shared__ int valid[256];
    const int l = threadIdx.x;
    const int a = plan->hist_c[l], b = plan->hist_s[l];
    // a / b < 100 and b / a < 100 in the reference's float arithmetic == a < 100 b and b < 100 a for positive counts
    valid[l] = a > 10 && b > 10 && (double)a / (double)b < 100.0 && (double)b / (double)a < 100.0;
    __syncthreads();
    if (l == 0) {
        int n = 0, over = 0;
        for (int k = 0; k < 256; ++k) {
            unsigned char s = 255;
            if (valid[k]) {
                if (n < CWCT_MAX_SLOTS) { s = (unsigned char)n; plan->slot_label[n] = (unsigned char)k; ++n; }
                else over = 1;
            }
            plan->lut[k] = s;
        }
        plan->n_slots = n; plan->overflow = over;
    }
}

// Statistics of KRES label slots [slot0, slot0 + KRES) in one pass over x: like cwct_stats_mfma_kernel, but the 64 pixels of
// a tile are staged into LDS SORTED by slot (stable; each slot's run padded to a multiple of 4 with zero columns), so each
// slot's X X^T runs over its own pixels only: about one tile's worth of MFMAs per tile however the labels are mixed.
// Every wave sorts for itself (lane = pixel: ballots and popcounts, destinations handed to the staging lanes by shuffles),
// so the tile costs the same two barriers as the unmasked kernel.  The products run on v_mfma_f32_16x16x4_f32 (exact fp32):
// the N x N covariance is (N/16)^2 blocks of 16 x 16, wave w owns blocks w, w + 4, ... for ALL resident slots and all of
// the tile's pixels (4 accumulator registers per block and slot), so nothing is combined across waves at the end.
template <int NBLK, int KRES>
__global__ __launch_bounds__(256, NBLK == 1 ? 3 : 1) void cwct_stats_labels_kernel(const float* __restrict__ x, long L,
                                                                                   const uint8_t* __restrict__ mask,
                                                                                   const LabelPlan* __restrict__ plan, int slot0,
                                                                                   float* __restrict__ partial, int px_per_wg) {
    constexpr int N = 32 * NBLK, PT = 64, LD = PT + 3 * KRES + 1;
    constexpr int NV = N * PT / 4 / 256;                  // float4 groups per thread and tile
    constexpr int TPC = 256 / N, SPAN = TPC;               // row sums: TPC threads per channel, each takes every TPC-th column
    constexpr int NB16 = N / 16, BPW = NB16 * NB16 / 4;    // 16 x 16 blocks per side / per wave
    __shared__ float xs[N * LD];
    __shared__ float sh[N];
    __shared__ unsigned char lut[256];
    if (slot0 >= plan->n_slots) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long p_begin = (long)blockIdx.x * px_per_wg;
    long p_end = p_begin + px_per_wg;
    if (p_end > L) p_end = L;
    for (int c = tid; c < N; c += 256) sh[c] = p_begin < L ? x[(size_t)c * L + p_begin] : 0.f;
    lut[tid] = plan->lut[tid];
    __syncthreads();

    // two tiles in flight per workgroup
    float4 pvq[2][NV];
    int relq[2];                                           // slot (relative to slot0) of pixel p0 + lane of a prefetched tile, -1 = none
    const bool vec = (L % 4) == 0 && ((uintptr_t)x % 16) == 0;
#define PREFETCH(slot_, p0_)                                                                        \
    {                                                                                               \
        const long p0__ = (p0_);                                                                    \
        const long pm = p0__ + lane;                                                                \
        int rel_next = -1;                                                                          \
        if (pm < p_end) {                                                                           \
            const int sl = lut[mask[pm]];                                                           \
            rel_next = (sl != 255 && sl >= slot0 && sl < slot0 + KRES) ? sl - slot0 : -1;           \
        }                                                                                           \
        relq[slot_] = rel_next;                                                                     \
        _Pragma("unroll") for (int it = 0; it < NV; ++it) {                                         \
            const int e = it * 256 + tid, c = e >> 4;                                               \
            const long p = p0__ + 4 * (e & 15);                                                     \
            float t[4] = {0.f, 0.f, 0.f, 0.f};                                                      \
            if (vec && p + 3 < p_end) {                                                             \
                const float4 v = *(const float4*)(x + (size_t)c * L + p);                           \
                t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;                                     \
            } else {                                                                                \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                       \
                    if (p + q < p_end) t[q] = x[(size_t)c * L + p + q];                             \
            }                                                                                       \
            pvq[slot_][it] = make_float4(t[0], t[1], t[2], t[3]);                                   \
        }                                                                                           \
    }

    f32x4 acc[KRES][BPW];
#pragma unroll
    for (int k = 0; k < KRES; ++k)
#pragma unroll
        for (int b = 0; b < BPW; ++b) acc[k][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum[KRES], cnt[KRES];
#pragma unroll
    for (int k = 0; k < KRES; ++k) { asum[k] = 0.f; cnt[k] = 0.f; }
    const int sc = tid % N, sp = tid / N;                  // row sums: channel and column phase of this thread
    const int fr = lane & 15, fq = lane >> 4;              // MFMA operand lane: row within the block, pixel within the group of 4
    PREFETCH(0, p_begin);
    PREFETCH(1, p_begin + PT);
#pragma unroll 1
    for (long p00 = p_begin; p00 < p_end; p00 += 2 * PT) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {               // register set `half` holds this tile (static indices)
        const long p0 = p00 + half * PT;
        if (p0 >= p_end) break;
        // ---- stable sort of the tile's pixels by slot, per wave in registers ------------------------------------------
        const int rel = relq[half];
        int seg_begin[KRES], seg_cnt[KRES];
        int base = 0, mine = -1;
#pragma unroll
        for (int k = 0; k < KRES; ++k) {
            const unsigned long long bal = __ballot(rel == k);
            const int c = __popcll(bal);
            if (rel == k) mine = base + __popcll(bal & ((1ull << lane) - 1ull));
            seg_begin[k] = base; seg_cnt[k] = c;
            base += (c + 3) & ~3;
        }
        if (base == 0) {                                   // uniform: no pixel of these slots in the tile
            PREFETCH(half, p0 + 2 * PT);
            continue;
        }
        __syncthreads();                                   // previous tile's MFMAs and row sums are done with xs
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            if (VST_LBL_ABL & 4) break;
            const int e = it * 256 + tid, c = e >> 4, pl = 4 * (e & 15);
            const float s0 = sh[c];
            const float v[4] = {pvq[half][it].x, pvq[half][it].y, pvq[half][it].z, pvq[half][it].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = __shfl(mine, pl + q, 64);   // destination column of pixel pl + q (lane = pixel in every wave)
                if (d >= 0) xs[c * LD + d] = v[q] - s0;
            }
        }
        if (tid < N) {                                     // zero pad columns of every run whose length is not a multiple of 4
#pragma unroll
            for (int k = 0; k < KRES; ++k)
                for (int j = seg_cnt[k]; j < ((seg_cnt[k] + 3) & ~3); ++j) xs[tid * LD + seg_begin[k] + j] = 0.f;
        }
        __syncthreads();
        PREFETCH(half, p0 + 2 * PT);                       // in flight during the row sums and MFMAs below and the whole next tile
#pragma unroll
        for (int k = 0; k < KRES; ++k) {
            if (VST_LBL_ABL & 1) break;
            float sacc = 0.f;
            const int nb = seg_begin[k], nc = seg_cnt[k];
            for (int j0 = sp; j0 < nc; j0 += 4 * SPAN) {     // four independent LDS reads per round (runtime trip count)
                float t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * SPAN;
                    t4[u] = xs[sc * LD + nb + (j < nc ? j : nc - 1)];
                    t4[u] = j < nc ? t4[u] : 0.f;
                }
                sacc += (t4[0] + t4[1]) + (t4[2] + t4[3]);
            }
            asum[k] += sacc;
            cnt[k] += (float)nc;
        }
#pragma unroll
        for (int k = 0; k < KRES; ++k) {
            if (VST_LBL_ABL & 2) break;
            const int nb = seg_begin[k], ng = (seg_cnt[k] + 3) >> 2;    // groups of 4 pixels
            for (int g = 0; g < ng; ++g) {
                float f[NB16];
#pragma unroll
                for (int b = 0; b < NB16; ++b) f[b] = xs[(16 * b + fr) * LD + nb + 4 * g + fq];
#pragma unroll
                for (int b = 0; b < BPW; ++b) {
                    const int blk = wave + 4 * b, bi = blk / NB16, bj = blk % NB16;     // wave-uniform block coordinates
                    float fa = f[0], fb = f[0];
#pragma unroll
                    for (int t = 1; t < NB16; ++t) { fa = bi == t ? f[t] : fa; fb = bj == t ? f[t] : fb; }
                    acc[k][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[k][b], 0, 0, 0);
                }
            }
        }
      }
    }
#undef PREFETCH
    // ---- one record per workgroup and slot ---------------------------------------------------------------------------
    const size_t PS = cwct_partial_stride(N);
    __syncthreads();
    float* red = xs;                                       // [KRES][TPC][N] partial row sums -> summed below
#pragma unroll
    for (int k = 0; k < KRES; ++k) red[(k * TPC + sp) * N + sc] = asum[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KRES; ++k) {
        float* rec = partial + ((size_t)blockIdx.x * KRES + k) * PS;
        if (tid < N) {
            float t = 0.f;
            for (int j = 0; j < TPC; ++j) t += red[(k * TPC + j) * N + tid];
            rec[4 + N + tid] = t;
        }
        if (tid == 0) rec[0] = cnt[k];
        for (int c = tid; c < N; c += 256) rec[4 + c] = sh[c];
        // accumulator layout of the 16x16 MFMA: lane (j = lane % 16, ig = lane / 16) holds rows 4 ig + r, column j
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int blk = wave + 4 * b, bi = blk / NB16, bj = blk % NB16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rec[4 + 2 * N + (size_t)(16 * bi + 4 * fq + r) * N + 16 * bj + fr] = acc[k][b][r];
        }
    }
}

// y[:,p] = T[slot(p)] x[:,p] + t0[slot(p)] for the slots [slot0, slot0 + KRES); FIRST pass (slot0 == 0): every other pixel
// gets y = x; later passes write only their own slots' pixels.  One MFMA sweep per slot PRESENT in a wave's pixel group,
// with the other pixels' columns zeroed: each column belongs to one slot, so all sweeps add into the same accumulators.
template <int NBLK, int PXV, int KRES>
__global__ __launch_bounds__(256) void cwct_apply_labels_kernel(const float* x, float* y, long L,
                                                                const float* __restrict__ affines,
                                                                const uint8_t* __restrict__ mask,
                                                                const LabelPlan* __restrict__ plan, int slot0, long ngroups) {
    constexpr int N = 32 * NBLK, LDT = N + 1, GP = 32 * PXV, TS = N * LDT + N;   // floats per slot in LDS: T (padded rows), t0
    extern __shared__ __attribute__((aligned(16))) float asm3_[];
    __shared__ unsigned char lut[256];
    const int n_slots = plan->n_slots;
    if (slot0 > 0 && slot0 >= n_slots) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = slot0 == 0;
    // slot index KRES is the identity map of the first pass (pixels of no slot or of a later pass: y = x)
    for (int k = 0; k <= KRES; ++k) {
        float* Tl = asm3_ + (size_t)k * TS;
        const bool ident = k == KRES || slot0 + k >= n_slots;
        const float* aff = affines + (size_t)(slot0 + k) * ((size_t)N * N + N);
        for (int idx = tid; idx < N * N; idx += 256) {
            const int i = idx / N, j = idx - i * N;
            Tl[i * LDT + j] = ident ? (i == j ? 1.f : 0.f) : aff[idx];
        }
        for (int idx = tid; idx < N; idx += 256) Tl[N * LDT + idx] = ident ? 0.f : aff[N * N + idx];
    }
    lut[tid] = plan->lut[tid];
    __syncthreads();
    const int col = lane & 31, kh = lane >> 5;
    for (long g = (long)blockIdx.x * 4 + wave; g < ngroups; g += (long)gridDim.x * 4) {
        const long pl = g * GP + (long)col * PXV;
        const bool inside = pl < L;
        const long plc = inside ? pl : L - PXV;
        int sl[PXV];                                         // slot relative to slot0 (KRES = identity / not mine)
        unsigned present = 0;
#pragma unroll
        for (int q = 0; q < PXV; ++q) {
            int s2 = inside ? (int)lut[mask[plc + q]] : 255;
            s2 = (s2 != 255 && s2 >= slot0 && s2 < slot0 + KRES) ? s2 - slot0 : KRES;
            sl[q] = inside ? s2 : -1;
            if (inside) present |= 1u << s2;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) present |= (unsigned)__shfl_xor((int)present, o, 64);   // slots present in the group
        if (!first) present &= (1u << KRES) - 1u;
        if (present == 0) continue;
        f32x16 acc[PXV][NBLK];
#pragma unroll
        for (int q = 0; q < PXV; ++q)
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][b][r] = 0.f;
        const float* xp = x + plc + (size_t)kh * L;
        for (int k = 0; k <= KRES; ++k) {
            if (!((present >> k) & 1u)) continue;            // uniform
            const float* tp = asm3_ + (size_t)k * TS + col * LDT + kh;
            bool mine[PXV];
#pragma unroll
            for (int q = 0; q < PXV; ++q) mine[q] = sl[q] == k;
#pragma unroll 4
            for (int t = 0; t < N / 2; ++t) {
                float v[PXV];
                if (PXV == 4) { const float4 w = *(const float4*)(xp + (size_t)2 * t * L); v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w; }
                else if (PXV == 2) { const float2 w = *(const float2*)(xp + (size_t)2 * t * L); v[0] = w.x; v[1] = w.y; }
                else v[0] = xp[(size_t)2 * t * L];
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const float av = tp[b * 32 * LDT + 2 * t];
#pragma unroll
                    for (int q = 0; q < PXV; ++q)
                        acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, mine[q] ? v[q] : 0.f, acc[q][b], 0, 0, 0);
                }
            }
        }
        bool on[PXV];
        bool all = true;
#pragma unroll
        for (int q = 0; q < PXV; ++q) { on[q] = sl[q] >= 0 && (first || sl[q] < KRES); all &= on[q]; }
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float o[PXV];
#pragma unroll
                for (int q = 0; q < PXV; ++q) {
                    const int s2 = sl[q] < 0 ? KRES : sl[q];
                    o[q] = acc[q][b][r] + asm3_[(size_t)s2 * TS + N * LDT + i];
                }
                float* dstp = y + (size_t)i * L + pl;
                if (PXV == 4 && all) *(float4*)dstp = make_float4(o[0], o[1], o[2], o[3]);
                else if (PXV == 2 && all) *(float2*)dstp = make_float2(o[0], o[1]);
                else {
#pragma unroll
                    for (int q = 0; q < PXV; ++q) if (on[q]) dstp[q] = o[q];
                }
            }
    }
}

// The same per-pixel map on the bf16 matrix cores with split operands (cwct_apply_split_kernel's arithmetic, ~1.5e-5):
// sixteen times the rate of the exact-fp32 MFMA, so one sweep per slot present in a 16*PQ-pixel group costs next to nothing
// and the pass stays HBM-bound however the labels are mixed.  A pixel's x fragments are split once and zeroed per slot by
// selects (one pixel = one lane's whole fragment).  Pixels that this pass does not transform are copied (first pass) or left
// alone (later passes): their lanes re-read x at the output channel positions.  Needs L % (16 PQ) == 0 and 16-byte rows.
template <int N, int KAPP>
__global__ __launch_bounds__(256, 2) void cwct_apply_labels_split_kernel(const float* x, float* y, long L,
                                                                         const float* __restrict__ affines,
                                                                         const uint8_t* __restrict__ mask,
                                                                         const LabelPlan* __restrict__ plan, int slot0, long niter) {
    constexpr int MB = N / 16, KS = N / 32, TF = MB * KS * 64;
    constexpr int PQ = N >= 128 ? 2 : 4;
    typedef __attribute__((ext_vector_type(PQ))) float fvec;
    extern __shared__ __attribute__((aligned(16))) unsigned char asm4_[];
    uint4* const th = (uint4*)asm4_;                        // [KAPP][MB][KS][4 kg][16 lrow] fragments of 8 bf16
    uint4* const tl = th + KAPP * TF;
    float* const t0 = (float*)(tl + KAPP * TF);             // [KAPP][N]
    __shared__ unsigned char lut[256];
    const int n_slots = plan->n_slots;
    if (slot0 > 0 && slot0 >= n_slots) return;
    const bool first = slot0 == 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int f0 = tid; f0 < KAPP * TF; f0 += 256) {
        const int k = f0 / TF, f = f0 - k * TF;
        const int lrow = f & 15, kg = (f >> 4) & 3, ks = (f >> 6) % KS, m = (f >> 6) / KS;
        bf16x8 h, l;
        const bool have = slot0 + k < n_slots;
        const float* src = affines + (size_t)(slot0 + (have ? k : 0)) * ((size_t)N * N + N) + (size_t)(16 * m + lrow) * N + 32 * ks + 8 * kg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = have ? src[i] : 0.f;
            h[i] = (__bf16)v;
            l[i] = (__bf16)(v - (float)h[i]);
        }
        th[f0] = __builtin_bit_cast(uint4, h);
        tl[f0] = __builtin_bit_cast(uint4, l);
    }
    for (int i = tid; i < KAPP * N; i += 256) {
        const int k = i / N;
        t0[i] = slot0 + k < n_slots ? affines[(size_t)(slot0 + k) * ((size_t)N * N + N) + (size_t)N * N + (i - k * N)] : 0.f;
    }
    lut[tid] = plan->lut[tid];
    __syncthreads();
    const int n16 = lane & 15, kg = lane >> 4;
    for (long it = (long)blockIdx.x * 4 + wave; it < niter; it += (long)gridDim.x * 4) {
        const long p = it * (16 * PQ) + PQ * n16;
        int sl[PQ];                                          // slot relative to slot0, or -1 = not transformed by this pass
        unsigned present = 0;
        bool any_other = false, all_mine = true;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int s2 = lut[mask[p + q]];
            sl[q] = (s2 != 255 && s2 >= slot0 && s2 < slot0 + KAPP) ? s2 - slot0 : -1;
            if (sl[q] >= 0) present |= 1u << sl[q];
            any_other |= sl[q] < 0;
            all_mine &= sl[q] >= 0;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) present |= (unsigned)__shfl_xor((int)present, o, 64);
        if (present == 0 && !first) continue;                // uniform: nothing of this pass in the wave's pixels
        f32x4 acc[PQ][MB];
#pragma unroll
        for (int q = 0; q < PQ; ++q)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[q][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (present != 0) {
#pragma unroll 1
            for (int ks = 0; ks < KS; ++ks) {
                fvec v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = *(const fvec*)(x + (size_t)(32 * ks + 8 * kg + c) * L + p);
                bf16x8 xh[PQ], xl[PQ];
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int q = 0; q < PQ; ++q) {
                        xh[q][c] = (__bf16)v[c][q];
                        xl[q][c] = (__bf16)(v[c][q] - (float)xh[q][c]);
                    }
                for (int k = 0; k < KAPP; ++k) {
                    if (!((present >> k) & 1u)) continue;    // uniform
                    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                    bf16x8 mh[PQ], ml[PQ];
#pragma unroll
                    for (int q = 0; q < PQ; ++q) { mh[q] = sl[q] == k ? xh[q] : zero; ml[q] = sl[q] == k ? xl[q] : zero; }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, th[k * TF + (m * KS + ks) * 64 + kg * 16 + n16]);
                        const bf16x8 wl = __builtin_bit_cast(bf16x8, tl[k * TF + (m * KS + ks) * 64 + kg * 16 + n16]);
#pragma unroll
                        for (int q = 0; q < PQ; ++q) {
                            acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ml[q], acc[q][m], 0, 0, 0);
                            acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, mh[q], acc[q][m], 0, 0, 0);
                            acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, mh[q], acc[q][m], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // lane (n16, kg) holds output channels 16 m + 4 kg + r of pixels p + q
        const bool copy_some = first && any_other;            // pixels of no slot / of a later pass: y = x
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * m + 4 * kg + r;
                fvec o;
#pragma unroll
                for (int q = 0; q < PQ; ++q) o[q] = acc[q][m][r] + (sl[q] >= 0 ? t0[sl[q] * N + co] : 0.f);
                if (copy_some) {
                    const fvec xin = *(const fvec*)(x + (size_t)co * L + p);
#pragma unroll
                    for (int q = 0; q < PQ; ++q) o[q] = sl[q] >= 0 ? o[q] : xin[q];
                }
                float* dstp = y + (size_t)co * L + p;
                if (first || all_mine) *(fvec*)dstp = o;
                else {
#pragma unroll
                    for (int q = 0; q < PQ; ++q) if (sl[q] >= 0) dstp[q] = o[q];
                }
            }
    }
}

// slots per statistics pass (KRES: 4 accumulator registers per 16x16 block, slot and wave) and per apply pass (KAPP: T tables in LDS)
template <int N> struct LabelCfg { static constexpr int NBLK = N / 32, KRES = N == 128 ? 1 : 8 / NBLK, KAPP = N == 128 ? 1 : (N == 64 ? 4 : 8); };


template <int N>
static int stats_labels(const float* x, long L, const uint8_t* mask, const LabelPlan* plan, double* stats, float* partial,
                        int max_slots, hipStream_t st) {
    using C = LabelCfg<N>;
    int per;
    const int G = cwct_stats_groups(L, &per);
    for (int slot0 = 0; slot0 < max_slots; slot0 += C::KRES) {
        cwct_stats_labels_kernel<C::NBLK, C::KRES><<<G, 256, 0, st>>>(x, L, mask, plan, slot0, partial, per);
        VST_RETURN_IF_LAUNCH_FAILED();
        cwct_stats_mean_kernel<<<dim3(N / 16, C::KRES), 256, 0, st>>>(partial, G, N, stats, C::KRES, slot0, &plan->n_slots);
        cwct_stats_cov_kernel<<<dim3(N * N / 16, C::KRES), 256, 0, st>>>(partial, G, N, stats, C::KRES, slot0, &plan->n_slots);
        VST_RETURN_IF_LAUNCH_FAILED();
    }
    return VST_OK;
}


template <int N>
static int apply_labels_split(const float* x, float* y, long L, const float* affines, const uint8_t* mask, const LabelPlan* plan,
                              int max_slots, hipStream_t st) {
    using C = LabelCfg<N>;
    constexpr int MB = N / 16, KS = N / 32, TF = MB * KS * 64, PQ = N >= 128 ? 2 : 4;
    const long niter = L / (16 * PQ);
    long wgs = (niter + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    const size_t lds = (size_t)C::KAPP * (2 * TF * 16 + N * sizeof(float));
    auto kern = cwct_apply_labels_split_kernel<N, C::KAPP>;
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)lds, &attr_done)) return rc_;
    for (int slot0 = 0; slot0 < max_slots; slot0 += C::KAPP) {
        kern<<<dim3((unsigned)wgs), 256, lds, st>>>(x, y, L, affines, mask, plan, slot0, niter);
        VST_RETURN_IF_LAUNCH_FAILED();
    }
    return VST_OK;
}

template <int N, int PXV>
static int apply_labels(const float* x, float* y, long L, const float* affines, const uint8_t* mask, const LabelPlan* plan,
                        int max_slots, hipStream_t st) {
    using C = LabelCfg<N>;
    constexpr int TS = N * (N + 1) + N;
    const long ngroups = (L + 32 * PXV - 1) / (32 * PXV);
    long wgs = (ngroups + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    const size_t lds = (size_t)(C::KAPP + 1) * TS * sizeof(float);
    auto kern = cwct_apply_labels_kernel<C::NBLK, PXV, C::KAPP>;
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)kern, (int)lds, &attr_done)) return rc_;
    for (int slot0 = 0; slot0 < max_slots; slot0 += C::KAPP) {
        kern<<<dim3((unsigned)wgs), 256, lds, st>>>(x, y, L, affines, mask, plan, slot0, ngroups);
        VST_RETURN_IF_LAUNCH_FAILED();
    }
    return VST_OK;
}


// ================================================================================================
// Pixel-major ("packed code") forms for N = 32.  The coupling state at the end of a photorealistic forward pass IS the
// code z, one 32-float row per full-resolution pixel: cell (h, w) of half i holds, contiguously, the 8 pixels
// (4h + 2i + i', 4w + 2j + j') (layout.hip, spread).  An unmasked cWCT does not care in which order the pixels come, so
// the statistics and the affine map run on the state itself and the spread / gather copies (2 x 256 MB per frame) go away.
// ================================================================================================

// Per-workgroup shifted sums of x[L][32] (same record as cwct_stats_mfma_kernel).  A lane's MFMA operand is one float it
// loads itself: lane (i = l & 31, h = l >> 5) of k-step t holds x[row 2t + h][channel i] - shift[i], A and B operand of
// v_mfma_f32_32x32x2_f32 are the same register (Q += v v^T over the two rows), no LDS in the loop.
// 16 waves per workgroup, 256 rows per wave at 1024 x 1024: at most 256 records for the combine kernel to read.
__global__ __launch_bounds__(1024) void cwct_stats_pm_kernel(const float* __restrict__ x, long L, float* __restrict__ partial,
                                                             int px_per_wg) {
    constexpr int N = 32, UNR = 16, NWV = 16;
    __shared__ float red[NWV / 2][17][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 31, h = lane >> 5;
    const long p_begin = (long)blockIdx.x * px_per_wg;
    long p_end = p_begin + px_per_wg;
    if (p_end > L) p_end = L;
    const float shift = p_begin < L ? x[(size_t)p_begin * N + ch] : 0.f;
    const long per_wave = px_per_wg / NWV;                  // px_per_wg is a multiple of 256: per_wave of 16
    long wb = p_begin + wave * per_wave, we = wb + per_wave;
    if (we > p_end) we = p_end;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    // two register sets: the loads of the next 32 rows are in flight during the MFMAs of the current ones.  Loads are
    // unconditional (a predicate per load would be a branch and a wait per load): rows past the end read the last row.
#define PM_LOAD(dst, base)                                                                            \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                 \
        const long row = (base) + 2 * u + h;                                                          \
        dst[u] = x[(size_t)(row < we ? row : we - 1) * N + ch];                                       \
    }
#define PM_USE(src, base)                                                                             \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                 \
        const float d = (base) + 2 * u + h < we ? src[u] - shift : 0.f;   /* we - wb is even: pairs are in or out together */ \
        asum += d;                                                                                    \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(d, d, acc, 0, 0, 0);                               \
    }
    if (wb < we) {
        float va[UNR], vb[UNR];
        PM_LOAD(va, wb)
        for (long r0 = wb; r0 < we; r0 += 4 * UNR) {
            PM_LOAD(vb, r0 + 2 * UNR)
            __builtin_amdgcn_sched_barrier(0);
            PM_USE(va, r0)
            __builtin_amdgcn_sched_barrier(0);
            PM_LOAD(va, r0 + 4 * UNR)
            __builtin_amdgcn_sched_barrier(0);
            PM_USE(vb, r0 + 2 * UNR)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef PM_LOAD
#undef PM_USE
    asum += __shfl_xor(asum, 32, 64);
    // fixed-order tree over the waves (8 + 4 + 2 + 1 rounds through LDS): bit-reproducible
#pragma unroll
    for (int half = NWV / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - half][r][lane] = acc[r];
            red[wave - half][16][lane] = asum;
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[wave][r][lane];
            asum += red[wave][16][lane];
        }
        __syncthreads();
    }
    float* rec = partial + (size_t)blockIdx.x * cwct_partial_stride(N);
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            rec[4 + 2 * N + (size_t)i * N + ch] = acc[r];
        }
        if (h == 0) {
            rec[4 + ch] = shift;
            rec[4 + N + ch] = asum;
        }
        if (lane == 0) rec[0] = p_end > p_begin ? (float)(p_end - p_begin) : 0.f;
    }
}

// Mean and covariance from the per-workgroup records in ONE launch (cwct_stats_mean_kernel + cwct_stats_cov_kernel, same
// arithmetic): workgroup = 16 covariance entries x 16 record strides; first the two means every entry needs, then the
// Chan et al. update of its co-moment, both in fp64; the workgroups owning column 0 also write the means and the count.
__global__ __launch_bounds__(256) void cwct_stats_finish_kernel(const float* __restrict__ partial, int G, int N,
                                                                double* __restrict__ stats) {
    __shared__ double sa[16][17], sb[16][17], sn[16][17];
    const int el = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int i = e / N, j = e - i * N;
    const size_t PS = cwct_partial_stride(N);
    double ai = 0.0, aj = 0.0, nt = 0.0;
#pragma unroll 4
    for (int g = gl; g < G; g += 16) {
        const float* rec = partial + (size_t)g * PS;
        const double n = rec[0];
        nt += n;
        ai += n * (double)rec[4 + i] + (double)rec[4 + N + i];
        aj += n * (double)rec[4 + j] + (double)rec[4 + N + j];
    }
    sa[gl][el] = ai; sb[gl][el] = aj; sn[gl][el] = nt;
    __syncthreads();
    double a2 = 0.0, b2 = 0.0, n2 = 0.0;
    for (int k = 0; k < 16; ++k) { a2 += sa[k][el]; b2 += sb[k][el]; n2 += sn[k][el]; }
    const double mu_i = n2 > 0.0 ? a2 / n2 : 0.0, mu_j = n2 > 0.0 ? b2 / n2 : 0.0;
    __syncthreads();
    double m2 = 0.0;
#pragma unroll 4
    for (int g = gl; g < G; g += 16) {
        const float* rec = partial + (size_t)g * PS;
        const float nf = rec[0], aif = rec[4 + N + i], ajf = rec[4 + N + j], si = rec[4 + i], sj = rec[4 + j], q = rec[4 + 2 * N + e];
        const double n = nf, pi = aif, pj = ajf;
        const double rn = nf > 0.f ? 1.0 / n : 0.0;
        const double di = (double)si + pi * rn - mu_i;
        const double dj = (double)sj + pj * rn - mu_j;
        m2 += nf > 0.f ? (double)q - pi * pj * rn + n * di * dj : 0.0;
    }
    sa[gl][el] = m2;
    __syncthreads();
    if (gl == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sa[k][el];
        stats[1 + N + e] = t / (n2 - 1.0);
        if (j == 0) stats[1 + i] = mu_i;
        if (e == 0) stats[0] = n2;
    }
}

// y[p][:] = T x[p][:] + t0 for the rows of one image's code, [2 halves][cells][8 groups][32].  One wave-tile = 32 rows:
// D[out channel][row] on 16 v_mfma_f32_32x32x2_f32 (exact fp32); A = T (lane (i, h) of k-step t: T[i][16h + t]), B = x
// (lane (row n, h): x[n][16h + t], i.e. 64 contiguous bytes per lane), so a lane ends with out channels 4h + 8q + {0..3},
// q = 0..3, of its row: float4 stores, and exactly the two 8-channel groups (q = 0, 2 and q = 1, 3) of the split-plane
// layout.  Rows of half 0 go to `planes0` (fp16 hi / lo, when given: the f16x2 inverse pass reads that half only through
// its planes) or to out0, rows of half 1 to out1.
__global__ __launch_bounds__(256) void cwct_apply_pm_kernel(const float* __restrict__ x, float* __restrict__ out0,
                                                            float* __restrict__ out1, unsigned char* __restrict__ planes0,
                                                            int Hq, int Wq, const float* __restrict__ affine, long tiles) {
    constexpr int N = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    float tfrag[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) tfrag[t] = affine[n * N + 16 * h + t];
    float4 t0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t0[q] = *(const float4*)(affine + N * N + 8 * q + 4 * h);
    const long rows_half = (long)Hq * Wq * 8;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
        const long row = tile * 32 + n;                      // a tile may straddle the halves or the end: decided per row
        const bool valid = row < 2 * rows_half;
        const float4* src = (const float4*)(x + (size_t)(valid ? row : 2 * rows_half - 1) * N + 16 * h);
        const float4 b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
        const float bv[16] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tfrag[t], bv[t], acc, 0, 0, 0);
        float o[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q][0] = acc[4 * q + 0] + t0[q].x; o[q][1] = acc[4 * q + 1] + t0[q].y;
            o[q][2] = acc[4 * q + 2] + t0[q].z; o[q][3] = acc[4 * q + 3] + t0[q].w;
        }
        const bool half1 = row >= rows_half;
        if (!valid) continue;
        if (!half1 && planes0 != nullptr) {
            const long cell = row >> 3;
            const int g = (int)(row & 7), y = (int)(cell / Wq), xx = (int)(cell - (long)y * Wq);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {                 // groups kg = h (q = 0, 2) and kg = 2 + h (q = 1, 3)
                const float f8[8] = {o[pr][0], o[pr][1], o[pr][2], o[pr][3], o[pr + 2][0], o[pr + 2][1], o[pr + 2][2], o[pr + 2][3]};
                u32x4 hi, lo;
                split8_sp(f8, hi, lo);
                const int cig = (g >> 1) * 8 + (g & 1) * 4 + 2 * pr + h;
                *(u32x4*)(planes0 + sp_offset(cig, 0, y, xx, Hq, Wq)) = hi;
                *(u32x4*)(planes0 + sp_offset(cig, 1, y, xx, Hq, Wq)) = lo;
            }
        } else {
            float* dst = half1 ? out1 + (size_t)(row - rows_half) * N : out0 + (size_t)row * N;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(float4*)(dst + 8 * q + 4 * h) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
        }
    }
}

// ---- artistic codes (N = 128, z = [B,128,H/2,W/2]): rows of 128 floats, 2 per quarter-resolution cell and half ----------------
// Statistics: the 128 x 128 co-moment as its 10 upper-triangular 32 x 32 blocks.  Lane (i, h) loads, per row pair, its float of
// each of the four 32-channel blocks (operands a_0..a_3); block (bi, bj) += a_bi a_bj^T on v_mfma_f32_32x32x2_f32.  8 waves per
// workgroup, two register sets of UNR row pairs in flight; the lower triangle is written as the mirror image.
__global__ __launch_bounds__(512) void cwct_stats_pm128_kernel(const float* __restrict__ x, long L, float* __restrict__ partial,
                                                               int px_per_wg) {
    constexpr int N = 128, NB = 4, NACC = 10, UNR = 4, NWV = 8;
    __shared__ float red[NWV / 2][16][64];
    __shared__ float reds[NWV][NB][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 31, h = lane >> 5;
    const long p_begin = (long)blockIdx.x * px_per_wg;
    long p_end = p_begin + px_per_wg;
    if (p_end > L) p_end = L;
    float shift[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) shift[b] = p_begin < L ? x[(size_t)p_begin * N + 32 * b + ch] : 0.f;
    const long per_wave = px_per_wg / NWV;
    long wb = p_begin + wave * per_wave, we = wb + per_wave;
    if (we > p_end) we = p_end;
    f32x16 acc[NACC];
    float asum[NB];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) asum[b] = 0.f;
#define PM128_LOAD(dst, base)                                                                         \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                 \
        const long row = (base) + 2 * u + h;                                                          \
        const float* p = x + (size_t)(row < we ? row : we - 1) * N + ch;                              \
        _Pragma("unroll") for (int b = 0; b < NB; ++b) dst[u][b] = p[32 * b];                         \
    }
#define PM128_USE(src, base)                                                                          \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                 \
        const bool in = (base) + 2 * u + h < we;                                                      \
        float d[NB];                                                                                  \
        _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                              \
            d[b] = in ? src[u][b] - shift[b] : 0.f;                                                   \
            asum[b] += d[b];                                                                          \
        }                                                                                             \
        int a_ = 0;                                                                                   \
        _Pragma("unroll") for (int bi = 0; bi < NB; ++bi)                                             \
            _Pragma("unroll") for (int bj = bi; bj < NB; ++bj, ++a_)                                  \
                acc[a_] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[bi], d[bj], acc[a_], 0, 0, 0);       \
    }
    if (wb < we) {
        float va[UNR][NB], vb[UNR][NB];
        PM128_LOAD(va, wb)
        for (long r0 = wb; r0 < we; r0 += 4 * UNR) {
            PM128_LOAD(vb, r0 + 2 * UNR)
            __builtin_amdgcn_sched_barrier(0);
            PM128_USE(va, r0)
            __builtin_amdgcn_sched_barrier(0);
            PM128_LOAD(va, r0 + 4 * UNR)
            __builtin_amdgcn_sched_barrier(0);
            PM128_USE(vb, r0 + 2 * UNR)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef PM128_LOAD
#undef PM128_USE
    float* rec = partial + (size_t)blockIdx.x * cwct_partial_stride(N);
    // row sums: one LDS round (fixed order), then the blocks one at a time through a fixed-order tree over the waves
#pragma unroll
    for (int b = 0; b < NB; ++b) reds[wave][b][lane] = asum[b] + __shfl_xor(asum[b], 32, 64);
    __syncthreads();
    if (wave == 0 && h == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float t = 0.f;
            for (int w = 0; w < NWV; ++w) t += reds[w][b][lane];
            rec[4 + N + 32 * b + ch] = t;
            rec[4 + 32 * b + ch] = shift[b];
        }
    }
    if (tid == 0) rec[0] = p_end > p_begin ? (float)(p_end - p_begin) : 0.f;
    int a_ = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = bi; bj < NB; ++bj, ++a_) {
#pragma unroll
            for (int half = NWV / 2; half >= 1; half >>= 1) {
                __syncthreads();
                if (wave >= half && wave < 2 * half) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[wave - half][r][lane] = acc[a_][r];
                }
                __syncthreads();
                if (wave < half) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a_][r] += red[wave][r][lane];
                }
            }
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * h, j = 32 * bj + ch;
                    rec[4 + 2 * N + (size_t)i * N + j] = acc[a_][r];
                    if (bi != bj) rec[4 + 2 * N + (size_t)j * N + i] = acc[a_][r];
                }
            }
        }
}

// y = T x + t0 on rows of 128: D[out block ob][row] on 64 v_mfma_f32_32x32x2_f32 per 32-channel out block and 32 rows (exact
// fp32); lane (row n, h) holds every other 16-byte piece of its row, T's fragments (64 KB, same channel order) sit in LDS.  Outputs as in cwct_apply_pm_kernel; a row is one of the
// two 128-channel groups of its cell, so its out block ob is 32-channel group g = 4 (row & 1) + ob of the split-plane layout.
__global__ __launch_bounds__(256) void cwct_apply_pm128_kernel(const float* __restrict__ x, float* __restrict__ out0,
                                                               float* __restrict__ out1, unsigned char* __restrict__ planes0,
                                                               int Hq, int Wq, const float* __restrict__ affine, long tiles) {
    constexpr int N = 128, NB = 4, KT = 64;
    extern __shared__ __attribute__((aligned(16))) float tl128[];        // [NB][KT][64]
    __shared__ __attribute__((aligned(16))) float t0s[N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    for (int idx = tid; idx < N * N; idx += 256) {          // coalesced read of T, scattered into fragment order
        const int i = idx >> 7, c = idx & 127;
        const int ob = i >> 5, hh = (c >> 2) & 1, t = 4 * (c >> 3) + (c & 3);
        tl128[(ob * KT + t) * 64 + (i & 31) + 32 * hh] = affine[idx];
    }
    if (tid < N) t0s[tid] = affine[N * N + tid];
    __syncthreads();
    const long rows_half = (long)Hq * Wq * 2;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
        const long row = tile * 32 + n;
        const bool valid = row < 2 * rows_half;
        // k-step t = 4 i + e of lane half h is channel 8 i + 4 h + e: the two halves read ADJACENT 16-byte pieces, so a 128-byte
        // line of the row is consumed by four consecutive load instructions (with 64 h + t the eight pieces of a line were 16
        // instructions apart and fell out of the 32 KB L1 in between: 142 us instead of 70)
        const float4* src = (const float4*)(x + (size_t)(valid ? row : 2 * rows_half - 1) * N) + h;
        float bv[KT];
#pragma unroll
        for (int i = 0; i < KT / 4; ++i) {
            const float4 v = src[2 * i];
            bv[4 * i] = v.x; bv[4 * i + 1] = v.y; bv[4 * i + 2] = v.z; bv[4 * i + 3] = v.w;
        }
        const bool half1 = row >= rows_half;
        const long rr = half1 ? row - rows_half : row;
        const long cell = rr >> 1;
        const int sub = (int)(rr & 1), y = (int)(cell / Wq), xx = (int)(cell - (long)y * Wq);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // T's fragments 16 k-steps at a time, the next 16 fetched from LDS during the current MFMAs (left to itself the
            // compiler waits for each ds_read in front of its MFMA: 133 us instead of 70 for a 512 x 512 code)
            float af[2][16];
#pragma unroll
            for (int t = 0; t < 16; ++t) af[0][t] = tl128[(ob * KT + t) * 64 + lane];
#pragma unroll
            for (int tb = 0; tb < KT / 16; ++tb) {
                __builtin_amdgcn_sched_barrier(0);
                if (tb + 1 < KT / 16) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) af[(tb + 1) & 1][t] = tl128[(ob * KT + 16 * (tb + 1) + t) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tb & 1][t], bv[16 * tb + t], acc, 0, 0, 0);
            }
            float o[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 tq = *(const float4*)&t0s[32 * ob + 8 * q + 4 * h];
                o[q][0] = acc[4 * q + 0] + tq.x; o[q][1] = acc[4 * q + 1] + tq.y;
                o[q][2] = acc[4 * q + 2] + tq.z; o[q][3] = acc[4 * q + 3] + tq.w;
            }
            if (!valid) continue;
            if (!half1 && planes0 != nullptr) {
                const int g = 4 * sub + ob;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float f8[8] = {o[pr][0], o[pr][1], o[pr][2], o[pr][3], o[pr + 2][0], o[pr + 2][1], o[pr + 2][2], o[pr + 2][3]};
                    u32x4 hi, lo;
                    split8_sp(f8, hi, lo);
                    const int cig = (g >> 1) * 8 + (g & 1) * 4 + 2 * pr + h;
                    *(u32x4*)(planes0 + sp_offset(cig, 0, y, xx, Hq, Wq)) = hi;
                    *(u32x4*)(planes0 + sp_offset(cig, 1, y, xx, Hq, Wq)) = lo;
                }
            } else {
                float* dst = (half1 ? out1 : out0) + (size_t)rr * N + 32 * ob;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(float4*)(dst + 8 * q + 4 * h) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
            }
        }
    }
}

// ---- masked forms on the packed rows -------------------------------------------------------------------------------------
// The label of row r of an image's code: rows are (half i, cell (h, w), group g = 4j + 2i' + j') <-> pixel (4h + 2i + i', 4w + 2j + j')
__global__ __launch_bounds__(256) void mask_to_code_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, int H, int W) {
    const int Hq = H >> 2, Wq = W >> 2;
    const long rows_half = (long)Hq * Wq * 8, total = 2 * rows_half;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < total; r += (long)gridDim.x * 256) {
        const int i = r >= rows_half;
        const long rr = r - i * rows_half;
        const long cell = rr >> 3;
        const int g = (int)(rr & 7), h = (int)(cell / Wq), w = (int)(cell - (long)h * Wq);
        const int y = 4 * h + 2 * i + ((g >> 1) & 1), x = 4 * w + 2 * (g >> 2) + (g & 1);
        out[r] = mask[(size_t)y * W + x];
    }
}

// Rows of a wave's 256-row window, bucketed by label slot: bucket[k][0 .. count[k]) = offsets (0..255) of the rows of slot k
// in ascending order; rows of no slot go to bucket NB - 1 when `keep_rest`, else nowhere.  Lane l owns rows l, l + 64, ...;
// ranks come from ballots + popcounts (NB x 4 of them per window), the bucket arrays are the wave's own LDS.
template <int NB>
__device__ __forceinline__ void bucket_rows(const uint8_t* __restrict__ mrow, long wbase, long wend, const unsigned char* lut,
                                            int slot0, int n_take, bool keep_rest, unsigned char (*bucket)[256], int* count) {
    const int lane = threadIdx.x & 63;
    int rel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long row = wbase + lane + 64 * i;
        int r = -1;
        if (row < wend) {
            const int sl = lut[mrow[row]];
            r = (sl != 255 && sl >= slot0 && sl < slot0 + n_take) ? sl - slot0 : (keep_rest ? NB - 1 : -1);
        }
        rel[i] = r;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        int run = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long bal = __ballot(rel[i] == k);
            if (rel[i] == k) bucket[k][run + __popcll(bal & lt)] = (unsigned char)(lane + 64 * i);
            run += __popcll(bal);
        }
        count[k] = run;                                      // wave-uniform
    }
}

// cwct_stats_pm_kernel for up to 8 label slots [slot0, slot0 + 8) in one pass over the rows.  Each wave buckets its rows by
// slot, 256 at a time; slot k's rows then feed accumulator set k two per MFMA, fetched in bucket order (a lane's operand is
// still one float it loads itself: which row is free) - about one window's worth of MFMAs per window however the labels are
// mixed.  8 waves per workgroup, records (workgroup, slot) as cwct_stats_labels_kernel writes them.
__global__ __launch_bounds__(512) void cwct_stats_labels_pm_kernel(const float* __restrict__ x, long L,
                                                                   const uint8_t* __restrict__ mrow,
                                                                   const LabelPlan* __restrict__ plan, int slot0,
                                                                   float* __restrict__ partial, int px_per_wg) {
    constexpr int N = 32, UNR = 16, NWV = 8, KRES = 8;
    __shared__ float red[NWV / 2][18][64];
    __shared__ unsigned char lut[256];
    __shared__ unsigned char buckets[NWV][KRES][256];
    if (slot0 >= plan->n_slots) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 31, h = lane >> 5;
    if (tid < 256) lut[tid] = plan->lut[tid];
    const long p_begin = (long)blockIdx.x * px_per_wg;
    long p_end = p_begin + px_per_wg;
    if (p_end > L) p_end = L;
    const float shift = p_begin < L ? x[(size_t)p_begin * N + ch] : 0.f;
    const long per_wave = px_per_wg / NWV;
    long wb = p_begin + wave * per_wave, we = wb + per_wave;
    if (we > p_end) we = p_end;
    f32x16 acc[KRES];
    float asum[KRES], cnt[KRES];
#pragma unroll
    for (int k = 0; k < KRES; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        asum[k] = 0.f; cnt[k] = 0.f;
    }
    __syncthreads();
    unsigned char (*bk)[256] = buckets[wave];
    for (long w0 = wb; w0 < we; w0 += 256) {
        int count[KRES];
        bucket_rows<KRES>(mrow, w0, we, lut, slot0, KRES, false, bk, count);
        __builtin_amdgcn_wave_barrier();                     // the wave's own LDS writes, before its reads below
#pragma unroll
        for (int k = 0; k < KRES; ++k) {
            const int nk = count[k];
            cnt[k] += (float)nk;                             // (every lane: the same number)
            for (int q0 = 0; q0 < nk; q0 += 2 * UNR) {       // 32 bucket positions = 16 MFMAs
                float v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int q = q0 + 2 * u + h;
                    v[u] = x[(size_t)(w0 + bk[k][q < nk ? q : nk - 1]) * N + ch];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (q0 + 2 * u < nk) {                   // uniform; an odd run's last pair has a zero partner
                        const float d = q0 + 2 * u + h < nk ? v[u] - shift : 0.f;
                        asum[k] += d;
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, d, acc[k], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // per slot: fixed-order tree over the 8 waves, then one record
#pragma unroll
    for (int k = 0; k < KRES; ++k) {
        float as = asum[k] + __shfl_xor(asum[k], 32, 64), cn = cnt[k];
#pragma unroll
        for (int half = NWV / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - half][r][lane] = acc[k][r];
                red[wave - half][16][lane] = as;
                red[wave - half][17][lane] = cn;
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][r] += red[wave][r][lane];
                as += red[wave][16][lane];
                cn += red[wave][17][lane];
            }
            __syncthreads();
        }
        if (wave == 0) {
            float* rec = partial + ((size_t)blockIdx.x * KRES + k) * cwct_partial_stride(N);
#pragma unroll
            for (int r = 0; r < 16; ++r) rec[4 + 2 * N + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * N + ch] = acc[k][r];
            if (h == 0) {
                rec[4 + ch] = shift;
                rec[4 + N + ch] = as;
            }
            if (lane == 0) rec[0] = cn;
        }
    }
}

// cwct_apply_pm_kernel with a map per row: y = T[slot(row)] x + t0[slot(row)] for up to 8 slots; rows of no slot keep x (the
// identity is the ninth "slot": exact on the fp32 MFMA).  A wave buckets 256 rows by slot and then runs 32-row tiles of ONE
// slot each (which rows form a tile is free: every lane loads and stores its own row), so the MFMA work is that of the unmasked
// kernel plus one partial tile per slot and window, however the labels are mixed.  T's fragments sit in LDS in lane order.
__global__ __launch_bounds__(256) void cwct_apply_labels_pm_kernel(const float* __restrict__ x, float* __restrict__ out0,
                                                                   float* __restrict__ out1, unsigned char* __restrict__ planes0,
                                                                   int Hq, int Wq, const float* __restrict__ affines,
                                                                   const uint8_t* __restrict__ mrow,
                                                                   const LabelPlan* __restrict__ plan, long windows,
                                                                   int max_slots) {
    constexpr int N = 32, KA = 8;
    extern __shared__ __attribute__((aligned(16))) float tl_dyn[];      // [max_slots][16][64]: only the slots in use
    float (*tl)[16][64] = (float (*)[16][64])tl_dyn;
    __shared__ __attribute__((aligned(16))) float t0s[KA + 1][N];
    __shared__ unsigned char lut[256];
    __shared__ unsigned char buckets[4][KA + 1][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int n_slots = plan->n_slots < max_slots ? plan->n_slots : max_slots;
    for (int idx = tid; idx < max_slots * 16 * 64; idx += 256) {
        const int k = idx >> 10, t = (idx >> 6) & 15, l = idx & 63;
        tl[k][t][l] = k < n_slots ? affines[(size_t)k * (N * N + N) + (l & 31) * N + 16 * (l >> 5) + t] : 0.f;
    }
    for (int idx = tid; idx < (KA + 1) * N; idx += 256) {
        const int k = idx / N;
        t0s[k][idx - k * N] = k < n_slots ? affines[(size_t)k * (N * N + N) + N * N + (idx - k * N)] : 0.f;
    }
    lut[tid] = plan->lut[tid];
    __syncthreads();
    const long rows_half = (long)Hq * Wq * 8, total = 2 * rows_half;
    unsigned char (*bk)[256] = buckets[wave];
    for (long win = (long)blockIdx.x * 4 + wave; win < windows; win += (long)gridDim.x * 4) {
        const long w0 = win * 256;
        int count[KA + 1];
        bucket_rows<KA + 1>(mrow, w0, total, lut, 0, n_slots, true, bk, count);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k <= KA; ++k) {
            const int nk = count[k];                         // (0 for every slot >= n_slots: their fragments are never read)
            for (int q0 = 0; q0 < nk; q0 += 32) {
                const bool valid = q0 + n < nk;
                const long row = w0 + bk[k][valid ? q0 + n : nk - 1];
                const float4* src = (const float4*)(x + (size_t)row * N + 16 * h);
                const float4 b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
                const float bv[16] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float a = k < KA ? tl[k < KA ? k : 0][t][lane] : (n == 16 * h + t ? 1.f : 0.f);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc, 0, 0, 0);
                }
                float o[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 tq = *(const float4*)&t0s[k][8 * q + 4 * h];
                    o[q][0] = acc[4 * q + 0] + tq.x; o[q][1] = acc[4 * q + 1] + tq.y;
                    o[q][2] = acc[4 * q + 2] + tq.z; o[q][3] = acc[4 * q + 3] + tq.w;
                }
                if (!valid) continue;
                const bool half1 = row >= rows_half;
                if (!half1 && planes0 != nullptr) {
                    const long cell = row >> 3;
                    const int g = (int)(row & 7), y = (int)(cell / Wq), xx = (int)(cell - (long)y * Wq);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const float f8[8] = {o[pr][0], o[pr][1], o[pr][2], o[pr][3], o[pr + 2][0], o[pr + 2][1], o[pr + 2][2], o[pr + 2][3]};
                        u32x4 hi, lo;
                        split8_sp(f8, hi, lo);
                        const int cig = (g >> 1) * 8 + (g & 1) * 4 + 2 * pr + h;
                        *(u32x4*)(planes0 + sp_offset(cig, 0, y, xx, Hq, Wq)) = hi;
                        *(u32x4*)(planes0 + sp_offset(cig, 1, y, xx, Hq, Wq)) = lo;
                    }
                } else {
                    float* dst = half1 ? out1 + (size_t)(row - rows_half) * N : out0 + (size_t)row * N;
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(float4*)(dst + 8 * q + 4 * h) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int vst3_apply_labels_code(const float* code, float* out0, float* out1, unsigned char* planes0, int H, int W,
                           const float* affines, const uint8_t* mask_rows, const void* plan, int max_slots, void* stream) {
    if (H < 8 || W < 8 || (H & 3) || (W & 3) || max_slots < 1 || max_slots > 8) return VST_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const long windows = ((long)H * W + 255) / 256;         // 256-row windows, one per wave at a time
    long wgs = (windows + 3) / 4;
    if (wgs > 4096) wgs = 4096;
    vst_prof_scope prof(VST_KERNEL_CWCT_APPLY, st);
    cwct_apply_labels_pm_kernel<<<dim3((unsigned)wgs), 256, (size_t)max_slots * 4096, st>>>(
        code, out0, out1, planes0, H >> 2, W >> 2, affines, mask_rows, (const LabelPlan*)plan, windows, max_slots);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

// internal (conv.hip's decode): out0 / out1 = where the transformed halves go, planes0 (nullable) = half 0 as split planes instead
int vst3_apply_code(const float* code, float* out0, float* out1, unsigned char* planes0, int H, int W, int sp_steps,
                    const float* affine, void* stream) {
    if (H < 8 || W < 8 || (H & 3) || (W & 3)) return VST_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_CWCT_APPLY, st);
    if (sp_steps == 2) {
        const long tiles = ((long)H * W + 31) / 32;         // 32-row tiles over the image's H * W rows
        long wgs = (tiles + 3) / 4;
        if (wgs > 8192) wgs = 8192;
        cwct_apply_pm_kernel<<<dim3((unsigned)wgs), 256, 0, st>>>(code, out0, out1, planes0, H >> 2, W >> 2, affine, tiles);
    } else if (sp_steps == 1) {
        const long tiles = ((long)H * W / 4 + 31) / 32;     // H * W / 4 rows of 128
        long wgs = (tiles + 3) / 4;
        if (wgs > 512) wgs = 512;                            // two per CU (64 KB of fragments each, staged once)
        static std::atomic<unsigned> attr_done{0};
        if (int rc = vst_ensure_dynamic_lds((const void*)cwct_apply_pm128_kernel, 65536, &attr_done)) return rc;
        cwct_apply_pm128_kernel<<<dim3((unsigned)wgs), 256, 65536, st>>>(code, out0, out1, planes0, H >> 2, W >> 2, affine, tiles);
    } else {
        return VST_E_MODE;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

VST_DEFINE_TU_RANGE(vst_range_tu_cwct)

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
    const bool vec = (L % 4) == 0 && ((uintptr_t)x % 16) == 0 && (mask == nullptr || ((uintptr_t)mask % 4) == 0);
    vst_prof_scope prof(VST_KERNEL_CWCT_STATS, st);
    switch (N) {
        case 16: cwct_stats_partial_kernel<1><<<G, 256, 0, st>>>(x, L, mask, label, partial, per); break;
        case 32: if (vec) cwct_stats_mfma_kernel<1, true><<<G, 256, 0, st>>>(x, L, mask, label, partial, per);
                 else cwct_stats_mfma_kernel<1, false><<<G, 256, 0, st>>>(x, L, mask, label, partial, per);
                 break;
        case 64: if (vec) cwct_stats_mfma_kernel<2, true><<<G, 256, 0, st>>>(x, L, mask, label, partial, per);
                 else cwct_stats_mfma_kernel<2, false><<<G, 256, 0, st>>>(x, L, mask, label, partial, per);
                 break;
        default: if (vec) cwct_stats_mfma_kernel<4, true><<<G, 256, 0, st>>>(x, L, mask, label, partial, per);
                 else cwct_stats_mfma_kernel<4, false><<<G, 256, 0, st>>>(x, L, mask, label, partial, per);
                 break;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    cwct_stats_mean_kernel<<<N / 16, 256, 0, st>>>(partial, G, N, stats, 1, 0, nullptr);
    VST_RETURN_IF_LAUNCH_FAILED();
    cwct_stats_cov_kernel<<<N * N / 16, 256, 0, st>>>(partial, G, N, stats, 1, 0, nullptr);
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
    const size_t lds = (size_t)N * N * 4 + (size_t)3 * N * 4 + 16;
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)cwct_factor_kernel<8>, (int)(80 * 1024), &attr_done)) return rc_;
    vst_prof_scope prof(VST_KERNEL_CWCT_FACTOR, st);
    switch (N) {
        case 16: cwct_factor_kernel<1><<<1, 256, lds, st>>>(a); break;
        case 32: cwct_factor_kernel<2><<<1, 256, lds, st>>>(a); break;
        case 64: cwct_factor_kernel<4><<<1, 256, lds, st>>>(a); break;
        default: cwct_factor_kernel<8><<<1, 256, lds, st>>>(a); break;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_prefactor(const double* stats, int N, float eps, double* out, int* info, void* stream) {
    if (!stats || !out || !info) return VST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (N) {
        case 16: cwct_prefactor_kernel<1><<<1, 256, 0, st>>>(stats, eps, out, info); break;
        case 32: cwct_prefactor_kernel<2><<<1, 256, 0, st>>>(stats, eps, out, info); break;
        case 64: cwct_prefactor_kernel<4><<<1, 256, 0, st>>>(stats, eps, out, info); break;
        case 128: cwct_prefactor_kernel<8><<<1, 256, 0, st>>>(stats, eps, out, info); break;
        default: return VST_E_SHAPE;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_apply_prec(const float* x, float* y, int N, long L, const float* affine, const uint8_t* mask, int label,
                        int precision, void* stream) {
    if (!x || !y || !affine || L <= 0) return VST_E_ARG;
    if (precision != VST_PREC_BF16X3 && precision != VST_PREC_FP32 && !vst_is_f16(precision)) return VST_E_MODE;
    const bool exact = precision == VST_PREC_FP32;
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_CWCT_APPLY, st);
    switch (N) {
        case 16: return launch_apply<16>(x, y, L, affine, mask, label, exact, st);
        case 32: return launch_apply<32>(x, y, L, affine, mask, label, exact, st);
        case 64: return launch_apply<64>(x, y, L, affine, mask, label, exact, st);
        case 128: return launch_apply<128>(x, y, L, affine, mask, label, exact, st);
        default: return VST_E_SHAPE;
    }
}

int vst_cwct_apply(const float* x, float* y, int N, long L, const float* affine, const uint8_t* mask, int label,
                   void* stream) {
    return vst_cwct_apply_prec(x, y, N, L, affine, mask, label, VST_PREC_BF16X3, stream);
}

// ---- single-pass masked transfer ----------------------------------------------------------------------------------------
int vst_label_plan(const uint8_t* cmask, long Lc, const uint8_t* smask, long Ls, void* plan, void* stream) {
    if (!cmask || !smask || !plan || Lc <= 0 || Ls <= 0) return VST_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    LabelPlan* p = (LabelPlan*)plan;
    hipError_t e = hipMemsetAsync(p, 0, sizeof(LabelPlan), st);
    if (e != hipSuccess) return (int)e;
    auto blocks = [](long L) { long b = (L + 256 * 16 - 1) / (256 * 16); return (unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); };
    label_hist_kernel<<<blocks(Lc), 256, 0, st>>>(cmask, Lc, p->hist_c);
    label_hist_kernel<<<blocks(Ls), 256, 0, st>>>(smask, Ls, p->hist_s);
    label_plan_kernel<<<1, 256, 0, st>>>(p);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

size_t vst_cwct_labels_workspace_bytes(int N, long L) {
    int per;
    const int g = cwct_stats_groups(L, &per);
    const int kres = N == 128 ? 1 : (N >= 32 ? 8 / (N / 32) : 8);
    return (size_t)g * kres * cwct_partial_stride(N) * sizeof(float);
}

int vst_cwct_stats_labels(const float* x, int N, long L, const uint8_t* mask, const void* plan, int max_slots, double* stats,
                          void* workspace, void* stream) {
    if (!x || !mask || !plan || !stats || L <= 0) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (max_slots <= 0 || max_slots > CWCT_MAX_SLOTS) max_slots = CWCT_MAX_SLOTS;
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_CWCT_STATS, st);
    const LabelPlan* p = (const LabelPlan*)plan;
    switch (N) {
        case 32: return stats_labels<32>(x, L, mask, p, stats, (float*)workspace, max_slots, st);
        case 64: return stats_labels<64>(x, L, mask, p, stats, (float*)workspace, max_slots, st);
        case 128: return stats_labels<128>(x, L, mask, p, stats, (float*)workspace, max_slots, st);
        default: return VST_E_SHAPE;
    }
}

int vst_cwct_factor_labels(const double* content_stats, const double* style_stats, const void* plan, int max_slots, float eps,
                           int N, float* affines, int* info, void* stream) {
    if (!content_stats || !style_stats || !plan || !affines || !info) return VST_E_ARG;
    if (!(N == 32 || N == 64 || N == 128)) return VST_E_SHAPE;
    if (max_slots <= 0 || max_slots > CWCT_MAX_SLOTS) max_slots = CWCT_MAX_SLOTS;
    FactorArgs a{};
    a.content = content_stats; a.styles[0] = style_stats; a.alphas[0] = 1.f; a.n_styles = 1; a.alpha_c = 0.f; a.eps = eps; a.N = N;
    a.affine = affines; a.info = info;
    a.content_stride = a.style_stride = 1 + N + (size_t)N * N; a.affine_stride = (size_t)N * N + N; a.info_stride = 3;
    a.n_slots = &((const LabelPlan*)plan)->n_slots;
    const size_t lds = (size_t)N * N * 4 + (size_t)3 * N * 4 + 16;
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<unsigned> attr_done{0};
    if (int rc_ = vst_ensure_dynamic_lds((const void*)cwct_factor_kernel<8>, (int)(80 * 1024), &attr_done)) return rc_;
    hipError_t e = hipMemsetAsync(info, 0, (size_t)max_slots * 3 * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    vst_prof_scope prof(VST_KERNEL_CWCT_FACTOR, st);
    switch (N) {
        case 32: cwct_factor_kernel<2><<<max_slots, 256, lds, st>>>(a); break;
        case 64: cwct_factor_kernel<4><<<max_slots, 256, lds, st>>>(a); break;
        default: cwct_factor_kernel<8><<<max_slots, 256, lds, st>>>(a); break;
    }
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_apply_labels(const float* x, float* y, int N, long L, const float* affines, const uint8_t* mask, const void* plan,
                          int max_slots, int precision, void* stream) {
    if (!x || !y || !affines || !mask || !plan || L <= 0) return VST_E_ARG;
    if (precision != VST_PREC_BF16X3 && precision != VST_PREC_FP32 && !vst_is_f16(precision)) return VST_E_MODE;
    if (max_slots <= 0 || max_slots > CWCT_MAX_SLOTS) max_slots = CWCT_MAX_SLOTS;
    hipStream_t st = (hipStream_t)stream;
    vst_prof_scope prof(VST_KERNEL_CWCT_APPLY, st);
    const LabelPlan* p = (const LabelPlan*)plan;
    // (in place is fine in every form: a wave reads its pixel group before it writes it)
    if (precision != VST_PREC_FP32 && (L % 64) == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0 && ((uintptr_t)mask % 4) == 0) {
        switch (N) {
            case 32: return apply_labels_split<32>(x, y, L, affines, mask, p, max_slots, st);
            case 64: return apply_labels_split<64>(x, y, L, affines, mask, p, max_slots, st);
            case 128: return apply_labels_split<128>(x, y, L, affines, mask, p, max_slots, st);
            default: return VST_E_SHAPE;
        }
    }
    const bool v4 = (L % 4) == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0;
    const bool v2 = (L % 2) == 0 && (((uintptr_t)x | (uintptr_t)y) % 8) == 0;
    switch (N) {
        case 32: return v4 ? apply_labels<32, 4>(x, y, L, affines, mask, p, max_slots, st)
                           : apply_labels<32, 1>(x, y, L, affines, mask, p, max_slots, st);
        case 64: return v2 ? apply_labels<64, 2>(x, y, L, affines, mask, p, max_slots, st)
                           : apply_labels<64, 1>(x, y, L, affines, mask, p, max_slots, st);
        case 128: return v2 ? apply_labels<128, 2>(x, y, L, affines, mask, p, max_slots, st)
                            : apply_labels<128, 1>(x, y, L, affines, mask, p, max_slots, st);
        default: return VST_E_SHAPE;
    }
}

size_t vst_cwct_stats_code_workspace_bytes(int H, int W, int sp_steps) {
    (void)H; (void)W;
    const int N = sp_steps == 1 ? 128 : 32;
    return (size_t)256 * cwct_partial_stride(N) * sizeof(float);      // at most 256 workgroup records
}

int vst_cwct_stats_code(const float* code, int H, int W, int sp_steps, double* stats, void* workspace, void* stream) {
    if (!code || !stats) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (H < 8 || W < 8 || (H & 3) || (W & 3)) return VST_E_SHAPE;
    if (sp_steps != 1 && sp_steps != 2) return VST_E_MODE;
    hipStream_t st = (hipStream_t)stream;
    const int N = sp_steps == 1 ? 128 : 32;
    const long L = sp_steps == 1 ? (long)H * W / 4 : (long)H * W;
    float* partial = (float*)workspace;                      // G <= 256 records: inside vst_cwct_stats_code_workspace_bytes
    int G;
    {
        vst_prof_scope prof(VST_KERNEL_CWCT_STATS, st);
        if (N == 32) {
            long per = ((L + 255) / 256 + 255) / 256 * 256;  // <= 256 workgroups of 16 waves, rows per workgroup a multiple of 256
            G = (int)((L + per - 1) / per);
            cwct_stats_pm_kernel<<<G, 1024, 0, st>>>(code, L, partial, (int)per);
        } else {
            long per = ((L + 255) / 256 + 127) / 128 * 128;  // <= 256 workgroups of 8 waves, 16-row multiples per wave
            G = (int)((L + per - 1) / per);
            cwct_stats_pm128_kernel<<<G, 512, 0, st>>>(code, L, partial, (int)per);
        }
        VST_RETURN_IF_LAUNCH_FAILED();
    }
    vst_prof_scope prof(VST_KERNEL_CWCT_FACTOR, st);
    cwct_stats_finish_kernel<<<N * N / 16, 256, 0, st>>>(partial, G, N, stats);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

int vst_cwct_apply_code(const float* code, float* out, int H, int W, int sp_steps, const float* affine, void* stream) {
    if (!code || !out || !affine) return VST_E_ARG;
    return vst3_apply_code(code, out, out + (size_t)H * W * 16, nullptr, H, W, sp_steps, affine, stream);
}

int vst_mask_to_code(const uint8_t* mask, uint8_t* mask_rows, int H, int W, void* stream) {
    if (!mask || !mask_rows) return VST_E_ARG;
    if (H < 8 || W < 8 || (H & 3) || (W & 3)) return VST_E_SHAPE;
    long blocks = ((long)H * W + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    mask_to_code_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(mask, mask_rows, H, W);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

size_t vst_cwct_stats_labels_code_workspace_bytes(int H, int W) {
    (void)H; (void)W;
    return (size_t)512 * 8 * cwct_partial_stride(32) * sizeof(float);     // at most 512 workgroups x 8 slots per pass
}

int vst_cwct_stats_labels_code(const float* code, int H, int W, const uint8_t* mask_rows, const void* plan, int max_slots,
                               double* stats, void* workspace, void* stream) {
    if (!code || !mask_rows || !plan || !stats) return VST_E_ARG;
    if (!workspace) return VST_E_WORKSPACE;
    if (H < 8 || W < 8 || (H & 3) || (W & 3)) return VST_E_SHAPE;
    if (max_slots <= 0 || max_slots > CWCT_MAX_SLOTS) max_slots = CWCT_MAX_SLOTS;
    hipStream_t st = (hipStream_t)stream;
    const long L = (long)H * W;
    const int N = 32, KRES = 8;
    long per = ((L + 511) / 512 + 2047) / 2048 * 2048;      // <= 512 workgroups of 8 waves, whole 256-row windows per wave
    const int G = (int)((L + per - 1) / per);
    float* partial = (float*)workspace;
    const LabelPlan* p = (const LabelPlan*)plan;
    vst_prof_scope prof(VST_KERNEL_CWCT_STATS, st);
    for (int slot0 = 0; slot0 < max_slots; slot0 += KRES) {
        cwct_stats_labels_pm_kernel<<<G, 512, 0, st>>>(code, L, mask_rows, p, slot0, partial, (int)per);
        VST_RETURN_IF_LAUNCH_FAILED();
        cwct_stats_mean_kernel<<<dim3(N / 16, KRES), 256, 0, st>>>(partial, G, N, stats, KRES, slot0, &p->n_slots);
        cwct_stats_cov_kernel<<<dim3(N * N / 16, KRES), 256, 0, st>>>(partial, G, N, stats, KRES, slot0, &p->n_slots);
        VST_RETURN_IF_LAUNCH_FAILED();
    }
    return VST_OK;
}

int vst_cwct_apply_labels_code(const float* code, float* out, int H, int W, const float* affines, const uint8_t* mask_rows,
                               const void* plan, int max_slots, void* stream) {
    if (!code || !out || !affines || !mask_rows || !plan) return VST_E_ARG;
    if (max_slots < 1 || max_slots > 8) return VST_E_SHAPE;               // one fragment set of 8 slots in LDS
    return vst3_apply_labels_code(code, out, out + (size_t)H * W * 16, nullptr, H, W, affines, mask_rows, plan, max_slots, stream);
}

}  // extern "C"

// Stage-3 / channel_reduction convolutions (64- and 256-channel tensors at quarter resolution, stride 1) fed by LDS-DMA:
// the VST_PREC_F16X2 path.
//
// Reference semantics: residual_block.conv of the 256-channel blocks, models/RevResNet.py:79-88 (ReflectionPad2d(1) +
// Conv2d 256->64, 64->64, 64->256, ReLU between), used by residual_block.forward/inverse :96-116.
//
// These three convs carry 75 % of the network's flops.  Two things differ from the generic kernels of conv.hip:
//
//  * Operands arrive already split.  Every activation tensor that one of these kernels reads is kept in HBM as two planes
//    of fp16 values, hi = round(x), lo = round(x - hi) ("SP" layout: [8-channel group][plane][y][x][8 x fp16], the same
//    4 bytes per value as fp32), written by the epilogue of the kernel that produced it (the fp32 state of the reversible
//    network is still what is read-modify-written; its SP copy is a shadow).  The 8 channels of a group are the ones a
//    producer lane holds, {c0 + 4 kg + r, c0 + 16 + 4 kg + r}; the weights' K order is permuted to match at pack time.
//    Staging is therefore a plain copy: global_load_lds_dwordx4 (64 lanes x 16 B = 1 KiB of the LDS image per wave
//    instruction, per-lane source address = reflection padding), no VGPRs, no conversion, no ds_write in the MFMA loop.
//  * fp16 operands, D += w_hi * (x_hi + x_lo): two MFMAs per product instead of the three of the bf16 split.  The
//    activations keep 22 bits; the weights are rounded to fp16 once (2^-12 relative), a fixed perturbation of the model that
//    forward and inverse share.  Measured end to end against the fp64 oracle: 9e-5 on the code, 3e-6 on the stylised frame
//    (budget 1e-3).
//
// Pipeline: a stage = one 32-channel chunk x all 9 taps = 9 k-steps (144 MFMAs per wave) between two barriers.  Both
// activation-chunk images and two weight-stage buffers live in LDS (156 KB); during stage s the waves issue the DMA of
// stage s+1's weights and of the next chunk's image, one piece per k-step between the MFMAs, and wait for them (vmcnt(0),
// by then most of a stage old) in front of the stage's closing raw s_barrier.
#include <atomic>
#include "common.h"

#ifndef VST_SP_BURST
#define VST_SP_BURST 1
#endif
#ifndef VST_SP_ABLATE
#define VST_SP_ABLATE 0      // timing-only builds: 1 = no DMA in the main loop, 2 = no MFMAs, 4 = no fragment re-reads,
                             // 16 = no deferred stores (the whole conv.7 epilogue is then dead code), 32 = no old-state loads
#endif

struct SpArgs {
    const unsigned char* in;     // SP planes of the input tensor (image 0)
    float* state;                // OUT_STATE: fp32 state half (ZC layout, level 2): the old values unless old_sp is given, and,
                                 // if store_f32, where the new values go
    const unsigned char* old_sp; // OUT_STATE: SP planes that hold the old state values (null: read them from `state`)
    int store_f32;               // OUT_STATE: write the new state as fp32 too (last writers of a half in a pass; lone blocks)
    unsigned char* out_sp;       // SP planes written by the epilogue: h1 / h2, or the new state (may be null)
    const unsigned char* wfrag;  // permuted-K fp16 weight fragments
    const float* bias;
    int H, W;                    // quarter-resolution image
    size_t in_img_bytes, out_img_bytes, state_img_floats;
    float sign;
    int tiles_x, tiles_y, tiles_total;
};

__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void sp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// IN1: the input tensor is ONE fp16 plane ([8-channel group][y][x][8 x fp16], the values rounded to fp16) instead of the hi / lo
// pair: one MFMA per product (VST_PREC_F16X2H: h2, the input of the 256-channel blocks' conv.7)
template <int CIN, int COUT, bool IN1 = false>
struct SpCfg {
    static constexpr int NW = 8;                             // waves, each owning 2 tile rows x 16 pixels x 64 output channels
    static constexpr int NCHUNK = CIN / 32, NCOT = COUT / 64, Q = NCHUNK * NCOT;
    static constexpr int MR = 2, TH = NW * MR, IW = 18, NPIX = (TH + 2) * IW;
    static constexpr int NSLOT = (NPIX + 15) / 16 * 16;      // multiple of 16: the four k-group planes start 256 B apart mod the 256-B bank row
    static constexpr int A_PLANE = 4 * NSLOT * 16, A_BUF = (IN1 ? 1 : 2) * A_PLANE;   // [plane][cig][slot][16 B]
    static constexpr int APIECES = A_BUF / 1024;             // DMA pieces (64 lanes x 16 B) per chunk image (42, or 21)
    // Only waves 0-3 issue DMA ("loaders"); their SIMD partners 4-7 run nothing but fragment reads and MFMAs, so the matrix
    // pipe of every SIMD has a wave to draw from while the other one is busy issuing pieces.
    static constexpr int NLOAD = 4;
    static constexpr int APW = (APIECES + NLOAD - 1) / NLOAD;                // image pieces per loader wave and chunk (11)
    static constexpr int WPIECES = 36, WPW = WPIECES / NLOAD;                // weight pieces per stage / per loader wave (9)
    static constexpr int B_BUF = 9 * 4 * 64 * 16;            // [k][kg][co][16 B] = 36864
    static constexpr int LDS_BYTES = 2 * A_BUF + 2 * B_BUF;  // 159744
    static_assert(A_BUF % 1024 == 0 && LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(APW <= 18 && WPW == 9, "one weight piece and at most two image pieces per k-step");
};

// INL: 0 = the input is a hi / lo pair (two MFMAs per product); 1 = a one-plane tensor; 2 = the HI plane of a hi / lo pair (the
// state planes as conv.1 of a 256-channel block reads them under VST_PREC_F16X2H).  1 and 2: one MFMA per product.
// OUT1 (kernels that write an intermediate, not the state): the output is written as one fp16 plane
template <int CIN, int COUT, bool OUT_STATE, int INL = 0, bool OUT1 = false>
__global__ __launch_bounds__(512, 2) void conv_sp_kernel(const SpArgs a) {
    constexpr bool IN1 = INL != 0;
    using C = SpCfg<CIN, COUT, IN1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + 2 * C::A_BUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, kg = lane >> 4;
    // XCD-aware tile order (see xcd_tile() in conv.hip)
    int bx, by, b;
    {
        const int g = blockIdx.x, per = gridDim.x >> 3;
        const int n = (g & 7) * per + (g >> 3);
        if (n >= a.tiles_total) return;
        bx = n % a.tiles_x;
        const int r = n / a.tiles_x;
        by = r % a.tiles_y;
        b = r / a.tiles_y;
    }
    const int tx0 = bx * 16, ty0 = by * C::TH, H = a.H, W = a.W;
#if VST_SP_ABLATE & 8
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* const stamp_lds = (unsigned long long*)(smem + C::LDS_BYTES);     // 40 words past the kernel's own LDS
    int stamp_n = 0;
#define SP_STAMP() { if (tid == 0 && stamp_n < 64) stamp_lds[stamp_n] = __builtin_amdgcn_s_memtime() - stamp_t0; ++stamp_n; }
#else
#define SP_STAMP()
#endif
    const unsigned char* const in_img = a.in + (size_t)b * a.in_img_bytes;
    const size_t chunk_bytes = (size_t)(INL == 1 ? 64 : 128) * H * W;   // 4 channel groups x 2 planes (or one) in HBM

    // ---- per-lane DMA source offsets ----------------------------------------------------------------------------
    // The activation image of one chunk (2 planes x 4 groups x NSLOT slots x 16 B) is APIECES pieces of 64 lanes, APW per
    // wave (pieces past the end repeat the last one: same bytes to the same place); a weight stage is 36 pieces (k, kg).
    unsigned a_off[C::APW], a_dst[C::APW];
#pragma unroll
    for (int u = 0; u < C::APW; ++u) {
        int i = wave * C::APW + u;
        i = i > C::APIECES - 1 ? C::APIECES - 1 : i;
        const int L = i * 64 + lane, pc = L / C::NSLOT;
        int slot = L - pc * C::NSLOT;
        slot = slot > C::NPIX - 1 ? C::NPIX - 1 : slot;
        const int iy = slot / C::IW, ix = slot - iy * C::IW;
        const int gy = reflect_clamp(ty0 - 1 + iy, H), gx = reflect_clamp(tx0 - 1 + ix, W);
        a_off[u] = INL == 1 ? (unsigned)sp_offset1(pc & 3, gy, gx, H, W)
                            : (unsigned)sp_offset(pc & 3, INL == 2 ? 0 : pc >> 2, gy, gx, H, W);
        a_dst[u] = i * 1024;
    }
    const bool loader = wave < C::NLOAD;
    // weight piece u of loader wave w: j = w * WPW + u (j = k*4 + kg): source (j * COUT + lane) * 16, destination j * 1024
    const unsigned w_off0 = (unsigned)((wave * C::WPW * COUT + lane) * 16);
    // piece u_ of the image of chunk chunk_ / of the weights of stage q_ (into weight buffer q_ & 1)
#define ISSUE_A1(chunk_, u_) \
    glds16(in_img + (size_t)(chunk_) * chunk_bytes + a_off[u_], Abuf + ((chunk_) & 1) * C::A_BUF + a_dst[u_])
#define ISSUE_W1(q_, u_)                                                                                          \
    {                                                                                                            \
        const int cot_ = (q_) / C::NCHUNK, chunk_ = (q_) - cot_ * C::NCHUNK;                                     \
        glds16(a.wfrag + ((size_t)chunk_ * 36 * COUT + cot_ * 64) * 16 + w_off0 + (u_) * (COUT * 16),                 \
               Bbuf + ((q_) & 1) * C::B_BUF + (wave * C::WPW + (u_)) * 1024);                                     \
    }

    // ---- prologue: chunk 0 and the weights of stage 0 ----------------------------------------------------------------
    if (loader) {
#pragma unroll
        for (int u = 0; u < C::APW; ++u) ISSUE_A1(0, u);
#pragma unroll
        for (int u = 0; u < C::WPW; ++u) ISSUE_W1(0, u);
    }
    sp_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 acc[C::MR][4];
#pragma unroll
    for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int slot_base = (wave * C::MR) * C::IW + lrow;
    const int oy0 = ty0 + wave * C::MR, ox = tx0 + lrow;
    const bool full_tile = ty0 + C::TH <= H && tx0 + 16 <= W;
    float* const st_img = OUT_STATE && a.state ? a.state + (size_t)b * a.state_img_floats : nullptr;
    const unsigned char* const old_img = OUT_STATE && a.old_sp ? a.old_sp + (size_t)b * a.out_img_bytes : nullptr;
    const bool store_f32 = OUT_STATE && a.store_f32;
    unsigned char* const sp_img = a.out_sp ? a.out_sp + (size_t)b * a.out_img_bytes : nullptr;

    struct Frags { f16x8 w[4], xh[C::MR], xl[C::MR]; };
    // k = 3 * dy + dx: tap (dy, dx) of the chunk
    auto read_frags = [&](Frags& f, const unsigned char* Ab, const unsigned char* Bb, int k) {
        const int dy = k / 3, dx = k - dy * 3;
#pragma unroll
        for (int n = 0; n < 4; ++n) f.w[n] = *(const f16x8*)(Bb + (k * 256 + n * 16) * 16);
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
            f.xh[m] = *(const f16x8*)(Ab + ((m + dy) * C::IW + dx) * 16);
            if (!IN1) f.xl[m] = *(const f16x8*)(Ab + C::A_PLANE + ((m + dy) * C::IW + dx) * 16);
        }
    };

    // Epilogue of a 64-channel output slice, in two halves: `finish` turns the accumulators into the slice's results
    // (res = old + sign * (acc + bias), or ReLU(acc + bias)) right after the slice's last MFMA; the stores (fp32 state and /
    // or split planes) are DEFERRED into the k-steps of the NEXT stage, one unit per k-step between its MFMAs, so that a
    // slice boundary costs no store-issue time and no barrier skew.  The last slice flushes its units at once.
    float4 bias[4], old[C::MR][4];
    f32x4 res[C::MR][4];
    bool pending = false;                                  // uniform: res holds a slice whose stores are still to be issued
    int pend_cot = 0, pend_slot0 = 0;                      // its slice; the k-step slots already used for it
    // unit u = 0..7: fp32 state store (m = u / 4, n = u % 4);  unit 8 + u2, u2 = 0..3: plane pair (m = u2 / 2, j = u2 % 2)
#define STORE_UNIT(u_)                                                                                                        \
    {                                                                                                                         \
        if ((u_) < 8) {                                                                                                       \
            const int m_ = (u_) / 4, n_ = (u_) % 4, oy_ = oy0 + m_;                                                           \
            if (store_f32 && (full_tile || (oy_ < H && ox < W)))                                                              \
                *(float4*)(st_img + (((unsigned)oy_ * W + ox) * 256u + pend_cot * 64 + n_ * 16 + 4 * kg)) =                   \
                    make_float4(res[m_][n_][0], res[m_][n_][1], res[m_][n_][2], res[m_][n_][3]);                              \
        } else {                                                                                                              \
            const int m_ = ((u_) - 8) / 2, j_ = ((u_) - 8) % 2, oy_ = oy0 + m_;                                               \
            if (sp_img && (full_tile || (oy_ < H && ox < W))) {                                                               \
                const float f8_[8] = {res[m_][2 * j_][0], res[m_][2 * j_][1], res[m_][2 * j_][2], res[m_][2 * j_][3],        \
                                      res[m_][2 * j_ + 1][0], res[m_][2 * j_ + 1][1], res[m_][2 * j_ + 1][2],                 \
                                      res[m_][2 * j_ + 1][3]};                                                                \
                u32x4 hi_, lo_;                                                                                               \
                float unused_ = 0.f;                        /* (range check: at `finish`, once per slice) */                  \
                split8_sp(f8_, hi_, lo_, unused_);                                                                            \
                const int cig_ = pend_cot * 8 + j_ * 4 + kg;                                                                  \
                if (OUT1) {                                                                                                   \
                    *(u32x4*)(sp_img + sp_offset1(cig_, oy_, ox, H, W)) = hi_;                                                \
                } else {                                                                                                      \
                    *(u32x4*)(sp_img + sp_offset(cig_, 0, oy_, ox, H, W)) = hi_;                                              \
                    *(u32x4*)(sp_img + sp_offset(cig_, 1, oy_, ox, H, W)) = lo_;                                              \
                }                                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
    }

#pragma unroll 1
    for (int q = 0; q < C::Q; ++q) {
        const int cot = q / C::NCHUNK, chunk = q - cot * C::NCHUNK;
        const bool slice_end = chunk == C::NCHUNK - 1;
        // ---- one stage before a slice's last (so that they are two stages old when used): its bias and old state values ------
        if (slice_end) {                                      // (the bias: a few cached bytes, one stage ahead is plenty)
#pragma unroll
            for (int n = 0; n < 4; ++n) bias[n] = *(const float4*)(a.bias + cot * 64 + n * 16 + 4 * kg);
        }
        // Its old state values too, HERE: at a stage's top no DMA is in flight (an ordinary load issued while LDS-DMA is
        // pending makes hipcc drain the whole queue first), the previous slice's deferred stores went out a stage ago (both
        // share one ~24 B/clk pipe), and the values are a whole stage old when the slice ends.  Pixels past the image edge read
        // a clamped, valid address and are never stored: no per-load predicate (the compiler would branch around every load and
        // wait for each one).  Plane loads are kept as raw bits (hi in old[m][2j], lo in old[m][2j+1]) and decoded at the slice
        // end: touching them here would be a wait.
        if (OUT_STATE && slice_end && !(VST_SP_ABLATE & 32)) {
            const int oxc = ox < W ? ox : W - 1;
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int oyc = oy0 + m < H ? oy0 + m : H - 1;
                    if (old_img == nullptr) {
                        old[m][2 * j] = *(const float4*)(st_img + (((unsigned)oyc * W + oxc) * 256u + cot * 64 + (2 * j) * 16 + 4 * kg));
                        old[m][2 * j + 1] = *(const float4*)(st_img + (((unsigned)oyc * W + oxc) * 256u + cot * 64 + (2 * j + 1) * 16 + 4 * kg));
                    } else {
                        const int cig = cot * 8 + j * 4 + kg;
                        old[m][2 * j] = *(const float4*)(old_img + sp_offset(cig, 0, oyc, oxc, H, W));
                        old[m][2 * j + 1] = *(const float4*)(old_img + sp_offset(cig, 1, oyc, oxc, H, W));
                    }
                }
            asm volatile("" ::: "memory");
        }
        SP_STAMP();
        const bool issue_w = q + 1 < C::Q;                    // next stage's weights -> the buffer stage q-1 used
        const bool issue_a = q + 1 < C::NCHUNK;               // next chunk's image (first output slice only) -> other image buffer
        const unsigned char* Ab = Abuf + (chunk & 1) * C::A_BUF + (kg * C::NSLOT + slot_base) * 16;
        const unsigned char* Bb = Bbuf + (q & 1) * C::B_BUF + (kg * 64 + lrow) * 16;
        Frags fr[2];
        read_frags(fr[0], Ab, Bb, 0);
        // One-term kernels (INL != 0): a stage's MFMAs (2.3 K cycles) are shorter than the DMA latency, so a piece issued at
        // the stage's last k-step is waited for in full; the next stage's pieces all go out HERE, at the stage's top (their
        // buffers were released by the barrier that ended stage q-1), and have the whole stage to land.
        constexpr bool BURST = VST_SP_BURST && IN1;
        if (BURST && !(VST_SP_ABLATE & 1) && loader) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (issue_w) ISSUE_W1(q + 1, k);
                if (issue_a && k < C::APW) ISSUE_A1(q + 1, k);
            }
        }
        // ---- 9 k-steps; fragments double-buffered in registers; one DMA piece of each kind per k-step ----------------------
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            if (k < 8 && !(VST_SP_ABLATE & 4)) read_frags(fr[(k + 1) & 1], Ab, Bb, k + 1);
            if (!BURST && !(VST_SP_ABLATE & 1) && loader) {
                if (issue_w) ISSUE_W1(q + 1, k);
                if (issue_a) {
                    if (k < C::APW) ISSUE_A1(q + 1, k);       // (a one-plane image has 6 pieces per loader wave, fewer than k-steps)
                    if (9 + k < C::APW) ISSUE_A1(q + 1, 9 + k);
                }
            }
            if (pending && !(VST_SP_ABLATE & 16)) {                                   // the previous slice's stores: ONE unit per k-step (the memory pipe
                constexpr int order[12] = {8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7};   // takes ~24 B/clk per CU: more would stall the
                if (pend_slot0 == 0) { STORE_UNIT(order[k]); }                      // waves at issue); plane pairs first
                else if (k < 3) { STORE_UNIT(order[9 + (k < 3 ? k : 0)]); }
            }
            __builtin_amdgcn_sched_barrier(0);
            const Frags& f = fr[(VST_SP_ABLATE & 4) ? 0 : (k & 1)];
            if (VST_SP_ABLATE & 2) {
                asm volatile("" ::"v"(f.w[0]), "v"(f.w[3]), "v"(f.xh[0]), "v"(f.xl[1]));
                continue;
            }
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (!IN1) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[n], f.xl[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[n], f.xh[m], acc[m][n], 0, 0, 0);
                }
            if (k == 2 || k == 5) SP_STAMP();
        }
        pend_slot0 += 9;
        if (pend_slot0 >= 12) pending = false;
        SP_STAMP();
        sp_wait_vm<0>();       // this stage's DMA and deferred stores (issued in its first k-steps); at a slice end also the bias /
                               // old state values fetched a stage earlier
        SP_STAMP();
        if (slice_end) {
            if (OUT_STATE && old_img != nullptr) {           // old = hi + lo of the lane's 8 channels per (m, j)
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f16x8 hi = __builtin_bit_cast(f16x8, old[m][2 * j]), lo = __builtin_bit_cast(f16x8, old[m][2 * j + 1]);
                        old[m][2 * j] = make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1],
                                                    (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
                        old[m][2 * j + 1] = make_float4((float)hi[4] + (float)lo[4], (float)hi[5] + (float)lo[5],
                                                        (float)hi[6] + (float)lo[6], (float)hi[7] + (float)lo[7]);
                    }
            }
#pragma unroll
            for (int m = 0; m < C::MR; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float bb[4] = {bias[n].x, bias[n].y, bias[n].z, bias[n].w};
                    const float oo[4] = {old[m][n].x, old[m][n].y, old[m][n].z, old[m][n].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[m][n][e] + bb[e];
                        res[m][n][e] = OUT_STATE ? oo[e] + a.sign * v : (v > 0.f ? v : 0.f);
                    }
                    acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            if (sp_img) {                                    // these values are about to be rounded to fp16: range flag, once per slice
                float amax = 0.f;
#pragma unroll
                for (int m = 0; m < C::MR; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(res[m][n][0])), fabsf(res[m][n][1])),
                                     fmaxf(fabsf(res[m][n][2]), fabsf(res[m][n][3])));
                vst_note_range(amax);
            }
            pending = true;
            pend_cot = cot;
            pend_slot0 = 0;
            if (q == C::Q - 1) {                             // nothing left to hide the stores behind
#pragma unroll
                for (int u = 0; u < 12; ++u) STORE_UNIT(u);
            }
            asm volatile("" ::: "memory");
        }
        SP_STAMP();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        SP_STAMP();
    }
#undef STORE_UNIT
#undef ISSUE_A1
#undef ISSUE_W1
#if VST_SP_ABLATE & 8
    // diagnostic build only: shader cycles and 100 MHz ticks of one mid-grid workgroup overwrite the head of the output planes
    if (blockIdx.x == gridDim.x / 2 && tid == 0) {
        unsigned long long* d = OUT_STATE ? (unsigned long long*)a.state : (unsigned long long*)a.out_sp;
        d[0] = __builtin_amdgcn_s_memtime() - stamp_t0;
        d[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        for (int i = 0; i < 64; ++i) d[2 + i] = i < stamp_n ? stamp_lds[i] : 0;
    }
#endif
}

// fp32 state half (ZC, level 2: [y][x][256]) -> its SP shadow.  One wave = 64 consecutive x of one 8-channel group.
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ state, unsigned char* __restrict__ sp,
                                                      int B, int H, int W) {
    const size_t total = (size_t)B * 32 * H * W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = idx % W;
        size_t rest = idx / W;
        const int y = rest % H; rest /= H;
        const int cig = rest % 32;
        const int b = rest / 32;
        const int cot = cig >> 3, j = (cig >> 2) & 1, kg = cig & 3;
        const float* p = state + (((size_t)b * H + y) * W + x) * 256 + cot * 64 + j * 32 + kg * 4;
        const float4 v0 = *(const float4*)p, v1 = *(const float4*)(p + 16);
        const float f8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        u32x4 hi, lo;
        split8_sp(f8, hi, lo);
        unsigned char* o = sp + (size_t)b * 1024 * H * W;
        *(u32x4*)(o + sp_offset(cig, 0, y, x, H, W)) = hi;
        *(u32x4*)(o + sp_offset(cig, 1, y, x, H, W)) = lo;
    }
}

template <int CIN, int COUT, bool OUT_STATE, int INL = 0, bool OUT1 = false>
static int launch_sp(SpArgs a, int B, hipStream_t st) {
    using C = SpCfg<CIN, COUT, INL != 0>;
    auto kern = conv_sp_kernel<CIN, COUT, OUT_STATE, INL, OUT1>;
    static std::atomic<unsigned> attr_done{0};
    if (int rc = vst_ensure_dynamic_lds((const void*)kern, C::LDS_BYTES + ((VST_SP_ABLATE & 8) ? 512 : 0), &attr_done)) return rc;
    a.tiles_x = (a.W + 15) / 16; a.tiles_y = (a.H + C::TH - 1) / C::TH; a.tiles_total = a.tiles_x * a.tiles_y * B;
    vst_prof_scope prof(VST_KERNEL_ID(CIN, COUT, 1), st);
    kern<<<dim3((a.tiles_total + 7) / 8 * 8), 64 * C::NW, C::LDS_BYTES + ((VST_SP_ABLATE & 8) ? 512 : 0), st>>>(a);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

static const unsigned char* sp_frag(const vst_conv_weights& c, int cout, int cin) {
    const PackedConvLayout L = packed_conv_layout(cout, cin);
    return (const unsigned char*)c.packed + L.f32_bytes + 2 * L.frag_bytes;
}

int vst3_conv_mid(const vst_conv_weights* c, const void* in_sp, void* out_sp, int out_single, int B, int H, int W, void* stream) {
    const int Hq = H >> 2, Wq = W >> 2;
    SpArgs a{};
    a.H = Hq; a.W = Wq;
    a.in = (const unsigned char*)in_sp; a.out_sp = (unsigned char*)out_sp;
    a.in_img_bytes = (size_t)Hq * Wq * 64 * 4;
    a.out_img_bytes = (size_t)Hq * Wq * 64 * (out_single ? 2 : 4);
    a.wfrag = sp_frag(*c, 64, 64); a.bias = c->bias;
    return out_single ? launch_sp<64, 64, false, 0, true>(a, B, (hipStream_t)stream)
                      : launch_sp<64, 64, false>(a, B, (hipStream_t)stream);
}

int vst3_conv_out(const vst_conv_weights* c, const void* in_sp, int in_single, float* state, void* out_sp, float sign, int B,
                  int H, int W, void* stream) {
    const int Hq = H >> 2, Wq = W >> 2;
    SpArgs a{};
    a.H = Hq; a.W = Wq; a.state_img_floats = (size_t)Hq * Wq * 256;
    a.in = (const unsigned char*)in_sp; a.in_img_bytes = (size_t)Hq * Wq * 64 * (in_single ? 2 : 4);
    a.state = state; a.old_sp = nullptr; a.store_f32 = 1;
    a.out_sp = (unsigned char*)out_sp; a.out_img_bytes = (size_t)Hq * Wq * 256 * 4;
    a.wfrag = sp_frag(*c, 256, 64); a.bias = c->bias; a.sign = sign;
    return in_single ? launch_sp<64, 256, true, 1>(a, B, (hipStream_t)stream) : launch_sp<64, 256, true>(a, B, (hipStream_t)stream);
}

VST_DEFINE_TU_RANGE(vst_range_tu_conv3)

// fp32 half state [B][H/4][W/4][256] -> its split planes
int vst3_presplit(const float* state, unsigned char* planes, int B, int H, int W, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Hq = H >> 2, Wq = W >> 2;
    const size_t total = (size_t)B * 32 * Hq * Wq;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    vst_prof_scope prof(VST_KERNEL_PRESPLIT, st);
    presplit_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(state, planes, B, Hq, Wq);
    VST_RETURN_IF_LAUNCH_FAILED();
    return VST_OK;
}

// the split-plane buffer `idx` (0 / 1) inside a pass's scratch (h1 | h2 | planes 0 | planes 1)
unsigned char* vst3_plane_buffer(void* tmp, int idx, int B, int H, int W) {
    const size_t px = (size_t)(H >> 2) * (W >> 2);
    return (unsigned char*)tmp + (size_t)B * px * 64 * 4 * 2 + (size_t)idx * B * px * 256 * 4;
}

int vst3_block256(const vst_block_weights* w, int direction, int precision, float* dst, const float* src, void* tmp,
                  int pos, int src_planes_ready, int B, int H, int W, void* stream) {
    if (!vst_is_f16(precision)) return VST_E_MODE;
    const bool h2_single = precision == VST_PREC_F16X2H;    // h2 as one fp16 plane: conv.7 issues one MFMA per product
    hipStream_t st = (hipStream_t)stream;
    const int Hq = H >> 2, Wq = W >> 2;
    const size_t mid_bytes = (size_t)Hq * Wq * 64 * 4, state_bytes = (size_t)Hq * Wq * 256 * 4;
    unsigned char* h1 = (unsigned char*)tmp;
    unsigned char* h2 = h1 + (size_t)B * mid_bytes;
    // two split-plane buffers: inside a pass's run of 256-channel blocks the state lives ONLY there (position pos = 0..10 of the
    // run; pos < 0 = a block on its own).  A block's src planes are the buffer its predecessor wrote, its dst planes the other.
    unsigned char* spbuf[2] = {h2 + (size_t)B * mid_bytes, h2 + (size_t)B * mid_bytes + (size_t)B * state_bytes};
    const bool alone = pos < 0;
    const int p = alone ? 0 : pos;
    unsigned char* sp_src = spbuf[p & 1];
    unsigned char* sp_dst = spbuf[(p & 1) ^ 1];
    if (p == 0 && !src_planes_ready) {   // first block of the run: the planes of its src come from the fp32 state
        if (int rc0 = vst3_presplit(src, sp_src, B, H, W, stream)) return rc0;
    }
    SpArgs a{};
    a.H = Hq; a.W = Wq; a.state_img_floats = (size_t)Hq * Wq * 256;
    auto frag = [](const vst_conv_weights& c, int cout, int cin) {
        const PackedConvLayout L = packed_conv_layout(cout, cin);
        return (const unsigned char*)c.packed + L.f32_bytes + 2 * L.frag_bytes;
    };
    // conv.1: planes(src) -> h1
    a.in = sp_src; a.in_img_bytes = state_bytes; a.out_sp = h1; a.out_img_bytes = mid_bytes; a.state = nullptr;
    a.wfrag = frag(w->conv[0], 64, 256); a.bias = w->conv[0].bias; a.sign = 0.f;
    // f16x2h: conv.1 reads only the hi plane of the state (one MFMA per product, half the image bytes; the state itself keeps both)
    if (h2_single) a.out_img_bytes = mid_bytes / 2;         // (h1 as one fp16 plane too)
    int rc = h2_single ? launch_sp<256, 64, false, 2, true>(a, B, st) : launch_sp<256, 64, false>(a, B, st);
    if (rc) return rc;
    // conv.4: h1 -> h2
    a.in = h1; a.in_img_bytes = h2_single ? mid_bytes / 2 : mid_bytes; a.out_sp = h2; a.out_img_bytes = h2_single ? mid_bytes / 2 : mid_bytes;
    a.wfrag = frag(w->conv[1], 64, 64); a.bias = w->conv[1].bias;
    rc = h2_single ? launch_sp<64, 64, false, 1, true>(a, B, st) : launch_sp<64, 64, false>(a, B, st);
    if (rc) return rc;
    // conv.7: h2 -> dst += sign * (.).  The old dst values: fp32 for the first block of a run, afterwards the planes of this
    // block's dst buffer (pos 1: block 0's src planes, from block 20 / the gather; later: what block pos-2 wrote; read before
    // written, same lane).  The new values go to
    // the planes (the next block's src / the block after's old values) except for the run's last block, and to the fp32 state
    // for the last writer of each half (pos >= 9) and for a block on its own.
    a.in = h2; a.in_img_bytes = h2_single ? mid_bytes / 2 : mid_bytes; a.out_img_bytes = state_bytes; a.state = dst;
    a.old_sp = (!alone && p >= 1) ? sp_dst : nullptr;
    a.store_f32 = alone || p >= 9;
    a.out_sp = (!alone && p < 10) ? sp_dst : nullptr;
    a.wfrag = frag(w->conv[2], 256, 64); a.bias = w->conv[2].bias;
    a.sign = direction > 0 ? 1.f : -1.f;
    return h2_single ? launch_sp<64, 256, true, 1>(a, B, st) : launch_sp<64, 256, true>(a, B, st);
}

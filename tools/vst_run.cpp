// vst_run — native host runner for libvstnet_hip.so (no Python, no torch): the MI355X counterpart of the reference
// fork's ggml engine entry point (project/ggml/include/ggml_engine.h:610 GGMLNetwork::engine_forward,
// project/ggml/src/vstmodel.h:433-688 VSTEncoder/VSTDecoder), driving the same C ABI the Python classes bind.
//
//   vst_run weights.bin content.rgb H W style.rgb Hs Ws out.rgb
//
// weights.bin: vstnet_amd/export.py; *.rgb: raw uint8 HWC frames.  Pipeline (image_transfer.py:172-201):
// encode content + style (uint8 edge on the device), cWCT statistics -> prefactor(style) -> factor -> apply, decode.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vstnet.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define VST_CALL(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, vst_error_string(rc_)); return 3; } } while (0)

static bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n);
    const bool ok = fread(buf.data(), 1, n, f) == (size_t)n;
    fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc != 9 && argc != 10) {
        fprintf(stderr, "usage: %s weights.bin content.rgb H W style.rgb Hs Ws out.rgb [f16x2h|f16x2|bf16x3|fp32]\n", argv[0]);
        return 1;
    }
    int prec = VST_PREC_BF16X3;                   // the drop-in classes' default (fp32-class; the fp16 modes are opt-in)
    if (argc == 10) {
        if (!strcmp(argv[9], "bf16x3")) prec = VST_PREC_BF16X3;
        else if (!strcmp(argv[9], "fp32")) prec = VST_PREC_FP32;
        else if (!strcmp(argv[9], "f16x2h")) prec = VST_PREC_F16X2H;
        else if (!strcmp(argv[9], "f16x2")) prec = VST_PREC_F16X2;
        else { fprintf(stderr, "unknown precision %s\n", argv[9]); return 1; }
    }
    const int H = atoi(argv[3]), W = atoi(argv[4]), Hs = atoi(argv[6]), Ws = atoi(argv[7]);
    std::vector<uint8_t> wfile, content, style;
    if (!read_file(argv[1], wfile) || !read_file(argv[2], content) || !read_file(argv[5], style)) { fprintf(stderr, "cannot read inputs\n"); return 1; }
    if (wfile.size() < 16 || memcmp(wfile.data(), "VSTW", 4) != 0) { fprintf(stderr, "bad weights file\n"); return 1; }
    int hdr[3];
    memcpy(hdr, wfile.data() + 4, 12);
    const int sp = hdr[2];
    if ((size_t)H * W * 3 != content.size() || (size_t)Hs * Ws * 3 != style.size()) { fprintf(stderr, "frame size mismatch\n"); return 1; }
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    // ---- weights: 32 blocks x 3 convs, shapes of models/RevResNet.py:72-88 ---------------------------------------
    vst_net_weights net;
    const float* wp = (const float*)(wfile.data() + 16);
    std::vector<void*> keep;
    for (int k = 0; k < VST_NUM_BLOCKS; ++k) {
        const int ch = k < 10 ? 16 : (k < 20 ? 64 : 256), stride = (k == 10 || k == 20) ? 2 : 1;
        const int cin[3] = {stride == 1 ? ch : ch / 4, ch / 4, ch / 4}, cout[3] = {ch / 4, ch / 4, ch};
        float* dw[3];
        float* db[3];
        for (int c = 0; c < 3; ++c) {
            const size_t nw = (size_t)cout[c] * cin[c] * 9;
            HIP_OK(hipMalloc(&dw[c], nw * 4));
            HIP_OK(hipMalloc(&db[c], cout[c] * 4));
            HIP_OK(hipMemcpyAsync(dw[c], wp, nw * 4, hipMemcpyHostToDevice, st)); wp += nw;
            HIP_OK(hipMemcpyAsync(db[c], wp, cout[c] * 4, hipMemcpyHostToDevice, st)); wp += cout[c];
            keep.push_back(dw[c]);
        }
        // the same exponent normalisation of h1 / h2 the Python classes apply at pack time (exact, function-preserving)
        VST_CALL(vst_normalize_block(dw[0], db[0], dw[1], db[1], dw[2], cin[0], cout[0], cout[2], nullptr, st));
        for (int c = 0; c < 3; ++c) {
            void* packed;
            HIP_OK(hipMalloc(&packed, vst_conv_packed_bytes(cout[c], cin[c])));
            VST_CALL(vst_pack_conv(dw[c], cout[c], cin[c], packed, st));
            net.blocks[k].conv[c].packed = packed;
            net.blocks[k].conv[c].bias = db[c];
        }
    }
    if ((const uint8_t*)wp != wfile.data() + wfile.size()) { fprintf(stderr, "weights file has the wrong length\n"); return 1; }

    // ---- buffers ----------------------------------------------------------------------------------------------------
    const int N = sp == 2 ? 32 : 128;
    const long L = sp == 2 ? (long)H * W : (long)H * W / 4, Ls = sp == 2 ? (long)Hs * Ws : (long)Hs * Ws / 4;
    uint8_t *d_c, *d_s, *d_out;
    float *z_c, *z_s, *affine;
    double *st_c, *st_s;
    int* info;
    void *ws, *cws;
    const size_t wbytes = vst_pass_workspace_bytes(1, H > Hs ? H : Hs, W > Ws ? W : Ws);
    HIP_OK(hipMalloc(&d_c, content.size())); HIP_OK(hipMalloc(&d_s, style.size())); HIP_OK(hipMalloc(&d_out, content.size()));
    HIP_OK(hipMalloc(&z_c, (size_t)N * L * 4)); HIP_OK(hipMalloc(&z_s, (size_t)N * Ls * 4));
    HIP_OK(hipMalloc(&affine, ((size_t)N * N + N) * 4));
    HIP_OK(hipMalloc(&st_c, (1 + N + (size_t)N * N) * 8)); HIP_OK(hipMalloc(&st_s, (1 + N + (size_t)N * N) * 8));
    HIP_OK(hipMalloc(&info, 16)); HIP_OK(hipMalloc(&ws, wbytes));
    size_t cwb = vst_cwct_stats_workspace_bytes(N, L), cwb2 = vst_cwct_stats_workspace_bytes(N, Ls);
    if (vst_cwct_stats_code_workspace_bytes(H, W, sp) > cwb) cwb = vst_cwct_stats_code_workspace_bytes(H, W, sp);
    HIP_OK(hipMalloc(&cws, cwb > cwb2 ? cwb : cwb2));
    HIP_OK(hipMemcpyAsync(d_c, content.data(), content.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_s, style.data(), style.size(), hipMemcpyHostToDevice, st));

    // ---- the hot path ---------------------------------------------------------------------------------------------------
    const double* styles[1] = {st_s};
    const float alphas[1] = {1.f};
    // the code stays in the coupling blocks' layout ("Packed code" in vstnet.h): no spread / gather, the statistics run on the
    // packed rows and the affine map is applied while the inverse pass loads its state
    VST_CALL(vst_revnet_encode_u8(&net, d_c, z_c, ws, 1, H, W, prec, st));
    VST_CALL(vst_revnet_encode_u8(&net, d_s, z_s, ws, 1, Hs, Ws, prec, st));
    VST_CALL(vst_cwct_stats_code(z_s, Hs, Ws, sp, st_s, cws, st));
    VST_CALL(vst_cwct_prefactor(st_s, N, 2e-5f, st_s, info, st));
    VST_CALL(vst_cwct_stats_code(z_c, H, W, sp, st_c, cws, st));
    VST_CALL(vst_cwct_factor(st_c, styles, alphas, 1, 0.f, 2e-5f, N, affine, info, st));
    VST_CALL(vst_revnet_decode_u8(&net, z_c, affine, d_out, ws, 1, H, W, sp, prec, st));

    std::vector<uint8_t> out(content.size());
    HIP_OK(hipMemcpyAsync(out.data(), d_out, out.size(), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    unsigned range = 0;
    VST_CALL(vst_range_flags(&range, 0));
    if (range) fprintf(stderr, "vst_run: WARNING fp16 range flags 0x%x (1 = an activation saturated at +-65504, 2 = a weight beyond fp16): "
                               "rerun with bf16x3\n", range);
    FILE* f = fopen(argv[8], "wb");
    if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[8]); return 1; }
    fclose(f);
    printf("vst_run: %dx%d stylised with a %dx%d style (%s), wrote %s\n", W, H, Ws, Hs, sp == 2 ? "photorealistic" : "artistic", argv[8]);
    return 0;
}

// api.hip -- C ABI of libsi_hip.so (include/si_hip.h): context, checkpoint packing, forward orchestration.
//
// Host-side work done here, once per checkpoint load:
//   * weight-norm folding, w = g * v / ||v||  (generator convs: norm over all dims but 0,
//     I_ea/hifi_gan/models.py:125-132; positional conv: over all dims but 2, modeling_hubert.py:78)
//   * re-layout of every Conv1d / ConvTranspose1d / Linear weight into the tap-GEMM form
//     W[group][tap][n][ci] (ci contiguous), q/k/v fused into one (3H, H) matrix, ConvTranspose1d split into
//     its `stride` output phases (n = phase*Cout + co, taps q = 0..k/stride-1 read input row u-q)
//   * conversion to the arithmetic of the model desc (fp32 / bf16 / bf16 hi+lo)
//   * codebook tables: centred centroids, 1/||centred||, raw centroids (I_ea/loss_fn.py:10-14)
// The result is ONE packed device blob whose layout depends only on the model desc, so ranks that did not
// read the checkpoint can receive it by a single RCCL broadcast (si_alloc_weights + si_weights_device_ptr).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct HostTensor {
    const float* data = nullptr;
    std::vector<long> shape;
    long numel() const { long n = 1; for (long d : shape) n *= d; return n; }
};

struct GemmW {              // one tap-GEMM weight inside the packed blob
    size_t w = 0, w_lo = 0, bias = 0;   // byte offsets (bias: 0 = none)
    int Cin = 0, N = 0, Npad = 0, ntaps = 0, groups = 1, math = 0;
    bool has_bias = false;
};

struct LayerW { GemmW qkv, out, ffn1, ffn2; size_t ln1_g, ln1_b, ln2_g, ln2_b; };
struct ConvW { GemmW g; size_t ln_g = 0, ln_b = 0; };
struct ResW { GemmW c1[SI_MAX_DIL], c2[SI_MAX_DIL]; };

struct Layout {
    size_t total = 0;
    // encoder
    size_t conv0_w, conv0_bias, conv0_g, conv0_b;
    std::vector<ConvW> convs;            // conv layers 1..n-1
    size_t fp_ln_g, fp_ln_b; GemmW proj;
    GemmW pos; size_t enc_ln_g, enc_ln_b;
    std::vector<LayerW> layers;
    size_t head_ln_g, head_ln_b; GemmW head;
    size_t cb_centered, cb_raw, cb_rnorm;
    // vocoder
    GemmW pre; std::vector<GemmW> ups; std::vector<ResW> rbs;   // rbs[stage*num_rb + j]
    size_t post_w, post_b; int post_C = 0;
    int mel_ld = 0;                      // padded mel channel count (conv_pre Cin)
};

}  // namespace

struct si_ctx {
    int device = 0;
    si_model_desc d{};
    char err[512] = {0};
    Layout lay;
    char* wdev = nullptr;
    bool weights_ready = false;
    bool weights_verified = false;           // the blob's layout fingerprint has been compared with this context's (si_weights_check)
    std::map<std::string, long> dbg_size;                        // floats of each intermediate of the last forward
    std::map<std::string, std::pair<float*, long>> dbg_capture;  // name -> (device dst, capacity in floats)
    // per-launch HIP-event timing (si_profile_start / si_profile_stop)
    struct ProfRec { int name; hipEvent_t a, b; double flops, bytes; };
    bool prof_on = false;
    std::string prof_filter;             // non-empty: only launches of this kernel family are bracketed
    std::vector<std::string> prof_names;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
    int prof_open = -1;
    int num_cus = 0;
    std::map<const void*, size_t> dyn_lds;   // per kernel: dynamic-LDS limit already raised on this context's device
    // arithmetic-path options, read from the environment when the context is created (all default to 1)
    bool opt_voc_opready = true, opt_voc_res16 = true, opt_enc_opready = true, opt_att_bf16 = true, opt_enc_lingemm = true;
    bool opt_enc_posconv = true;             // the positional conv on posconv.hip in the bf16 encoder mode (SI_ENC_POSCONV=0: the generic tap-GEMM)
    int opt_ln_fuse = 1;                     // post-LN layers, bf16 mode: LayerNorm outputs read as residuals are recomputed in the GEMM epilogues (SI_ENC_LNFUSE=0: written)
    int opt_voc_upsgemm = 1;                 // the generator's early upsamplers on gemmcu.hip's TC instantiations (SI_VOC_UPSGEMM=0: the tap-GEMM)
    int opt_ffn_pad = 64;                    // elements of padding behind each row of the bf16 FFN intermediate (SI_ENC_FFNPAD; multiple of 8, <= 128):
                                             // rows 6144 bytes apart are 6272 apart instead -- FFN2 -1.5 % (profiles/r04_ffnpad_ab.txt), same values
    int opt_gemmcu = 1;                      // encoder GEMMs as one tile per CU (gemmcu.hip): 0 never, 1 by the shape rule, 2 whenever the shape allows, 10 + c (A/B)
    int opt_gemm256 = 1;                     // encoder GEMMs on 256 x 256 tiles: 0 never, 1 by the shape rule, 2 whenever the shape allows (tests)
    int opt_voc_chain = 1;                   // whole-resblock kernel on the C = 32 stage (SI_VOC_CHAIN=0: one launch per conv pair)
    int opt_voc_fuse = 1;                    // 0: never, 1: every covered width, otherwise a mask of the channel counts to fuse (32 | 64 | 128 | 256)
    // constant tables of the mel front-end (built on first use): DFT matrix [Npad][n_fft] = rows cos | -sin, periodic
    // Hann window, transposed Slaney mel basis with the non-zero bin span of every band
    char* fe_dev = nullptr;
    size_t fe_dft = 0, fe_hann = 0, fe_basis = 0, fe_lo = 0, fe_hi = 0;
    // ragged batches: per-clip length tables are built on the host (launch grids depend on them) and reach the device through a
    // small ring of pinned staging slots (an async copy from pageable memory may still be reading its source after the call returns)
    std::vector<int32_t> vl_host;            // the table of the call in progress (launchers read it during the call only)
    static constexpr int VL_SLOTS = 4;
    int32_t* vl_pin[VL_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t vl_pin_ints[VL_SLOTS] = {0, 0, 0, 0};
    hipEvent_t vl_ev[VL_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int vl_next = 0;
    float* km_cnorm = nullptr;               // |c_k|^2 scratch of si_kmeans_assign (K floats; library-owned, stream-ordered reuse)
    int km_cnorm_cap = 0;
};

static char g_create_err[512] = "";

int si_fail(si_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}
int si_fail_hip(si_ctx* ctx, hipError_t e, const char* what, const char* file, int line) {
    return si_fail(ctx, SI_EHIP, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
}

// Test hook: remember the size of a named intermediate and, when a capture is registered for it, copy it out
// the moment it is produced (workspace buffers are recycled later in the same forward).
static int si_tap(si_ctx* ctx, const char* name, const float* src, long n, hipStream_t st) {
    ctx->dbg_size[name] = n;
    auto it = ctx->dbg_capture.find(name);
    if (it == ctx->dbg_capture.end()) return SI_OK;
    const long m = n < it->second.second ? n : it->second.second;
    if (m > 0) SI_HIP_CHECK(hipMemcpyAsync(it->second.first, src, (size_t)m * 4, hipMemcpyDeviceToDevice, st));
    return SI_OK;
}

// Per-launch timing with HIP events recorded on the launch stream (the same stream the kernel runs on).
// Called by every si_launch_* around its kernel; a no-op unless si_profile_start armed it.
void si_prof_begin(si_ctx* ctx, const char* name, double flops, double bytes, hipStream_t st) {
    if (!ctx->prof_on || ctx->prof_used + 2 > ctx->prof_pool.size()) { ctx->prof_open = -1; return; }
    if (!ctx->prof_filter.empty() && ctx->prof_filter != name) { ctx->prof_open = -1; return; }
    int id = -1;
    for (size_t i = 0; i < ctx->prof_names.size(); ++i) if (ctx->prof_names[i] == name) { id = (int)i; break; }
    if (id < 0) { id = (int)ctx->prof_names.size(); ctx->prof_names.push_back(name); }
    si_ctx::ProfRec r{id, ctx->prof_pool[ctx->prof_used], ctx->prof_pool[ctx->prof_used + 1], flops, bytes};
    ctx->prof_used += 2;
    (void)hipEventRecord(r.a, st);
    ctx->prof_open = (int)ctx->prof_recs.size();
    ctx->prof_recs.push_back(r);
}
const char* si_prof_shape_name(const char* name, long M, int N, int K) {
    static const bool on = getenv("SI_PROF_SHAPES") && atoi(getenv("SI_PROF_SHAPES")) != 0;
    if (!on) return name;
    static thread_local char buf[48];
    snprintf(buf, sizeof(buf), "%s_%ldx%dx%d", name, M, N, K);
    return buf;
}
int si_ensure_dyn_lds(si_ctx* ctx, const void* kern, size_t bytes) {
    if (bytes <= 64 * 1024) return SI_OK;
    size_t& have = ctx->dyn_lds[kern];
    if (bytes > have) {
        SI_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return SI_OK;
}
int si_opt_gemm256(const si_ctx* ctx) { return ctx->opt_gemm256; }
int si_opt_gemmcu(const si_ctx* ctx) { return ctx->opt_gemmcu; }
int si_num_cus(si_ctx* ctx) {
    if (ctx->num_cus <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        ctx->num_cus = n;
    }
    return ctx->num_cus;
}
// host table (ctx->vl_host) -> device `dst`, stream-ordered, through the next pinned slot
static int vl_upload(si_ctx* ctx, int32_t* dst, hipStream_t st) {
    const size_t n = ctx->vl_host.size();
    const int slot = ctx->vl_next;
    ctx->vl_next = (slot + 1) % si_ctx::VL_SLOTS;
    if (!ctx->vl_ev[slot]) SI_HIP_CHECK(hipEventCreateWithFlags(&ctx->vl_ev[slot], hipEventDisableTiming));
    else SI_HIP_CHECK(hipEventSynchronize(ctx->vl_ev[slot]));                  // the copy that last used this slot has finished
    if (ctx->vl_pin_ints[slot] < n) {
        if (ctx->vl_pin[slot]) SI_HIP_CHECK(hipHostFree(ctx->vl_pin[slot]));
        ctx->vl_pin[slot] = nullptr; ctx->vl_pin_ints[slot] = 0;
        SI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->vl_pin[slot]), std::max<size_t>(n, 4096) * sizeof(int32_t), hipHostMallocDefault));
        ctx->vl_pin_ints[slot] = std::max<size_t>(n, 4096);
    }
    memcpy(ctx->vl_pin[slot], ctx->vl_host.data(), n * sizeof(int32_t));
    SI_HIP_CHECK(hipMemcpyAsync(dst, ctx->vl_pin[slot], n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SI_HIP_CHECK(hipEventRecord(ctx->vl_ev[slot], st));
    return SI_OK;
}
void si_prof_end(si_ctx* ctx, hipStream_t st) {
    if (ctx->prof_open < 0) return;
    (void)hipEventRecord(ctx->prof_recs[ctx->prof_open].b, st);
    ctx->prof_open = -1;
}

// ------------------------------------------------------------------------------------------------ layout
namespace {

size_t elem_bytes(int math) { return math == SI_MATH_F32 ? 4 : 2; }

struct Planner {
    size_t cur = 0;
    size_t take(size_t bytes) { size_t o = cur; cur += (bytes + 255) / 256 * 256; return o; }
    size_t floats(size_t n) { return take(n * 4); }
    GemmW gemm(int math, int groups, int ntaps, int N, int Cin, bool bias) {
        GemmW g;
        g.math = math; g.groups = groups; g.ntaps = ntaps; g.N = N; g.Cin = Cin; g.has_bias = bias;
        g.Npad = si_round_up(N, si_pick_bn(N));
        const size_t n = (size_t)groups * ntaps * g.Npad * Cin;
        g.w = take(n * elem_bytes(math));
        if (math == SI_MATH_BF16X3) g.w_lo = take(n * 2);
        if (bias) g.bias = floats((size_t)groups * N);
        return g;
    }
};

// Channel count of upsampling stage i as the kernels see it.  On the fp16 activation stream a stage narrower than 32 channels
// (I_da's unit vocoder ends on 16) is carried PADDED to 32: its weights, biases and therefore activations are zero in the extra
// channels, and the stage runs on the 32-channel chain kernel (three launches per stage) instead of 18 tap-GEMM launches that
// move the stream at a tenth of the HBM rate; `conv_post` reads the padded rows with zero weights.  Other modes keep the real width
// (their per-stage taps are compared against the reference's tensors).
int stage_channels(const si_ctx* ctx, int i) {
    const int real = ctx->d.up_initial_channel >> (i + 1);
    const bool r16 = ctx->opt_voc_opready && ctx->opt_voc_res16 && ctx->d.vocoder_math == SI_MATH_F16;
    return (r16 && real < 32 && real >= 4) ? 32 : real;
}

int plan_layout(si_ctx* ctx) {
    const si_model_desc& d = ctx->d;
    Layout& L = ctx->lay;
    Planner P;
    P.take(256);                                           // offset 0 is reserved to mean "absent"
    const int em = d.encoder_math, vm = d.vocoder_math;
    const int H = d.hidden_size;
    // encoder
    L.conv0_w = P.floats((size_t)d.conv_dim[0] * d.conv_kernel[0]);
    L.conv0_bias = d.conv_bias ? P.floats(d.conv_dim[0]) : 0;
    L.conv0_g = P.floats(d.conv_dim[0]);
    L.conv0_b = P.floats(d.conv_dim[0]);
    L.convs.clear();
    for (int i = 1; i < d.num_conv; ++i) {
        ConvW c;
        c.g = P.gemm(em, 1, d.conv_kernel[i], d.conv_dim[i], d.conv_dim[i - 1], d.conv_bias != 0);
        if (d.feat_norm_layer) { c.ln_g = P.floats(d.conv_dim[i]); c.ln_b = P.floats(d.conv_dim[i]); }
        L.convs.push_back(c);
    }
    const int CF = d.conv_dim[d.num_conv - 1];
    L.fp_ln_g = P.floats(CF); L.fp_ln_b = P.floats(CF);
    L.proj = P.gemm(em, 1, 1, H, CF, true);
    L.pos = P.gemm(em, d.pos_conv_groups, d.pos_conv_kernel, H / d.pos_conv_groups, H / d.pos_conv_groups, true);
    L.enc_ln_g = P.floats(H); L.enc_ln_b = P.floats(H);
    L.layers.clear();
    for (int l = 0; l < d.num_layers; ++l) {
        LayerW w;
        w.qkv = P.gemm(em, 1, 1, 3 * H, H, true);
        w.out = P.gemm(em, 1, 1, H, H, true);
        w.ln1_g = P.floats(H); w.ln1_b = P.floats(H);
        w.ffn1 = P.gemm(em, 1, 1, d.intermediate_size, H, true);
        w.ffn2 = P.gemm(em, 1, 1, H, d.intermediate_size, true);
        w.ln2_g = P.floats(H); w.ln2_b = P.floats(H);
        L.layers.push_back(w);
    }
    L.head_ln_g = P.floats(H); L.head_ln_b = P.floats(H);
    L.head = P.gemm(SI_MATH_F32, 1, 1, d.codebook_dim, H, true);       // the arg-max input stays fp32
    L.cb_centered = P.floats((size_t)d.num_clusters * d.codebook_dim);
    L.cb_raw = P.floats((size_t)d.num_clusters * d.codebook_dim);
    L.cb_rnorm = P.floats(d.num_clusters);
    // vocoder
    L.mel_ld = si_round_up(d.num_mels, 32);
    const int C0 = d.up_initial_channel;
    L.pre = P.gemm(vm, 1, 7, C0, L.mel_ld, true);
    L.ups.clear(); L.rbs.clear();
    int c = C0;
    for (int i = 0; i < d.num_ups; ++i) {
        const int u = d.up_rates[i], k = d.up_kernels[i], cout = stage_channels(ctx, i);
        L.ups.push_back(P.gemm(vm, 1, (k + u - 1) / u, u * cout, c, true));   // taps q = j div u; absent (phase, tap) pairs stay zero
        c = cout;
        for (int j = 0; j < d.num_rb; ++j) {
            ResW r;
            for (int n = 0; n < d.num_dil; ++n) {
                r.c1[n] = P.gemm(vm, 1, d.rb_kernels[j], c, c, true);
                if (d.resblock_type != 2) r.c2[n] = P.gemm(vm, 1, d.rb_kernels[j], c, c, true);   // ResBlock2 has one conv per dilation
            }
            L.rbs.push_back(r);
        }
    }
    L.post_C = c;
    L.post_w = P.floats((size_t)7 * c);
    L.post_b = P.floats(1);
    L.total = P.cur;
    return SI_OK;
}

// The packed blob's layout is a function of the model desc AND of the context's arithmetic-path options (stage_channels: the
// padded widths of the fp16 stream).  A rank that receives the blob by broadcast must have planned the SAME layout as the rank
// that packed it; the first 32 bytes of the blob (offset 0 is reserved) carry a fingerprint of the plan so that a mismatch is an
// error instead of silently mis-read weights.
struct BlobHeader { uint32_t magic, version; uint64_t fingerprint, total; uint64_t reserved; };
constexpr uint32_t BLOB_MAGIC = 0x42574953u;   // "SIWB"
uint64_t layout_fingerprint(const si_ctx* ctx) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    mix(&ctx->d, sizeof(ctx->d));
    const Layout& L = ctx->lay;
    const uint64_t total = L.total; mix(&total, 8);
    for (int i = 0; i < ctx->d.num_ups; ++i) { const int c = stage_channels(ctx, i); mix(&c, 4); mix(&L.ups[i].w, sizeof(size_t)); }
    mix(&L.post_w, sizeof(size_t)); mix(&L.post_C, 4); mix(&L.cb_raw, sizeof(size_t)); mix(&L.head.w, sizeof(size_t));
    return h;
}
BlobHeader blob_header(const si_ctx* ctx) { return BlobHeader{BLOB_MAGIC, (uint32_t)SI_ABI_VERSION, layout_fingerprint(ctx), (uint64_t)ctx->lay.total, 0}; }

int check_desc(si_ctx* ctx, const si_model_desc* d) {
    if (!d || d->struct_size != (int32_t)sizeof(si_model_desc))
        return si_fail(ctx, SI_EINVAL, "si_model_desc size mismatch (got %d, library expects %zu)", d ? d->struct_size : -1,
                       sizeof(si_model_desc));
    if (d->num_conv < 2 || d->num_conv > SI_MAX_CONV) return si_fail(ctx, SI_EINVAL, "num_conv=%d out of range", d->num_conv);
    if (d->num_ups < 1 || d->num_ups > SI_MAX_UPS) return si_fail(ctx, SI_EINVAL, "num_ups=%d out of range", d->num_ups);
    if (d->num_rb < 1 || d->num_rb > SI_MAX_RB || d->num_dil < 1 || d->num_dil > SI_MAX_DIL)
        return si_fail(ctx, SI_EINVAL, "resblock shape %dx%d out of range", d->num_rb, d->num_dil);
    if (d->hidden_size <= 0 || d->num_heads <= 0 || d->hidden_size != d->num_heads * 64)
        return si_fail(ctx, SI_EINVAL, "hidden_size=%d / heads=%d: the attention kernel needs head_dim 64", d->hidden_size, d->num_heads);
    if (d->pos_conv_groups <= 0 || d->hidden_size % d->pos_conv_groups || (d->hidden_size / d->pos_conv_groups) % 16)
        return si_fail(ctx, SI_EINVAL, "positional conv: hidden/groups must be a multiple of 16");
    for (int i = 0; i < d->num_conv; ++i)
        if (d->conv_dim[i] % 16 || d->conv_kernel[i] <= 0 || d->conv_stride[i] <= 0)
            return si_fail(ctx, SI_EINVAL, "conv layer %d: dim %d must be a multiple of 16", i, d->conv_dim[i]);
    if (d->hidden_size % 16 || d->intermediate_size % 16) return si_fail(ctx, SI_EINVAL, "hidden / intermediate sizes must be multiples of 16");
    if (d->codebook_dim <= 0 || d->codebook_dim > 128 || d->num_clusters <= 0) return si_fail(ctx, SI_EINVAL, "codebook %dx%d unsupported", d->num_clusters, d->codebook_dim);
    int c = d->up_initial_channel;
    for (int i = 0; i < d->num_ups; ++i) {
        // output length rate * L needs padding (k - u) / 2 exact; k need not be a multiple of u (I_da's 11 / 5)
        if (d->up_kernels[i] < d->up_rates[i] || (d->up_kernels[i] - d->up_rates[i]) % 2)
            return si_fail(ctx, SI_EINVAL, "upsample %d: kernel %d must be >= rate %d with an even difference", i, d->up_kernels[i], d->up_rates[i]);
        if (c % 32) return si_fail(ctx, SI_EINVAL, "upsample %d: %d input channels must be a multiple of 32", i, c);
        c /= 2;
    }
    if (c % 4) return si_fail(ctx, SI_EINVAL, "final generator width %d must be a multiple of 4", c);
    for (int j = 0; j < d->num_rb; ++j)
        if (d->rb_kernels[j] % 2 == 0) return si_fail(ctx, SI_EINVAL, "resblock kernel %d must be odd", d->rb_kernels[j]);
    if (d->resblock_type < 0 || d->resblock_type > 2) return si_fail(ctx, SI_EINVAL, "resblock_type %d: 1 (ResBlock1) or 2 (ResBlock2)", d->resblock_type);
    for (int m : {d->encoder_math, d->vocoder_math})
        if (m < SI_MATH_F32 || m > SI_MATH_F16) return si_fail(ctx, SI_EINVAL, "unknown math mode %d", m);
    return SI_OK;
}

// ---- host packing helpers ------------------------------------------------------------------------
unsigned short h_f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// fp32 -> fp16, round to nearest even, saturating at +-65504 (NaN stays NaN)
unsigned short h_f2h(float f) {
    if (f != f) return 0x7e00;
    const float c = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
    const _Float16 h = (_Float16)c;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}
float h_bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

struct Packer {
    si_ctx* ctx;
    std::map<std::string, HostTensor> t;
    std::vector<char> out;
    int rc = SI_OK;

    const HostTensor* get(const std::string& name, std::initializer_list<long> shape) {
        auto it = t.find(name);
        if (it == t.end()) { if (!rc) rc = si_fail(ctx, SI_EWEIGHTS, "checkpoint is missing tensor '%s'", name.c_str()); return nullptr; }
        std::vector<long> want(shape);
        if (it->second.shape != want) {
            std::ostringstream a, b;
            for (long v : it->second.shape) a << v << ' ';
            for (long v : want) b << v << ' ';
            if (!rc) rc = si_fail(ctx, SI_EWEIGHTS, "tensor '%s' has shape [%s] but the model needs [%s]", name.c_str(), a.str().c_str(), b.str().c_str());
            return nullptr;
        }
        return &it->second;
    }
    bool has(const std::string& n) const { return t.count(n) != 0; }
    float* fptr(size_t off) { return reinterpret_cast<float*>(out.data() + off); }
    void copy_floats(size_t off, const std::string& name, long n) {
        const HostTensor* h = get(name, {n});
        if (h) memcpy(out.data() + off, h->data, (size_t)n * 4);
    }
    // Folded weight of a conv-like module (shape `shape`); norm_dim = the dim weight_norm keeps.
    bool folded(const std::string& prefix, std::initializer_list<long> shape, int norm_dim, std::vector<float>& w) {
        std::vector<long> sh(shape);
        long n = 1; for (long v : sh) n *= v;
        if (has(prefix + ".weight")) {
            const HostTensor* h = get(prefix + ".weight", shape);
            if (!h) return false;
            w.assign(h->data, h->data + n);
            return true;
        }
        std::string gn = prefix + ".weight_g", vn = prefix + ".weight_v";
        if (!has(gn)) { gn = prefix + ".parametrizations.weight.original0"; vn = prefix + ".parametrizations.weight.original1"; }
        if (!has(gn)) { if (!rc) rc = si_fail(ctx, SI_EWEIGHTS, "checkpoint has neither '%s.weight' nor a weight-norm pair", prefix.c_str()); return false; }
        const HostTensor* v = get(vn, shape);
        if (!v) return false;
        auto git = t.find(gn);
        const long keep = sh[norm_dim];
        if (git->second.numel() != keep) { if (!rc) rc = si_fail(ctx, SI_EWEIGHTS, "'%s' must hold %ld magnitudes", gn.c_str(), keep); return false; }
        long inner = 1; for (size_t i = norm_dim + 1; i < sh.size(); ++i) inner *= sh[i];
        std::vector<double> ss(keep, 0.0);
        for (long i = 0; i < n; ++i) { const long kidx = (i / inner) % keep; ss[kidx] += (double)v->data[i] * v->data[i]; }
        w.resize(n);
        for (long i = 0; i < n; ++i) {
            const long kidx = (i / inner) % keep;
            // same operation order as torch._weight_norm: v * (g / norm), all in fp32
            const float nrm = (float)std::sqrt(ss[kidx]);
            w[i] = v->data[i] * (git->second.data[kidx] / nrm);
        }
        return true;
    }
    // store element (g, tap, n, ci) of a tap-GEMM weight
    void put(const GemmW& G, int g, int tap, int n, int ci, float v) {
        const size_t idx = (((size_t)g * G.ntaps + tap) * G.Npad + n) * G.Cin + ci;
        if (G.math == SI_MATH_F32) { fptr(G.w)[idx] = v; return; }
        if (G.math == SI_MATH_F16) { reinterpret_cast<unsigned short*>(out.data() + G.w)[idx] = h_f2h(v); return; }
        const unsigned short hi = h_f2bf(v);
        reinterpret_cast<unsigned short*>(out.data() + G.w)[idx] = hi;
        if (G.math == SI_MATH_BF16X3) reinterpret_cast<unsigned short*>(out.data() + G.w_lo)[idx] = h_f2bf(v - h_bf2f(hi));
    }
    // Conv1d weight (Cout, Cin/groups, k) -> W[g][tap][n][ci]; cin_valid < G.Cin leaves zero padding
    void conv(const GemmW& G, const std::vector<float>& w, int cout_total, int cin_g, int k) {
        const int npg = cout_total / G.groups;
        for (int co = 0; co < cout_total; ++co)
            for (int ci = 0; ci < cin_g; ++ci)
                for (int tp = 0; tp < k; ++tp)
                    put(G, co / npg, tp, co % npg, ci, w[((size_t)co * cin_g + ci) * k + tp]);
    }
    void bias(const GemmW& G, const std::string& name, long n) { if (G.has_bias) copy_floats(G.bias, name, n); }
};

int parse_index(si_ctx* ctx, const char* index, const void* blob, size_t nbytes, std::map<std::string, HostTensor>& out) {
    std::istringstream in(index ? index : "");
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        if (line.empty()) continue;
        std::istringstream ls(line);
        std::string name; long long off; int nd;
        if (!(ls >> name >> off >> nd) || nd < 0 || nd > 8) return si_fail(ctx, SI_EWEIGHTS, "index line %d is malformed: '%s'", lineno, line.c_str());
        HostTensor t;
        for (int i = 0; i < nd; ++i) { long v; if (!(ls >> v) || v < 0) return si_fail(ctx, SI_EWEIGHTS, "index line %d: bad dims", lineno); t.shape.push_back(v); }
        if (off < 0 || off % 4 || (size_t)off + (size_t)t.numel() * 4 > nbytes)
            return si_fail(ctx, SI_EWEIGHTS, "tensor '%s' (offset %lld, %ld floats) does not fit the %zu-byte blob", name.c_str(), off, t.numel(), nbytes);
        t.data = reinterpret_cast<const float*>(static_cast<const char*>(blob) + off);
        out[name] = t;
    }
    if (out.empty()) return si_fail(ctx, SI_EWEIGHTS, "checkpoint index is empty");
    return SI_OK;
}

int pack_weights(si_ctx* ctx, Packer& P) {
    const si_model_desc& d = ctx->d;
    const Layout& L = ctx->lay;
    P.out.assign(L.total, 0);
    const std::string B = "base_model.";
    const int H = d.hidden_size;
    std::vector<float> w;
    // ---- feature extractor
    {
        const std::string p0 = B + "feature_extractor.conv_layers.0.";
        const HostTensor* h = P.get(p0 + "conv.weight", {d.conv_dim[0], 1, d.conv_kernel[0]});
        if (h) memcpy(P.out.data() + L.conv0_w, h->data, (size_t)h->numel() * 4);
        if (d.conv_bias) P.copy_floats(L.conv0_bias, p0 + "conv.bias", d.conv_dim[0]);
        P.copy_floats(L.conv0_g, p0 + "layer_norm.weight", d.conv_dim[0]);
        P.copy_floats(L.conv0_b, p0 + "layer_norm.bias", d.conv_dim[0]);
    }
    for (int i = 1; i < d.num_conv; ++i) {
        const std::string p = B + "feature_extractor.conv_layers." + std::to_string(i) + ".";
        const ConvW& c = L.convs[i - 1];
        const HostTensor* h = P.get(p + "conv.weight", {d.conv_dim[i], d.conv_dim[i - 1], d.conv_kernel[i]});
        if (h) { w.assign(h->data, h->data + h->numel()); P.conv(c.g, w, d.conv_dim[i], d.conv_dim[i - 1], d.conv_kernel[i]); }
        P.bias(c.g, p + "conv.bias", d.conv_dim[i]);
        if (d.feat_norm_layer) { P.copy_floats(c.ln_g, p + "layer_norm.weight", d.conv_dim[i]); P.copy_floats(c.ln_b, p + "layer_norm.bias", d.conv_dim[i]); }
    }
    const int CF = d.conv_dim[d.num_conv - 1];
    if (d.feat_proj_layer_norm) {
        P.copy_floats(L.fp_ln_g, B + "feature_projection.layer_norm.weight", CF);
        P.copy_floats(L.fp_ln_b, B + "feature_projection.layer_norm.bias", CF);
    }
    auto linear = [&](const GemmW& G, const std::string& name, int nout, int nin, int row0 = 0) {
        const HostTensor* h = P.get(name + ".weight", {nout, nin});
        if (h) for (int o = 0; o < nout; ++o) for (int i = 0; i < nin; ++i) P.put(G, 0, 0, row0 + o, i, h->data[(size_t)o * nin + i]);
        const HostTensor* b = P.get(name + ".bias", {nout});
        if (b) memcpy(P.fptr(G.bias) + row0, b->data, (size_t)nout * 4);
    };
    linear(L.proj, B + "feature_projection.projection", H, CF);
    {
        const int cg = H / d.pos_conv_groups;
        if (P.folded(B + "encoder.pos_conv_embed.conv", {H, cg, d.pos_conv_kernel}, 2, w)) P.conv(L.pos, w, H, cg, d.pos_conv_kernel);
        P.bias(L.pos, B + "encoder.pos_conv_embed.conv.bias", H);
    }
    P.copy_floats(L.enc_ln_g, B + "encoder.layer_norm.weight", H);
    P.copy_floats(L.enc_ln_b, B + "encoder.layer_norm.bias", H);
    for (int l = 0; l < d.num_layers; ++l) {
        const std::string p = B + "encoder.layers." + std::to_string(l) + ".";
        const LayerW& W = L.layers[l];
        linear(W.qkv, p + "attention.q_proj", H, H, 0);
        linear(W.qkv, p + "attention.k_proj", H, H, H);
        linear(W.qkv, p + "attention.v_proj", H, H, 2 * H);
        linear(W.out, p + "attention.out_proj", H, H);
        P.copy_floats(W.ln1_g, p + "layer_norm.weight", H); P.copy_floats(W.ln1_b, p + "layer_norm.bias", H);
        linear(W.ffn1, p + "feed_forward.intermediate_dense", d.intermediate_size, H);
        linear(W.ffn2, p + "feed_forward.output_dense", H, d.intermediate_size);
        P.copy_floats(W.ln2_g, p + "final_layer_norm.weight", H); P.copy_floats(W.ln2_b, p + "final_layer_norm.bias", H);
    }
    P.copy_floats(L.head_ln_g, "final_layers.0.weight", H);
    P.copy_floats(L.head_ln_b, "final_layers.0.bias", H);
    linear(L.head, "final_layers.1", d.codebook_dim, H);
    // ---- codebook tables (I_ea/loss_fn.py:10-14): centre = mean over K; centred; 1/max(||centred||, 1e-8)
    {
        const int K = d.num_clusters, D = d.codebook_dim;
        const HostTensor* c = P.get("codebook", {K, D});
        if (c) {
            std::vector<float> center(D, 0.f);
            for (int j = 0; j < D; ++j) { float s = 0.f; for (int k = 0; k < K; ++k) s += c->data[(size_t)k * D + j]; center[j] = s / K; }
            float* cc = P.fptr(L.cb_centered); float* raw = P.fptr(L.cb_raw); float* rn = P.fptr(L.cb_rnorm);
            for (int k = 0; k < K; ++k) {
                double ss = 0;
                for (int j = 0; j < D; ++j) {
                    const float v = c->data[(size_t)k * D + j] - center[j];
                    cc[(size_t)k * D + j] = v;
                    raw[(size_t)k * D + j] = v + center[j];        // predict.py:184: all_embeds_t_c + center_
                    ss += (double)v * v;
                }
                rn[k] = 1.0f / fmaxf((float)std::sqrt(ss), 1e-8f);
            }
        }
    }
    // ---- generator
    const std::string G = "generator.";
    const int C0 = d.up_initial_channel;
    if (P.folded(G + "conv_pre", {C0, d.num_mels, 7}, 0, w)) P.conv(L.pre, w, C0, d.num_mels, 7);
    P.bias(L.pre, G + "conv_pre.bias", C0);
    int c = C0;                                            // REAL channel counts here; the packed matrices have the padded strides
    for (int i = 0; i < d.num_ups; ++i) {
        const int u = d.up_rates[i], k = d.up_kernels[i], cout = c / 2, coutp = stage_channels(ctx, i);
        const GemmW& U = L.ups[i];
        // ConvTranspose1d weight (Cin, Cout, k), weight-norm over dim 0 (= Cin).  Phase p = j mod u, tap q = j / u.
        if (P.folded(G + "ups." + std::to_string(i), {c, cout, k}, 0, w))
            for (int ci = 0; ci < c; ++ci)
                for (int co = 0; co < cout; ++co)
                    for (int j = 0; j < k; ++j) P.put(U, 0, j / u, (j % u) * coutp + co, ci, w[((size_t)ci * cout + co) * k + j]);
        const HostTensor* b = P.get(G + "ups." + std::to_string(i) + ".bias", {cout});
        if (b) for (int ph = 0; ph < u; ++ph) memcpy(P.fptr(U.bias) + (size_t)ph * coutp, b->data, (size_t)cout * 4);
        c = cout;
        for (int j = 0; j < d.num_rb; ++j) {
            const ResW& R = L.rbs[(size_t)i * d.num_rb + j];
            const std::string rp = G + "resblocks." + std::to_string(i * d.num_rb + j) + ".";
            for (int n = 0; n < d.num_dil; ++n) {
                if (d.resblock_type == 2) {                       // ResBlock2: `convs.<n>` (I_ea/hifi_gan/models.py:56-61)
                    const std::string a = rp + "convs." + std::to_string(n);
                    if (P.folded(a, {c, c, d.rb_kernels[j]}, 0, w)) P.conv(R.c1[n], w, c, c, d.rb_kernels[j]);
                    P.bias(R.c1[n], a + ".bias", c);
                    continue;
                }
                const std::string a = rp + "convs1." + std::to_string(n), bb = rp + "convs2." + std::to_string(n);
                if (P.folded(a, {c, c, d.rb_kernels[j]}, 0, w)) P.conv(R.c1[n], w, c, c, d.rb_kernels[j]);
                P.bias(R.c1[n], a + ".bias", c);
                if (P.folded(bb, {c, c, d.rb_kernels[j]}, 0, w)) P.conv(R.c2[n], w, c, c, d.rb_kernels[j]);
                P.bias(R.c2[n], bb + ".bias", c);
            }
        }
    }
    if (P.folded(G + "conv_post", {1, c, 7}, 0, w))
        for (int ci = 0; ci < c; ++ci) for (int k = 0; k < 7; ++k) P.fptr(L.post_w)[(size_t)k * L.post_C + ci] = w[(size_t)ci * 7 + k];
    P.copy_floats(L.post_b, G + "conv_post.bias", 1);
    return P.rc;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t cap; size_t cur = 0; bool ok = true;
    float* floats(size_t n) { return reinterpret_cast<float*>(bytes(n * 4)); }
    char* bytes(size_t n) {
        size_t o = cur; cur = align_up(cur + n, 256);
        if (cur > cap) { ok = false; return base; }
        return base + o;
    }
};

struct EncDims { int T; std::vector<int> L; };
EncDims enc_dims(const si_model_desc& d, int N) {
    EncDims e; e.L.push_back(N);
    int n = N;
    for (int i = 0; i < d.num_conv; ++i) { n = n >= d.conv_kernel[i] ? (n - d.conv_kernel[i]) / d.conv_stride[i] + 1 : 0; e.L.push_back(n); }
    e.T = n;
    return e;
}

size_t encoder_ws_bytes(const si_model_desc& d, int B, int N) {
    EncDims e = enc_dims(d, N);
    size_t cmax = 0;
    for (int i = 0; i < d.num_conv; ++i) cmax = std::max(cmax, (size_t)e.L[i + 1] * d.conv_dim[i]);
    const size_t BT = (size_t)B * std::max(e.T, 1);
    size_t f = 2 * (size_t)B * cmax + BT * d.conv_dim[d.num_conv - 1] + BT * d.hidden_size * 3 + BT * 3 * d.hidden_size + BT * d.intermediate_size +
               (size_t)B * d.conv_dim[0] * 2 + BT * 2;                    // (+ the LayerNorm (mean, rstd) rows of the fused post-LN layers)
    // bf16 operand-ready copies (encoder in bf16 mode): LN(features), hidden, attention output, FFN intermediate
    const size_t h16 = (BT * d.conv_dim[d.num_conv - 1] + 2 * BT * d.hidden_size + BT * (d.intermediate_size + 128)) * 2;   // (+ the row padding of the FFN intermediate)
    return f * 4 + h16 + (size_t)B * 16 + (size_t)B * 4 + si_conv0_partials_bytes(B, N) + 41 * 256 +
           align_up((size_t)(SI_MAX_CONV + 3) * (B + 1) * 4, 256);    // ragged batches: per-layer length table + row offsets
}

// Clips per vocoder pass.  Measured on MI355X (B = 32, fp32): 4 -> 109 ms/step, 8 -> 93, 16 -> 89, 32 -> 88: small
// chunks leave the early stages (L = 2752 rows per clip) with fewer workgroups than CUs, and the fp32 MFMA path
// is compute-bound, so cache residency of a small chunk buys nothing.
int vocoder_chunk(const si_model_desc& d, int B) { int c = d.vocoder_chunk > 0 ? d.vocoder_chunk : 32; return std::min(c, std::max(B, 1)); }

long voc_tout(int Tm, int stretch) { return stretch ? (long)std::floor((double)Tm * (441.0 / 256.0)) : Tm; }

size_t vocoder_ws_bytes(const si_ctx* ctx, int B, int Tm, int stretch) {
    const si_model_desc& d = ctx->d;
    const int Bc = vocoder_chunk(d, B);
    const long Tout = voc_tout(Tm, stretch);
    size_t lc_max = (size_t)Tout * d.up_initial_channel;
    long L = Tout; int c = d.up_initial_channel;
    for (int i = 0; i < d.num_ups; ++i) { L *= d.up_rates[i]; c = stage_channels(ctx, i); lc_max = std::max(lc_max, (size_t)L * c); }
    // 6 fp32 activation buffers + 6 half-size buffers for the operand-ready 16-bit copies (bf16 / fp16 modes)
    const size_t f = (size_t)Bc * Tout * ctx->lay.mel_ld + 9 * (size_t)Bc * lc_max;
    return f * 4 + 32 * 256 + align_up((size_t)(2 * SI_MAX_UPS + 3) * B * 4, 256);   // ragged batches: per-stage length table
}

TapGemmParams gemm_params(const si_ctx* ctx, const GemmW& G) {
    TapGemmParams p{};
    p.w = ctx->wdev + G.w;
    p.w_lo = G.math == SI_MATH_BF16X3 ? ctx->wdev + G.w_lo : nullptr;
    p.bias = G.has_bias ? reinterpret_cast<const float*>(ctx->wdev + G.bias) : nullptr;
    p.Cin = G.Cin; p.N = G.N; p.Npad = G.Npad; p.ntaps = G.ntaps; p.groups = G.groups;
    p.stride = 1; p.dil = 1; p.pad = 0; p.pro_slope = 1.f; p.act = SI_ACT_NONE; p.alpha = 1.f; p.accumulate = 0;
    p.out16_slope = 1.f;
    p.nseg = 1;
    return p;
}

// y(rows x N) = x(rows x K) W^T + b [+act] [+res]
// x16 / y16: operand-ready bf16 input (instead of x) / additional-or-only bf16 output, see TapGemmParams
// ld_in / ld_out (elements; 0 = dense): row strides of the 16-bit input / of the output when they are padded (the FFN intermediate)
// res_ln (stats, gamma, beta): the residual is LayerNorm(res), recomputed in the GEMM's epilogue (TapGemmParams::res_stats)
struct ResLn { const float* stats = nullptr; const float* gamma = nullptr; const float* beta = nullptr; };
int linear(si_ctx* ctx, const GemmW& G, const float* x, float* y, long rows, int act, const float* res, hipStream_t st,
           const unsigned short* x16 = nullptr, unsigned short* y16 = nullptr, int ld_in = 0, int ld_out = 0, ResLn res_ln = ResLn()) {
    TapGemmParams p = gemm_params(ctx, G);
    p.x = x16 ? nullptr : x; p.x16 = x16; p.out = y; p.out16 = y16; p.res = res; p.act = act;
    p.res_stats = res_ln.stats; p.res_gamma = res_ln.gamma; p.res_beta = res_ln.beta;
    p.nseg = 1; p.Lin = (int)rows; p.M = (int)rows; p.ldx = ld_in ? ld_in : G.Cin; p.x_seg_stride = 0;
    p.ldo = ld_out ? ld_out : G.N; p.o_seg_stride = 0; p.ooff = 0; p.olimit = rows * p.ldo;
    p.lingemm = ctx->opt_enc_lingemm;
    return si_launch_tapgemm(ctx, G.math, p, st);
}

const float* wf(const si_ctx* ctx, size_t off) { return reinterpret_cast<const float*>(ctx->wdev + off); }

}  // namespace

static int hubert_run(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, const int32_t* valid_len,
                      int normalize, float norm_eps, const double* pre_add, int output_layer, int B, int N, float* out_feats, float* out_hidden,
                      void* workspace, size_t workspace_bytes, si_stream_t stream, const int32_t* host_len = nullptr);

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int si_version(void) { return SI_ABI_VERSION; }

const char* si_last_error(const si_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

int si_create(si_ctx** out, int device_id, const si_model_desc* desc) {
    if (!out) return si_fail(nullptr, SI_EINVAL, "si_create: out is NULL");
    *out = nullptr;
    int rc = check_desc(nullptr, desc);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
        return si_fail(nullptr, SI_EHIP, "si_create: device %d not available (%d HIP devices visible)", device_id, ndev);
    si_ctx* ctx = new si_ctx();
    ctx->device = device_id;
    ctx->d = *desc;
    auto env_flag = [](const char* name) { const char* v = getenv(name); return v ? atoi(v) != 0 : true; };
    ctx->opt_voc_opready = env_flag("SI_VOC_OPREADY");
    ctx->opt_voc_res16 = env_flag("SI_VOC_RES16");
    ctx->opt_voc_fuse = getenv("SI_VOC_FUSE") ? atoi(getenv("SI_VOC_FUSE")) : 1;
    ctx->opt_voc_chain = getenv("SI_VOC_CHAIN") ? atoi(getenv("SI_VOC_CHAIN")) : 1;
    ctx->opt_enc_opready = env_flag("SI_ENC_OPREADY");
    ctx->opt_att_bf16 = env_flag("SI_ATT_BF16");
    ctx->opt_enc_lingemm = env_flag("SI_ENC_LINGEMM");
    ctx->opt_enc_posconv = env_flag("SI_ENC_POSCONV");
    ctx->opt_gemm256 = getenv("SI_ENC_GEMM256") ? atoi(getenv("SI_ENC_GEMM256")) : 1;
    ctx->opt_gemmcu = getenv("SI_ENC_GEMMCU") ? atoi(getenv("SI_ENC_GEMMCU")) : 1;
    ctx->opt_ln_fuse = getenv("SI_ENC_LNFUSE") ? atoi(getenv("SI_ENC_LNFUSE")) : 1;
    ctx->opt_voc_upsgemm = getenv("SI_VOC_UPSGEMM") ? atoi(getenv("SI_VOC_UPSGEMM")) : 1;
    ctx->opt_ffn_pad = getenv("SI_ENC_FFNPAD") ? std::min(128, std::max(0, atoi(getenv("SI_ENC_FFNPAD")) / 8 * 8)) : 64;
    plan_layout(ctx);
    *out = ctx;
    return SI_OK;
}

void si_destroy(si_ctx* ctx) {
    if (!ctx) return;
    if (ctx->wdev) { (void)hipSetDevice(ctx->device); (void)hipFree(ctx->wdev); }
    if (ctx->fe_dev) { (void)hipSetDevice(ctx->device); (void)hipFree(ctx->fe_dev); }
    if (ctx->km_cnorm) { (void)hipSetDevice(ctx->device); (void)hipFree(ctx->km_cnorm); }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < si_ctx::VL_SLOTS; ++i) {
        if (ctx->vl_ev[i]) { (void)hipEventSynchronize(ctx->vl_ev[i]); (void)hipEventDestroy(ctx->vl_ev[i]); }
        if (ctx->vl_pin[i]) (void)hipHostFree(ctx->vl_pin[i]);
    }
    delete ctx;
}

int si_alloc_weights(si_ctx* ctx) {
    if (!ctx) return SI_EINVAL;
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->wdev) SI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->wdev), ctx->lay.total));
    ctx->weights_ready = true;          // bytes arrive by broadcast before the first forward
    ctx->weights_verified = false;      // ... and are checked against this context's layout then (si_weights_check)
    return SI_OK;
}

int si_load_weights(si_ctx* ctx, const void* host_blob, size_t nbytes, const char* index) {
    if (!ctx || !host_blob || !index) return si_fail(ctx, SI_EINVAL, "si_load_weights: NULL argument");
    Packer P;
    P.ctx = ctx;
    int rc = parse_index(ctx, index, host_blob, nbytes, P.t);
    if (rc) return rc;
    rc = pack_weights(ctx, P);
    if (rc) return rc;
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->wdev) SI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->wdev), ctx->lay.total));
    const BlobHeader hd = blob_header(ctx);
    memcpy(P.out.data(), &hd, sizeof(hd));
    SI_HIP_CHECK(hipMemcpy(ctx->wdev, P.out.data(), ctx->lay.total, hipMemcpyHostToDevice));
    ctx->weights_ready = true;
    ctx->weights_verified = true;
    return SI_OK;
}

int si_weights_check(si_ctx* ctx) {
    if (!ctx) return SI_EINVAL;
    if (!ctx->wdev) return si_fail(ctx, SI_ESTATE, "weights are not allocated: call si_load_weights or si_alloc_weights first");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    BlobHeader got{};
    SI_HIP_CHECK(hipMemcpy(&got, ctx->wdev, sizeof(got), hipMemcpyDeviceToHost));
    const BlobHeader want = blob_header(ctx);
    if (got.magic != want.magic || got.version != want.version || got.fingerprint != want.fingerprint || got.total != want.total)
        return si_fail(ctx, SI_EWEIGHTS, "the weight blob in this context was packed for a different layout (fingerprint %016llx / %llu bytes, this "
                       "context plans %016llx / %llu): the source rank's model desc or SI_VOC_* environment differs, or the broadcast has not arrived",
                       (unsigned long long)got.fingerprint, (unsigned long long)got.total, (unsigned long long)want.fingerprint, (unsigned long long)want.total);
    ctx->weights_verified = true;
    return SI_OK;
}

int si_weights_device_ptr(si_ctx* ctx, void** ptr, size_t* nbytes) {
    if (!ctx || !ptr || !nbytes) return si_fail(ctx, SI_EINVAL, "si_weights_device_ptr: NULL argument");
    if (!ctx->wdev) return si_fail(ctx, SI_ESTATE, "weights are not allocated: call si_load_weights or si_alloc_weights first");
    *ptr = ctx->wdev;
    *nbytes = ctx->lay.total;
    return SI_OK;
}

int si_num_frames(const si_ctx* ctx, int N) { return ctx ? enc_dims(ctx->d, N).T : SI_EINVAL; }

int si_vocoder_samples(const si_ctx* ctx, int Tm, int stretch) {
    if (!ctx) return SI_EINVAL;
    long L = voc_tout(Tm, stretch);
    for (int i = 0; i < ctx->d.num_ups; ++i) L *= ctx->d.up_rates[i];
    return (int)L;
}

int si_workspace_bytes(si_ctx* ctx, int B, int N, int Tm, size_t* out) {
    if (!ctx || !out || B <= 0) return si_fail(ctx, SI_EINVAL, "si_workspace_bytes: bad argument");
    size_t a = N > 0 ? encoder_ws_bytes(ctx->d, B, N) : 0;
    size_t b = Tm > 0 ? vocoder_ws_bytes(ctx, B, Tm, 1) : 0;
    *out = std::max(a, b);
    return SI_OK;
}

int si_hubert_forward(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, int normalize, int B, int N,
                      float* out_feats, void* workspace, size_t workspace_bytes, si_stream_t stream) {
    return si_hubert_forward_padded(ctx, wav, mask_start, mask_len, nullptr, normalize, B, N, out_feats, workspace, workspace_bytes, stream);
}

int si_hubert_forward_padded(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, const int32_t* valid_len,
                             int normalize, int B, int N, float* out_feats, void* workspace, size_t workspace_bytes, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!out_feats) return si_fail(ctx, SI_EINVAL, "si_hubert_forward: NULL / empty argument");
    return hubert_run(ctx, wav, mask_start, mask_len, valid_len, normalize, 1e-7f, nullptr, 0, B, N, out_feats, nullptr, workspace,
                      workspace_bytes, stream);
}

int si_hubert_forward_varlen(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, const int32_t* sample_len,
                             int normalize, int B, int N, float* out_feats, void* workspace, size_t workspace_bytes, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!out_feats || !sample_len) return si_fail(ctx, SI_EINVAL, "si_hubert_forward_varlen: NULL argument");
    return hubert_run(ctx, wav, mask_start, mask_len, nullptr, normalize, 1e-7f, nullptr, 0, B, N, out_feats, nullptr, workspace, workspace_bytes,
                      stream, sample_len);
}

int si_hubert_extract_features(si_ctx* ctx, const si_extract_desc* x, const float* wav, const int32_t* mask_start, const int32_t* mask_len,
                               const double* pre_mask_add, int B, int N, float* out_hidden, void* workspace, size_t workspace_bytes,
                               si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!x || x->struct_size != (int32_t)sizeof(si_extract_desc)) return si_fail(ctx, SI_EINVAL, "si_hubert_extract_features: si_extract_desc size mismatch");
    if (!out_hidden) return si_fail(ctx, SI_EINVAL, "si_hubert_extract_features: NULL output");
    if (x->output_layer < 1 || x->output_layer > ctx->d.num_layers)
        return si_fail(ctx, SI_EINVAL, "output_layer %d outside 1..%d", x->output_layer, ctx->d.num_layers);
    if (x->normalize < 0 || x->normalize > 2) return si_fail(ctx, SI_EINVAL, "normalize mode %d unknown (0 none, 1 processor, 2 layer_norm)", x->normalize);
    return hubert_run(ctx, wav, mask_start, mask_len, nullptr, x->normalize != 0, x->normalize == 2 ? 1e-5f : 1e-7f, pre_mask_add,
                      x->output_layer, B, N, nullptr, out_hidden, workspace, workspace_bytes, stream);
}

int si_code_splice(si_ctx* ctx, const int64_t* code_clean, const int64_t* code_masked, const int32_t* first, const int32_t* last, int B, int T,
                   int64_t* out, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!code_clean || !code_masked || !first || !last || !out || B <= 0 || T <= 0) return si_fail(ctx, SI_EINVAL, "si_code_splice: NULL / empty argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_code_splice(ctx, code_clean, code_masked, first, last, B, T, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// The encoder.  output_layer = 0: all layers [+ the stable flavour's final LayerNorm] + final_layers -> out_feats (B, T, codebook_dim).
// output_layer = L >= 1: stop after L transformer layers and copy the hidden state (the residual stream in the pre-LN flavour, the
// layer's output LayerNorm in the post-LN one) to out_hidden (B, T, H) -- fairseq's extract_features(output_layer = L).
// host_len (B, HOST) or null: ragged batch -- clip b holds host_len[b] samples of its row of N; each clip's result equals that clip
// run alone (conv0 statistics over its own rows, convolutions stop at its own lengths, the transformer runs on PACKED rows).
static int hubert_run(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, const int32_t* valid_len,
                      int normalize, float norm_eps, const double* pre_add, int output_layer, int B, int N, float* out_feats, float* out_hidden,
                      void* workspace, size_t workspace_bytes, si_stream_t stream, const int32_t* host_len) {
    if (!ctx->weights_ready) return si_fail(ctx, SI_ESTATE, "si_hubert_forward before weights were loaded");
    if (!ctx->weights_verified) if (int rcw = si_weights_check(ctx)) return rcw;      // a received blob: once, before its first use
    if (!wav || !workspace || B <= 0) return si_fail(ctx, SI_EINVAL, "si_hubert_forward: NULL / empty argument");
    const si_model_desc& d = ctx->d;
    const Layout& L = ctx->lay;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const EncDims e = enc_dims(d, N);
    if (e.T < 1) return si_fail(ctx, SI_EINVAL, "clip of %d samples is shorter than the conv stack's receptive field", N);
    if (workspace_bytes < encoder_ws_bytes(d, B, N))
        return si_fail(ctx, SI_ENOMEM, "workspace of %zu bytes < %zu needed for B=%d N=%d", workspace_bytes, encoder_ws_bytes(d, B, N), B, N);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    const int T = e.T, H = d.hidden_size, I = d.intermediate_size, CF = d.conv_dim[d.num_conv - 1];
    size_t cmax = 0;
    for (int i = 0; i < d.num_conv; ++i) cmax = std::max(cmax, (size_t)e.L[i + 1] * d.conv_dim[i]);

    Carver W{static_cast<char*>(workspace), workspace_bytes};
    // ---- ragged batch: the table [samples | L_1 .. L_nconv | row offsets (B + 1)] on the host and on the device
    const bool vl = host_len != nullptr;
    const int32_t* h_tab = nullptr;      // host
    int32_t* d_tab = nullptr;            // device
    long rows_packed = 0;
    double sum_t2 = 0.0;
    if (vl) {
        if (valid_len || output_layer) return si_fail(ctx, SI_EINVAL, "ragged batches: not combined with padded batches or output_layer");
        if (d.codebook_dim % 4) return si_fail(ctx, SI_EINVAL, "ragged batches need codebook_dim %% 4 == 0");
        std::vector<int32_t>& tab = ctx->vl_host;
        tab.assign((size_t)(d.num_conv + 1) * B + B + 1, 0);
        for (int b = 0; b < B; ++b) {
            if (host_len[b] < 1 || host_len[b] > N) return si_fail(ctx, SI_EINVAL, "ragged batch: clip %d holds %d samples, outside 1..%d", b, host_len[b], N);
            const EncDims eb = enc_dims(d, host_len[b]);
            if (eb.T < 1) return si_fail(ctx, SI_EINVAL, "ragged batch: clip %d (%d samples) is shorter than the conv stack's receptive field", b, host_len[b]);
            for (int i = 0; i <= d.num_conv; ++i) tab[(size_t)i * B + b] = eb.L[i];
            tab[(size_t)(d.num_conv + 1) * B + b] = (int32_t)rows_packed;
            rows_packed += eb.T;
            sum_t2 += (double)eb.T * eb.T;
        }
        tab[(size_t)(d.num_conv + 1) * B + B] = (int32_t)rows_packed;
        d_tab = reinterpret_cast<int32_t*>(W.bytes(tab.size() * 4));
        if (!W.ok) return si_fail(ctx, SI_ENOMEM, "internal: encoder workspace carve exceeded its own estimate");
        if (int rc = vl_upload(ctx, d_tab, st)) return rc;
        h_tab = tab.data();
        valid_len = d_tab;                                   // the statistics and loaders of conv0 stop at each clip's own samples
    }
    auto dL = [&](int i) { return d_tab + (size_t)i * B; };            // device / host rows of conv layer i's output (0: samples)
    auto hL = [&](int i) { return h_tab + (size_t)i * B; };
    const int32_t* d_rowoff = vl ? d_tab + (size_t)(d.num_conv + 1) * B : nullptr;
    const long BT = vl ? rows_packed : (long)B * T;          // transformer rows
    double* stats = reinterpret_cast<double*>(W.bytes((size_t)B * 16));
    int32_t* vframes = (valid_len && !vl) ? reinterpret_cast<int32_t*>(W.bytes((size_t)B * 4)) : nullptr;
    double* partials = reinterpret_cast<double*>(W.bytes(si_conv0_partials_bytes(B, N)));
    float* affine = W.floats((size_t)B * d.conv_dim[0] * 2);
    float* cbuf[2] = {W.floats((size_t)B * cmax), W.floats((size_t)B * cmax)};
    const bool c16 = ctx->opt_enc_opready && d.encoder_math == SI_MATH_BF16 && !d.feat_norm_layer && d.num_conv >= 2;
    unsigned short* cb16[2] = {reinterpret_cast<unsigned short*>(cbuf[0]), reinterpret_cast<unsigned short*>(cbuf[1])};   // same storage, bf16 view
    float* lnf = W.floats(BT * CF);
    float* h = W.floats(BT * H);
    float* h2 = W.floats(BT * H);
    float* att = W.floats(BT * H);
    float* qkv = W.floats(BT * 3 * H);
    float* ffn = W.floats(BT * I);
    // Operand-ready bf16 activations for the encoder GEMMs (bf16 mode): LayerNorm, attention and the FFN's first GEMM
    // also / only write bf16(x) -- the rounding the consuming GEMM would apply while staging -- so the four GEMMs of a
    // layer load 2-byte operands (their dominant traffic: M = B*T rows re-read by every N-tile) and skip the
    // conversion.  Bit-identical to converting in the consumer.  SI_ENC_OPREADY=0 restores fp32 inputs.
    const bool e16 = ctx->opt_enc_opready && d.encoder_math == SI_MATH_BF16;
    unsigned short* lnf16 = e16 ? reinterpret_cast<unsigned short*>(W.bytes((size_t)BT * CF * 2)) : nullptr;
    unsigned short* h16 = e16 ? reinterpret_cast<unsigned short*>(W.bytes((size_t)BT * H * 2)) : nullptr;
    unsigned short* att16 = e16 ? reinterpret_cast<unsigned short*>(W.bytes((size_t)BT * H * 2)) : nullptr;
    // Post-LN layers in bf16 mode: a LayerNorm's output is read as a bf16 GEMM operand and as the fp32 residual of the second GEMM
    // behind it.  The fp32 copy (19.5 MB per launch at the bench shape, 4.7 of the LayerNorm's 14.4 us) is not written: the
    // LayerNorm stores (mean, rstd) per row and the residual-adding epilogue recomputes the element from the row it was
    // normalised from -- the SAME expression on the same floats (si_ln_apply), so the values do not change.  The pre-LN sums then
    // live in ONE buffer that out-proj / FFN2 update in place (each element is read and written by the same lane).
    // SI_ENC_LNFUSE=0, debug captures, fp32 mode or a width the bf16 GEMM kernels do not cover: every LayerNorm writes its rows.
    const bool ln_fuse = e16 && ctx->opt_ln_fuse && ctx->opt_enc_lingemm && !d.stable_layer_norm && ctx->dbg_capture.empty() && H % 128 == 0 && I % 64 == 0 &&
                         (double)(BT + 128) * (I + 128) * 2.0 < 2.0e9;
    float* ln_stats = ln_fuse ? W.floats(BT * 2) : nullptr;
    const int ffn_ld = e16 ? I + ctx->opt_ffn_pad : I;                 // row stride of the bf16 FFN intermediate (SI_ENC_FFNPAD)
    unsigned short* ffn16 = e16 ? reinterpret_cast<unsigned short*>(W.bytes((size_t)BT * ffn_ld * 2)) : nullptr;
    if (!W.ok) return si_fail(ctx, SI_ENOMEM, "internal: encoder workspace carve exceeded its own estimate");
    // bf16 mode with the bf16-MFMA attention: the QKV GEMM writes q | k | v as bf16 only (the rounding the attention
    // kernel's staging would apply), into the storage of the fp32 matrix
    const bool qkv_bf16 = e16 && ctx->opt_att_bf16;
    unsigned short* qkv16 = reinterpret_cast<unsigned short*>(qkv);

    int rc;
    // A0 + A1: normalise fused into conv0
    WaveNormParams wp{wav, mask_start, mask_len, B, N, e.L[1], d.conv_dim[0], d.conv_kernel[0], d.conv_stride[0], normalize, valid_len, norm_eps, pre_add,
                      vl ? dL(1) : nullptr};
    if ((rc = si_launch_wave_stats(ctx, wp, stats, st))) return rc;
    if (vframes && (rc = si_launch_frame_lengths(ctx, valid_len, B, d.num_conv, d.conv_kernel, d.conv_stride, e.T, vframes, st))) return rc;
    // layer-norm flavour (HuBERT-large) in bf16 mode: every conv is followed by LayerNorm + GELU over its 512 channels; the
    // LayerNorm writes ONLY the bf16 operand of the next conv (the rounding that conv would apply while staging), into the
    // other buffer (its bf16 rows would overlap unread fp32 rows of its own input), so that the convolutions run on the
    // dedicated bf16 GEMM kernels (lingemm / gemm256) like the group-norm flavour's; the last one writes the fp32 features.
    const bool l16 = ctx->opt_enc_opready && d.encoder_math == SI_MATH_BF16 && d.feat_norm_layer && d.num_conv >= 2;
    if (!d.feat_norm_layer) {
        // group-norm flavour in bf16 mode: the conv chain runs on operand-ready bf16 activations (conv0 and convs 1..n-2
        // write ONLY the bf16 operand of their single consumer; the last conv writes fp32 for the LayerNorm that follows)
        rc = si_launch_conv0_groupnorm(ctx, wp, stats, wf(ctx, L.conv0_w), wf(ctx, L.conv0_g), wf(ctx, L.conv0_b), partials, affine, cbuf[0], st,
                                       c16 ? cb16[0] : nullptr);
    } else {
        rc = si_launch_conv0_affine(ctx, wp, stats, wf(ctx, L.conv0_w), d.conv_bias ? wf(ctx, L.conv0_bias) : nullptr, affine, cbuf[0], st);
        if (!rc) rc = si_launch_layernorm(ctx, cbuf[0], nullptr, wf(ctx, L.conv0_g), wf(ctx, L.conv0_b), l16 ? nullptr : cbuf[0], (long)B * e.L[1], d.conv_dim[0],
                                          1e-5f, 1, st, l16 ? cb16[1] : nullptr);
    }
    if (rc) return rc;
    // A2: strided convs as tap-GEMMs
    int cur = 0;
    for (int i = 1; i < d.num_conv; ++i) {
        const ConvW& c = L.convs[i - 1];
        TapGemmParams p = gemm_params(ctx, c.g);
        p.x = cbuf[cur]; p.out = cbuf[cur ^ 1];
        if (c16) {
            p.x = nullptr; p.x16 = cb16[cur];
            if (i + 1 < d.num_conv) { p.out = nullptr; p.out16 = cb16[cur ^ 1]; p.out16_slope = 1.f; }
        }
        if (l16) { p.x = nullptr; p.x16 = cb16[1]; p.out = cbuf[0]; }          // bf16 in from buffer 1, fp32 out to buffer 0
        p.nseg = B; p.Lin = e.L[i]; p.M = e.L[i + 1]; p.ldx = d.conv_dim[i - 1]; p.x_seg_stride = (long)e.L[i] * d.conv_dim[i - 1];
        p.stride = d.conv_stride[i]; p.ldo = d.conv_dim[i]; p.o_seg_stride = (long)e.L[i + 1] * d.conv_dim[i];
        p.olimit = p.o_seg_stride;
        p.act = d.feat_norm_layer ? SI_ACT_NONE : SI_ACT_GELU;
        p.lingemm = ctx->opt_enc_lingemm;
        if (vl) {                                            // each clip's own input / output rows (modeling_hubert.py:664-677)
            p.seg_lin = dL(i); p.seg_m = dL(i + 1); p.seg_orows = dL(i + 1); p.olim_mul = p.ldo; p.seg_m_host = hL(i + 1);
            double rows = 0; for (int b = 0; b < B; ++b) rows += hL(i + 1)[b];
            p.algo_macs = rows * d.conv_dim[i] * (double)d.conv_dim[i - 1] * d.conv_kernel[i];
        }
        if ((rc = si_launch_tapgemm(ctx, c.g.math, p, st))) return rc;
        if (l16) {
            const bool last = i + 1 == d.num_conv;
            if ((rc = si_launch_layernorm(ctx, cbuf[0], nullptr, wf(ctx, c.ln_g), wf(ctx, c.ln_b), last ? cbuf[1] : nullptr, (long)B * e.L[i + 1], d.conv_dim[i],
                                          1e-5f, 1, st, last ? nullptr : cb16[1])))
                return rc;
            cur = 1;
            continue;
        }
        cur ^= 1;
        if (d.feat_norm_layer &&
            (rc = si_launch_layernorm(ctx, cbuf[cur], nullptr, wf(ctx, c.ln_g), wf(ctx, c.ln_b), cbuf[cur], (long)B * e.L[i + 1], d.conv_dim[i], 1e-5f, 1, st)))
            return rc;
    }
    const float* feat = cbuf[cur];                                   // (B, T, CF)
    if (vl) {                                                        // -> packed rows (sum of T_b, CF): everything behind is row-wise
        if ((rc = si_launch_repack_rows(ctx, cbuf[cur], cbuf[cur ^ 1], B, T, CF, d_rowoff, false, st))) return rc;
        feat = cbuf[cur ^ 1];
    }
    if ((rc = si_tap(ctx, "features", feat, BT * CF, st))) return rc;
    // A3: LN + projection
    const float* pin = feat;
    if (d.feat_proj_layer_norm) {
        // (bf16 mode: the projection reads the bf16 operand only)
        if ((rc = si_launch_layernorm(ctx, feat, nullptr, wf(ctx, L.fp_ln_g), wf(ctx, L.fp_ln_b), lnf16 ? nullptr : lnf, BT, CF, d.layer_norm_eps, 0, st, lnf16))) return rc;
        pin = lnf;
    }
    if ((rc = linear(ctx, L.proj, pin, h, BT, SI_ACT_NONE, nullptr, st, d.feat_proj_layer_norm ? lnf16 : nullptr))) return rc;
    if ((rc = si_tap(ctx, "projected", h, BT * H, st))) return rc;
    // right-padded batches: padded frames of the projected states are zeroed before the positional conv and excluded
    // as attention keys (modeling_hubert.py:428-437 / 573-582)
    if (vframes && (rc = si_launch_zero_padded_rows(ctx, h, B, T, H, vframes, st))) return rc;
    // A4: h2 = h + gelu(pos_conv(h) + b)
    bool pos_done = false;
    if (d.encoder_math == SI_MATH_BF16 && ctx->opt_enc_posconv && L.pos.has_bias) {
        // the dedicated kernel (posconv.hip): N = the group's own width, the group's weights once per workgroup
        const int prc = si_launch_posconv(ctx, h, h2, ctx->wdev + L.pos.w, wf(ctx, L.pos.bias), B, T, T, H, d.pos_conv_groups, d.pos_conv_kernel,
                                          L.pos.Npad, d.pos_conv_kernel / 2, d_rowoff, vl ? dL(d.num_conv) : nullptr, (double)BT, st);
        if (prc < 0) return prc;
        pos_done = prc == 0;
    }
    if (!pos_done) {
        TapGemmParams p = gemm_params(ctx, L.pos);
        p.x = h; p.out = h2; p.res = h;
        p.nseg = B; p.Lin = T; p.M = T; p.ldx = H; p.x_seg_stride = (long)T * H;
        p.pad = d.pos_conv_kernel / 2; p.ldo = H; p.o_seg_stride = (long)T * H; p.olimit = p.o_seg_stride;
        p.act = SI_ACT_GELU;
        if (vl) {                                            // packed rows: the conv's zero padding begins at each clip's own last frame
            p.seg_lin = dL(d.num_conv); p.seg_m = dL(d.num_conv); p.seg_orows = dL(d.num_conv); p.olim_mul = H; p.seg_row_off = d_rowoff;
            p.algo_macs = (double)BT * H * (double)(H / d.pos_conv_groups) * d.pos_conv_kernel;
        }
        if ((rc = si_launch_tapgemm(ctx, L.pos.math, p, st))) return rc;
    }
    const float eps = d.layer_norm_eps;
    ResLn cur_ln;                                                      // ln_fuse: the LayerNorm whose (unwritten) output is the current hidden state
    if (!d.stable_layer_norm) {
        if ((rc = si_launch_layernorm(ctx, h2, nullptr, wf(ctx, L.enc_ln_g), wf(ctx, L.enc_ln_b), ln_fuse ? nullptr : h, BT, H, eps, 0, st, h16, ln_stats))) return rc;
        cur_ln = ResLn{ln_stats, wf(ctx, L.enc_ln_g), wf(ctx, L.enc_ln_b)};
    } else {
        std::swap(h, h2);
    }
    if ((rc = si_tap(ctx, "encoder_in", h, BT * H, st))) return rc;
    // A5..A8
    for (int l = 0; l < d.num_layers; ++l) {
        const LayerW& Wl = L.layers[l];
        if (!d.stable_layer_norm) {       // post-LN (modeling_hubert.py:371-404); h16 = bf16(h) when e16
            if ((rc = linear(ctx, Wl.qkv, h, qkv_bf16 ? nullptr : qkv, BT, SI_ACT_NONE, nullptr, st, h16, qkv_bf16 ? qkv16 : nullptr))) return rc;
            if (qkv_bf16) rc = si_launch_attention_bf16in(ctx, qkv16, B, T, H, d.num_heads, st, att16, vframes, d_rowoff, sum_t2);
            else rc = si_launch_attention(ctx, qkv, att, B, T, H, d.num_heads, st, att16, ctx->opt_att_bf16, vframes, d_rowoff, sum_t2);
            if (rc) return rc;
            if (ln_fuse) {
                // h2 holds the rows the current hidden state was normalised FROM; out-proj / FFN2 add their residual LayerNorm(h2)
                // from it and write the next pre-LN sum over it.  The last layer's (or the asked-for layer's) output is written.
                const bool want_rows = l + 1 == d.num_layers || output_layer == l + 1;
                if ((rc = linear(ctx, Wl.out, att, h2, BT, SI_ACT_NONE, h2, st, att16, nullptr, 0, 0, cur_ln))) return rc;
                if ((rc = si_launch_layernorm(ctx, h2, nullptr, wf(ctx, Wl.ln1_g), wf(ctx, Wl.ln1_b), nullptr, BT, H, eps, 0, st, h16, ln_stats))) return rc;
                cur_ln = ResLn{ln_stats, wf(ctx, Wl.ln1_g), wf(ctx, Wl.ln1_b)};
                if ((rc = linear(ctx, Wl.ffn1, h, nullptr, BT, SI_ACT_GELU, nullptr, st, h16, ffn16, 0, ffn_ld))) return rc;
                if ((rc = linear(ctx, Wl.ffn2, ffn, h2, BT, SI_ACT_NONE, h2, st, ffn16, nullptr, ffn_ld, 0, cur_ln))) return rc;
                if ((rc = si_launch_layernorm(ctx, h2, nullptr, wf(ctx, Wl.ln2_g), wf(ctx, Wl.ln2_b), want_rows ? h : nullptr, BT, H, eps, 0, st, h16, ln_stats))) return rc;
                cur_ln = ResLn{ln_stats, wf(ctx, Wl.ln2_g), wf(ctx, Wl.ln2_b)};
            } else {
            if ((rc = linear(ctx, Wl.out, att, h2, BT, SI_ACT_NONE, h, st, att16))) return rc;
            if ((rc = si_launch_layernorm(ctx, h2, nullptr, wf(ctx, Wl.ln1_g), wf(ctx, Wl.ln1_b), h, BT, H, eps, 0, st, h16))) return rc;
            if ((rc = linear(ctx, Wl.ffn1, h, e16 ? nullptr : ffn, BT, SI_ACT_GELU, nullptr, st, h16, ffn16, 0, e16 ? ffn_ld : 0))) return rc;
            if ((rc = linear(ctx, Wl.ffn2, ffn, h2, BT, SI_ACT_NONE, h, st, ffn16, nullptr, e16 ? ffn_ld : 0))) return rc;
            if ((rc = si_launch_layernorm(ctx, h2, nullptr, wf(ctx, Wl.ln2_g), wf(ctx, Wl.ln2_b), h, BT, H, eps, 0, st, h16))) return rc;
            }
        } else {                          // pre-LN "stable" (modeling_hubert.py:504-547); residual adds are in place
            // (bf16 mode: the normalised rows feed GEMMs only -- their bf16 operand is all that is written)
            if ((rc = si_launch_layernorm(ctx, h, nullptr, wf(ctx, Wl.ln1_g), wf(ctx, Wl.ln1_b), e16 ? nullptr : h2, BT, H, eps, 0, st, h16))) return rc;
            if ((rc = linear(ctx, Wl.qkv, h2, qkv_bf16 ? nullptr : qkv, BT, SI_ACT_NONE, nullptr, st, h16, qkv_bf16 ? qkv16 : nullptr))) return rc;
            if (qkv_bf16) rc = si_launch_attention_bf16in(ctx, qkv16, B, T, H, d.num_heads, st, att16, vframes, d_rowoff, sum_t2);
            else rc = si_launch_attention(ctx, qkv, att, B, T, H, d.num_heads, st, att16, ctx->opt_att_bf16, vframes, d_rowoff, sum_t2);
            if (rc) return rc;
            if ((rc = linear(ctx, Wl.out, att, h, BT, SI_ACT_NONE, h, st, att16))) return rc;
            if ((rc = si_launch_layernorm(ctx, h, nullptr, wf(ctx, Wl.ln2_g), wf(ctx, Wl.ln2_b), e16 ? nullptr : h2, BT, H, eps, 0, st, h16))) return rc;
            if ((rc = linear(ctx, Wl.ffn1, h2, e16 ? nullptr : ffn, BT, SI_ACT_GELU, nullptr, st, h16, ffn16, 0, e16 ? ffn_ld : 0))) return rc;
            if ((rc = linear(ctx, Wl.ffn2, ffn, h, BT, SI_ACT_NONE, h, st, ffn16, nullptr, e16 ? ffn_ld : 0))) return rc;
        }
        if (output_layer == l + 1) {
            // fairseq `extract_features(output_layer = L)` (I_da/src/hubert_feature_reader.py:60-65): the loop stops after layer
            // L - 1 and returns its output; the pre-LN flavour's final LayerNorm is applied only when no layer was asked for
            SI_HIP_CHECK(hipMemcpyAsync(out_hidden, h, (size_t)BT * H * sizeof(float), hipMemcpyDeviceToDevice, st));
            return SI_OK;
        }
    }
    if (d.stable_layer_norm) {
        if ((rc = si_launch_layernorm(ctx, h, nullptr, wf(ctx, L.enc_ln_g), wf(ctx, L.enc_ln_b), h2, BT, H, eps, 0, st))) return rc;
        std::swap(h, h2);
    }
    if ((rc = si_tap(ctx, "last_hidden", h, BT * H, st))) return rc;
    // A9: final_layers = LN -> Linear(H, codebook_dim), always fp32
    if ((rc = si_launch_layernorm(ctx, h, nullptr, wf(ctx, L.head_ln_g), wf(ctx, L.head_ln_b), h2, BT, H, 1e-5f, 0, st))) return rc;
    if (!vl) return linear(ctx, L.head, h2, out_feats, BT, SI_ACT_NONE, nullptr, st);
    // ragged batch: the head on the packed rows (into the dead q|k|v storage), then out to (B, T, D) with zero rows past each clip
    if ((rc = linear(ctx, L.head, h2, qkv, BT, SI_ACT_NONE, nullptr, st))) return rc;
    return si_launch_repack_rows(ctx, qkv, out_feats, B, T, d.codebook_dim, d_rowoff, true, st);
}

extern "C" {

int si_codebook_splice(si_ctx* ctx, const float* feats, int B, int T, const int32_t* frame_pos, int Lm, float* mel, int Tm,
                       int64_t* labels, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!ctx->weights_ready) return si_fail(ctx, SI_ESTATE, "si_codebook_splice before weights were loaded");
    if (!ctx->weights_verified) if (int rcw = si_weights_check(ctx)) return rcw;
    if (!feats || !frame_pos || !mel || B <= 0 || Lm < 0) return si_fail(ctx, SI_EINVAL, "si_codebook_splice: NULL / empty argument");
    if (ctx->d.codebook_dim != ctx->d.num_mels)
        return si_fail(ctx, SI_EINVAL, "codebook_dim %d != generator input width %d: centroids are not frames of this generator's input",
                       ctx->d.codebook_dim, ctx->d.num_mels);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    const Layout& L = ctx->lay;
    return si_launch_codebook_splice(ctx, feats, B, T, ctx->d.codebook_dim, frame_pos, Lm, wf(ctx, L.cb_centered), wf(ctx, L.cb_raw),
                                     wf(ctx, L.cb_rnorm), ctx->d.num_clusters, mel, Tm, labels, static_cast<hipStream_t>(stream));
}

int si_codebook_splice_varlen(si_ctx* ctx, const float* feats, int B, int T, const int32_t* frame_pos, const int32_t* frame_cnt, int Lm,
                              float* mel, int Tm, int64_t* labels, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!ctx->weights_ready) return si_fail(ctx, SI_ESTATE, "si_codebook_splice_varlen before weights were loaded");
    if (!feats || !frame_pos || !frame_cnt || !mel || B <= 0 || Lm < 0) return si_fail(ctx, SI_EINVAL, "si_codebook_splice_varlen: NULL / empty argument");
    if (ctx->d.codebook_dim != ctx->d.num_mels)
        return si_fail(ctx, SI_EINVAL, "codebook_dim %d != generator input width %d", ctx->d.codebook_dim, ctx->d.num_mels);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    const Layout& L = ctx->lay;
    return si_launch_codebook_splice(ctx, feats, B, T, ctx->d.codebook_dim, frame_pos, Lm, wf(ctx, L.cb_centered), wf(ctx, L.cb_raw),
                                     wf(ctx, L.cb_rnorm), ctx->d.num_clusters, mel, Tm, labels, static_cast<hipStream_t>(stream), frame_cnt);
}

int si_codebook_splice_labels(si_ctx* ctx, const int64_t* labels, int B, const int32_t* frame_pos, int Lm, float* mel, int Tm,
                              si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!ctx->weights_ready) return si_fail(ctx, SI_ESTATE, "si_codebook_splice_labels before weights were loaded");
    if (!labels || !frame_pos || !mel || B <= 0 || Lm < 0) return si_fail(ctx, SI_EINVAL, "si_codebook_splice_labels: NULL / empty argument");
    if (ctx->d.codebook_dim != ctx->d.num_mels)
        return si_fail(ctx, SI_EINVAL, "codebook_dim %d != generator input width %d", ctx->d.codebook_dim, ctx->d.num_mels);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_codebook_gather(ctx, labels, B, ctx->d.codebook_dim, frame_pos, Lm, wf(ctx, ctx->lay.cb_raw), ctx->d.num_clusters,
                                     mel, Tm, static_cast<hipStream_t>(stream));
}

int si_codebook_metrics(si_ctx* ctx, const float* feats, int B, int T, const int32_t* frame_pos, int Lm,
                        const int64_t* target_labels, float* loss_terms, float* loss, int64_t* pred_labels, float* cos_pred_target,
                        si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!ctx->weights_ready) return si_fail(ctx, SI_ESTATE, "si_codebook_metrics before weights were loaded");
    if (!feats || !frame_pos || !target_labels || !loss_terms || !loss || !cos_pred_target || B <= 0 || Lm <= 0)
        return si_fail(ctx, SI_EINVAL, "si_codebook_metrics: NULL / empty argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    const Layout& L = ctx->lay;
    return si_launch_codebook_metrics(ctx, feats, B, T, ctx->d.codebook_dim, frame_pos, Lm, wf(ctx, L.cb_centered), wf(ctx, L.cb_rnorm),
                                      ctx->d.num_clusters, target_labels, loss_terms, loss, pred_labels, cos_pred_target,
                                      static_cast<hipStream_t>(stream));
}

int si_kmeans_assign(si_ctx* ctx, const float* feats, int64_t rows, int D, const float* centroids, int K, int64_t* labels,
                     float* sq_dist, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!feats || !centroids || !labels || rows < 0) return si_fail(ctx, SI_EINVAL, "si_kmeans_assign: NULL / bad argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    if (K > ctx->km_cnorm_cap) {                                       // (a context is single-threaded and its calls stream-ordered by contract)
        if (ctx->km_cnorm) { SI_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream))); SI_HIP_CHECK(hipFree(ctx->km_cnorm)); ctx->km_cnorm = nullptr; ctx->km_cnorm_cap = 0; }
        const int cap = std::max(K, 1024);
        SI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->km_cnorm), (size_t)cap * sizeof(float)));
        ctx->km_cnorm_cap = cap;
    }
    const bool mfma = !(getenv("SI_KMEANS_MFMA") && atoi(getenv("SI_KMEANS_MFMA")) == 0);   // (read per call: the test compares both kernels)
    return si_launch_kmeans_assign(ctx, feats, (long)rows, D, centroids, K, labels, sq_dist, static_cast<hipStream_t>(stream),
                                   mfma ? ctx->km_cnorm : nullptr);
}

int si_mel_metrics(si_ctx* ctx, const float* mel_a, const float* mel_b, int B, int D, int L, const float* center, float* out3,
                   si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!mel_a || !mel_b || !out3) return si_fail(ctx, SI_EINVAL, "si_mel_metrics: NULL argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_mel_metrics(ctx, mel_a, mel_b, B, D, L, center, out3, static_cast<hipStream_t>(stream));
}

int si_sisdr(si_ctx* ctx, const float* est, const float* ref, int B, int n, float* out, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!est || !ref || !out) return si_fail(ctx, SI_EINVAL, "si_sisdr: NULL argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_sisdr(ctx, est, ref, B, n, out, static_cast<hipStream_t>(stream));
}

// ---- F0 VQ-VAE encoder (row f-2): I_da/src/model.py:160-163 runs `self.fo_vqvae.encoder(fo)` -- jukebox.py `Encoder` with one
//      level = `EncoderConvBlock` (:11-116): down_t x [Conv1d(k = 2 s, stride s, pad s / 2) + Resnet1D(depth blocks of
//      x + Conv1(ReLU(Conv3_dil(ReLU(x)))), dilation growth^j, resnet.py:29-97)] + Conv1d(width -> out, 3, 1, 1).
namespace {
struct F0Shape { int k, pad; };
F0Shape f0_down_shape(int s) { return (s % 2 == 0) ? F0Shape{2 * s, s / 2} : F0Shape{2 * s + 1, s / 2 + 1}; }   // jukebox.py:54-57
bool f0_desc_ok(const si_f0enc_desc* d) {
    return d && d->in_width >= 1 && d->out_width >= 1 && d->width >= 1 && d->n_state >= 1 && d->depth >= 0 && d->down_t >= 1 && d->stride_t >= 1 &&
           d->dilation_growth >= 1 && d->down_t <= 16 && d->depth <= 32;
}
}  // namespace

size_t si_f0_encoder_weight_floats(const si_f0enc_desc* d) {
    if (!f0_desc_ok(d)) return 0;
    const F0Shape ds = f0_down_shape(d->stride_t);
    size_t n = 0;
    for (int i = 0; i < d->down_t; ++i) {
        n += (size_t)d->width * (i == 0 ? d->in_width : d->width) * ds.k + d->width;
        n += (size_t)d->depth * ((size_t)d->n_state * d->width * 3 + d->n_state + (size_t)d->width * d->n_state + d->width);
    }
    return n + (size_t)d->out_width * d->width * 3 + d->out_width;
}

int si_f0_encoder_frames(const si_f0enc_desc* d, int T) {
    if (!f0_desc_ok(d) || T <= 0) return 0;
    const F0Shape ds = f0_down_shape(d->stride_t);
    for (int i = 0; i < d->down_t; ++i) {
        if (T + 2 * ds.pad < ds.k) return 0;                           // (torch's Conv1d raises: the padded input is shorter than the kernel)
        T = (T + 2 * ds.pad - ds.k) / d->stride_t + 1;
    }
    return T;
}

size_t si_f0_encoder_workspace_bytes(const si_f0enc_desc* d, int B, int T) {
    if (!f0_desc_ok(d) || B <= 0 || T <= 0) return 0;
    const F0Shape ds = f0_down_shape(d->stride_t);
    const int T1 = (T + 2 * ds.pad - ds.k) / d->stride_t + 1;          // the longest intermediate
    const size_t c = (size_t)std::max(d->width, d->n_state);
    return 3 * ((size_t)B * c * std::max(T1, 1) * sizeof(float) + 256);
}

int si_f0_encoder_forward(si_ctx* ctx, const si_f0enc_desc* d, const float* weights, const float* f0, int B, int T, float* h_out,
                          void* workspace, size_t workspace_bytes, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!f0_desc_ok(d)) return si_fail(ctx, SI_EINVAL, "si_f0_encoder_forward: bad descriptor");
    if (!weights || !f0 || !h_out || !workspace || B <= 0 || T <= 0) return si_fail(ctx, SI_EINVAL, "si_f0_encoder_forward: NULL / empty argument");
    if (si_f0_encoder_frames(d, T) <= 0) return si_fail(ctx, SI_EINVAL, "si_f0_encoder_forward: %d frames are too few for %d stride-%d convolutions", T, d->down_t, d->stride_t);
    const size_t need = si_f0_encoder_workspace_bytes(d, B, T);
    if (workspace_bytes < need) return si_fail(ctx, SI_ENOMEM, "workspace of %zu bytes < %zu needed", workspace_bytes, need);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const F0Shape ds = f0_down_shape(d->stride_t);
    {
        // one persistent launch with the track's activations in LDS (bit-identical to the layer-by-layer form below, which
        // remains for tracks too long for LDS); SI_F0_FUSED=0 forces the latter (the test compares the two)
        const bool fused_on = !(getenv("SI_F0_FUSED") && atoi(getenv("SI_F0_FUSED")) == 0);   // (read per call: the test flips it in-process)
        if (fused_on) {
            double macs = 0;
            int Tc = T, cin = d->in_width;
            for (int i = 0; i < d->down_t; ++i) {
                const int To = (Tc + 2 * ds.pad - ds.k) / d->stride_t + 1;
                macs += (double)B * To * d->width * (double)cin * ds.k + (double)d->depth * B * To * ((double)d->n_state * d->width * 3 + (double)d->width * d->n_state);
                Tc = To; cin = d->width;
            }
            macs += (double)B * Tc * d->out_width * (double)d->width * 3;
            const int frc = si_launch_f0enc_fused(ctx, weights, f0, B, T, h_out, d->in_width, d->out_width, d->width, d->n_state, d->depth, d->down_t,
                                                  d->stride_t, d->dilation_growth, ds.k, ds.pad, macs, st);
            if (frc <= 0) return frc;
        }
    }
    const size_t slot = need / 3 / sizeof(float);
    float* buf[3] = {static_cast<float*>(workspace), static_cast<float*>(workspace) + slot, static_cast<float*>(workspace) + 2 * slot};
    const float* w = weights;
    const float* x = f0;                                               // (B, in_width, T)
    int cin = d->in_width, Tc = T, cur = -1;
    int rc;
    for (int i = 0; i < d->down_t; ++i) {
        const int To = (Tc + 2 * ds.pad - ds.k) / d->stride_t + 1;
        const float* cw = w; w += (size_t)d->width * cin * ds.k;
        const float* cb = w; w += d->width;
        const int o = (cur + 1) % 3;
        if ((rc = si_launch_small_conv1d(ctx, x, cw, cb, nullptr, buf[o], B, cin, Tc, d->width, To, ds.k, d->stride_t, 1, ds.pad, 0, 0, st))) return rc;
        cur = o; x = buf[cur]; cin = d->width; Tc = To;
        long dil = 1;
        for (int j = 0; j < d->depth; ++j) {
            const float* w3 = w; w += (size_t)d->n_state * d->width * 3;
            const float* b3 = w; w += d->n_state;
            const float* w1 = w; w += (size_t)d->width * d->n_state;
            const float* b1 = w; w += d->width;
            const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
            if ((rc = si_launch_small_conv1d(ctx, x, w3, b3, nullptr, buf[t1], B, d->width, Tc, d->n_state, Tc, 3, 1, (int)dil, (int)dil, 1, 0, st))) return rc;
            if ((rc = si_launch_small_conv1d(ctx, buf[t1], w1, b1, x, buf[t2], B, d->n_state, Tc, d->width, Tc, 1, 1, 1, 0, 1, 0, st))) return rc;
            cur = t2; x = buf[cur];
            dil *= d->dilation_growth;
            if (dil > (1 << 20)) return si_fail(ctx, SI_EINVAL, "si_f0_encoder_forward: dilation overflow");
        }
    }
    const float* fw = w; w += (size_t)d->out_width * d->width * 3;
    return si_launch_small_conv1d(ctx, x, fw, w, nullptr, h_out, B, d->width, Tc, d->out_width, Tc, 3, 1, 1, 1, 0, 1, st);
}

int si_unit_frontend(si_ctx* ctx, const int64_t* code, int Fc, const int64_t* f0_code, int Fp, const float* spk_emb, const float* emb_c,
                     int Kc, const float* emb_p, int Kp, int E, int B, float* out, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!code || !emb_c || !out || B <= 0 || Fc <= 0 || E <= 0 || Kc <= 0) return si_fail(ctx, SI_EINVAL, "si_unit_frontend: NULL / empty argument");
    if (f0_code && (!emb_p || Fp <= 0 || Kp <= 0)) return si_fail(ctx, SI_EINVAL, "si_unit_frontend: f0_code needs its embedding table");
    if (f0_code) {
        const int F = Fp > Fc ? Fp : Fc, s = Fp > Fc ? Fc : Fp;
        // `_upsample` repeats each frame F // len times and refuses anything that does not fill F (I_da/src/model.py:104-112)
        if (F % s) return si_fail(ctx, SI_EINVAL, "si_unit_frontend: %d and %d frames: misalignment between condition features", Fc, Fp);
    }
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_unit_frontend(ctx, code, Fc, f0_code, Fp, spk_emb, emb_c, Kc, emb_p, Kp, E, B, out, static_cast<hipStream_t>(stream));
}

int si_resample_poly(si_ctx* ctx, const float* x, int B, int n_in, const float* taps, int ntaps, int up, int down,
                     int pre_remove, int n_out, float* y, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!x || !taps || !y || B <= 0 || n_in <= 0) return si_fail(ctx, SI_EINVAL, "si_resample_poly: NULL / empty argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_resample_poly(ctx, x, B, n_in, taps, ntaps, up, down, pre_remove, n_out, y, static_cast<hipStream_t>(stream));
}

static int hifigan_run(si_ctx* ctx, const float* mel, int B, int Tm, int stretch, float* wav_out, void* workspace,
                       size_t workspace_bytes, si_stream_t stream, const int32_t* host_len);

int si_resample_sinc(si_ctx* ctx, const float* x, const int32_t* n_len, int B, int n_in, const si_sinc_filter* f, int n_out, float* y,
                     si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!x || !y || !f || !f->win || !f->dwin || !f->time_reg || B <= 0 || n_in <= 0) return si_fail(ctx, SI_EINVAL, "si_resample_sinc: NULL / empty argument");
    if (f->struct_size != (int32_t)sizeof(si_sinc_filter)) return si_fail(ctx, SI_EINVAL, "si_resample_sinc: si_sinc_filter size mismatch");
    if (n_out > f->n_time) return si_fail(ctx, SI_EINVAL, "si_resample_sinc: %d outputs but the time-register table holds %d", n_out, f->n_time);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_resample_sinc(ctx, x, n_len, B, n_in, f->win, f->dwin, f->nwin, f->num_table, f->step, f->scale, f->ratio, f->time_reg, n_out, y,
                                   static_cast<hipStream_t>(stream));
}

int si_extend_mel(si_ctx* ctx, const float* mel, int B, int Tm, float* out, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!mel || !out || B <= 0 || Tm <= 0) return si_fail(ctx, SI_EINVAL, "si_extend_mel: NULL / empty argument");
    const long Tout = voc_tout(Tm, 1);
    if (Tout < 1) return si_fail(ctx, SI_EINVAL, "mel of %d frames stretches to nothing", Tm);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_extend_mel_cf(ctx, mel, B, ctx->d.num_mels, Tm, (int)Tout, out, static_cast<hipStream_t>(stream));
}

int si_pcm16(si_ctx* ctx, const float* wav, int64_t n, int16_t* out, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!wav || !out || n < 0) return si_fail(ctx, SI_EINVAL, "si_pcm16: NULL / bad argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    return si_launch_pcm16(ctx, wav, (long)n, out, static_cast<hipStream_t>(stream));
}

int si_hifigan_forward(si_ctx* ctx, const float* mel, int B, int Tm, int stretch, float* wav_out, void* workspace,
                       size_t workspace_bytes, si_stream_t stream) {
    return hifigan_run(ctx, mel, B, Tm, stretch, wav_out, workspace, workspace_bytes, stream, nullptr);
}

int si_hifigan_forward_varlen(si_ctx* ctx, const float* mel, const int32_t* mel_len, int B, int Tm, int stretch, float* wav_out, void* workspace,
                              size_t workspace_bytes, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!mel_len) return si_fail(ctx, SI_EINVAL, "si_hifigan_forward_varlen: NULL lengths");
    return hifigan_run(ctx, mel, B, Tm, stretch, wav_out, workspace, workspace_bytes, stream, mel_len);
}

// host_len (B, HOST) or null: ragged batch -- clip b holds host_len[b] mel frames of its Tm; every convolution's zero padding
// begins at the clip's OWN end, so its waveform equals that clip's alone; wav_out rows are the longest clip's, zero past each clip.
static int hifigan_run(si_ctx* ctx, const float* mel, int B, int Tm, int stretch, float* wav_out, void* workspace,
                       size_t workspace_bytes, si_stream_t stream, const int32_t* host_len) {
    if (!ctx) return SI_EINVAL;
    if (!ctx->weights_ready) return si_fail(ctx, SI_ESTATE, "si_hifigan_forward before weights were loaded");
    if (!ctx->weights_verified) if (int rcw = si_weights_check(ctx)) return rcw;
    if (!mel || !wav_out || !workspace || B <= 0 || Tm <= 0) return si_fail(ctx, SI_EINVAL, "si_hifigan_forward: NULL / empty argument");
    const si_model_desc& d = ctx->d;
    const Layout& Ly = ctx->lay;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t need = vocoder_ws_bytes(ctx, B, Tm, stretch);
    if (workspace_bytes < need) return si_fail(ctx, SI_ENOMEM, "workspace of %zu bytes < %zu needed for B=%d Tm=%d", workspace_bytes, need, B, Tm);
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    const int Bc_max = vocoder_chunk(d, B);
    const long Tout = voc_tout(Tm, stretch);
    if (Tout < 1) return si_fail(ctx, SI_EINVAL, "mel of %d frames stretches to nothing", Tm);
    size_t lc_max = (size_t)Tout * d.up_initial_channel;
    {
        long L = Tout; int c = d.up_initial_channel;
        for (int i = 0; i < d.num_ups; ++i) { L *= d.up_rates[i]; c = stage_channels(ctx, i); lc_max = std::max(lc_max, (size_t)L * c); }
    }
    const long Lwav = si_vocoder_samples(ctx, Tm, stretch);
    Carver W{static_cast<char*>(workspace), workspace_bytes};
    // ---- ragged batch: the table [mel frames | rows of stage 0 (stretched frames) .. stage num_ups | GEMM rows of upsampler 1 .. num_ups]
    const bool vl = host_len != nullptr;
    const int32_t* h_tab = nullptr;
    int32_t* d_tab = nullptr;
    if (vl) {
        std::vector<int32_t>& tab = ctx->vl_host;
        tab.assign((size_t)(2 * d.num_ups + 2) * B, 0);
        for (int b = 0; b < B; ++b) {
            if (host_len[b] < 1 || host_len[b] > Tm) return si_fail(ctx, SI_EINVAL, "ragged batch: clip %d holds %d mel frames, outside 1..%d", b, host_len[b], Tm);
            long Lb = voc_tout(host_len[b], stretch);
            if (Lb < 1) return si_fail(ctx, SI_EINVAL, "ragged batch: clip %d (%d mel frames) stretches to nothing", b, host_len[b]);
            tab[b] = host_len[b];
            tab[(size_t)B + b] = (int32_t)Lb;
            for (int i = 0; i < d.num_ups; ++i) {
                const int u = d.up_rates[i], pad = (d.up_kernels[i] - u) / 2;
                Lb *= u;
                tab[(size_t)(2 + i) * B + b] = (int32_t)Lb;
                tab[(size_t)(2 + d.num_ups + i) * B + b] = (int32_t)((pad + Lb - 1) / u + 1);
            }
        }
        d_tab = reinterpret_cast<int32_t*>(W.bytes(tab.size() * 4));
        if (!W.ok) return si_fail(ctx, SI_ENOMEM, "internal: vocoder workspace carve exceeded its own estimate");
        if (int rc = vl_upload(ctx, d_tab, st)) return rc;
        h_tab = tab.data();
        SI_HIP_CHECK(hipMemsetAsync(wav_out, 0, (size_t)B * Lwav * sizeof(float), st));   // samples past a clip's own end read as silence
    }
    float* ext_ws = W.floats((size_t)Bc_max * Tout * Ly.mel_ld);
    float* buf_ws[6];
    unsigned short* h16_ws[6];
    for (auto& b : buf_ws) b = W.floats((size_t)Bc_max * lc_max);
    for (auto& b : h16_ws) b = reinterpret_cast<unsigned short*>(W.bytes((size_t)Bc_max * lc_max * 2));
    // Operand-ready activations (bf16 / fp16 vocoder): every producer also writes type16(leaky_relu(x, 0.1)) -- exactly
    // what the next convolution would compute while staging -- so consumers copy 2-byte operands instead of loading
    // fp32 and converting; the ResBlock intermediate exists only in that form.  Same arithmetic, bit-identical output;
    // 20-40 % less HBM / L2 traffic on the convolution inputs.  SI_VOC_OPREADY=0 restores the fp32-input path.
    const bool opr = ctx->opt_voc_opready && (d.vocoder_math == SI_MATH_BF16 || d.vocoder_math == SI_MATH_F16);
    // fp16 activation stream (fp16 mode): activations live ONLY as raw fp16 -- the residual stream too -- and the
    // consumer applies its leaky-ReLU to the packed halves while staging: 10 instead of 16 bytes of HBM traffic per
    // element and conv pair (these stages run on the memory side in fp16).  The rounding it adds (fp16 after every
    // residual add) is small next to the operand rounding the mode already has: waveform RMS error 1.35e-4 vs 1.10e-4
    // (gate 1e-3).  SI_VOC_RES16=0 keeps the fp32 residual stream.
    const bool r16 = opr && ctx->opt_voc_res16 && d.vocoder_math == SI_MATH_F16;
    const int fuse_mask = ctx->opt_voc_fuse == 1 ? ~0 : ctx->opt_voc_fuse;
    if (!W.ok) return si_fail(ctx, SI_ENOMEM, "internal: vocoder workspace carve exceeded its own estimate");
    const int nk = d.num_rb;
    static const char* upn[] = {"ups0", "ups1", "ups2", "ups3", "ups4", "ups5", "ups6", "ups7"};
    static const char* stn[] = {"stage0", "stage1", "stage2", "stage3", "stage4", "stage5", "stage6", "stage7"};

    // the generator on clips [b0, b0 + Bc) with its own scratch, enqueued on stream `st`
    auto run = [&](int b0, int Bc, float* ext, float* const* buf, unsigned short* const* h16, hipStream_t st) -> int {
        int rc;
        // ragged batch: device / host rows of the chunk's clips -- dTm mel frames, dLs(s) rows at stage s (0 = stretched frames), dMt(i) GEMM rows of upsampler i
        const int32_t* dTm = vl ? d_tab + b0 : nullptr;
        auto dLs = [&](int sidx) -> const int32_t* { return vl ? d_tab + (size_t)(1 + sidx) * B + b0 : nullptr; };
        auto hLs = [&](int sidx) -> const int32_t* { return vl ? h_tab + (size_t)(1 + sidx) * B + b0 : nullptr; };
        auto dMt = [&](int i) -> const int32_t* { return vl ? d_tab + (size_t)(2 + d.num_ups + i) * B + b0 : nullptr; };
        auto hMt = [&](int i) -> const int32_t* { return vl ? h_tab + (size_t)(2 + d.num_ups + i) * B + b0 : nullptr; };
        auto rows_of = [&](const int32_t* h, long uniform) { if (!h) return (double)Bc * uniform; double r = 0; for (int b = 0; b < Bc; ++b) r += h[b]; return r; };
        // a same-length convolution at stage sidx: every clip reads and writes its own rows
        auto seg_conv = [&](TapGemmParams& q, int sidx) { if (vl) { q.seg_lin = q.seg_m = q.seg_orows = dLs(sidx); q.olim_mul = q.ldo; q.seg_m_host = hLs(sidx); } };
        // A14: stretch + transpose to channels-last
        if ((rc = si_launch_extend_mel(ctx, mel + (size_t)b0 * d.num_mels * Tm, Bc, d.num_mels, Tm, (int)Tout, stretch, ext, Ly.mel_ld, st, dTm, dLs(0)))) return rc;
        // Will upsampler i run on gemmcu.hip's TC instantiations?  Its producer (conv_pre / the last launch of the previous stage's MRF
        // sum) then stores leaky_relu(x, 0.1) -- the upsampler's own first statement (models.py:110) applied once to the fp32 value
        // instead of to every fragment it is read into -- and the upsampler takes its input as it is.
        auto ups_on_gemmcu = [&](int i, long Lc_i, int c_i) -> bool {
            if (!r16 || !ctx->opt_voc_upsgemm || i >= d.num_ups) return false;
            const GemmW& G = Ly.ups[i];
            if (G.math != SI_MATH_F16 || !G.has_bias) return false;
            const int u = d.up_rates[i], k = d.up_kernels[i], cout = stage_channels(ctx, i), pad = (k - u) / 2;
            const long Lo_i = Lc_i * u;
            TapGemmParams p = gemm_params(ctx, G);
            p.x16 = h16[0]; p.out16 = h16[2]; p.out16_slope = 1.f;
            p.nseg = Bc; p.Lin = (int)Lc_i; p.M = (int)((pad + Lo_i - 1) / u + 1); p.ldx = c_i; p.x_seg_stride = Lc_i * c_i;
            p.dil = -1; p.ldo = u * cout; p.o_seg_stride = Lo_i * cout; p.ooff = -(long)pad * cout; p.olimit = Lo_i * cout; p.pro_slope = 0.1f;
            if (vl) { p.seg_lin = dLs(i); p.seg_m = dMt(i); p.seg_orows = dLs(i + 1); p.olim_mul = cout; p.seg_m_host = hMt(i); }
            return si_gemmcu_tc_covers(ctx, p, true);                 // by the layer's geometry alone: a clip's samples must not depend on its batch
        };
        // B1: conv_pre
        float* x = buf[0];
        float* xs = buf[1];
        unsigned short *x16 = h16[0], *xs16 = h16[1], *U16 = h16[2], *t16 = h16[3];
        {
            TapGemmParams p = gemm_params(ctx, Ly.pre);
            p.x = ext; p.out = x;
            if (opr) { p.out16 = x16; p.out16_slope = (r16 && !ups_on_gemmcu(0, Tout, d.up_initial_channel)) ? 1.f : 0.1f; }
            if (r16) p.out = nullptr;
            p.nseg = Bc; p.Lin = (int)Tout; p.M = (int)Tout; p.ldx = Ly.mel_ld; p.x_seg_stride = Tout * Ly.mel_ld;
            p.algo_macs = rows_of(hLs(0), Tout) * d.up_initial_channel * (double)d.num_mels * 7;
            p.pad = 3; p.ldo = d.up_initial_channel; p.o_seg_stride = Tout * d.up_initial_channel; p.olimit = p.o_seg_stride;
            seg_conv(p, 0);
            if ((rc = si_launch_tapgemm(ctx, Ly.pre.math, p, st))) return rc;
        }
        long Lc = Tout; int c = d.up_initial_channel;
        for (int i = 0; i < d.num_ups; ++i) {
            const int u = d.up_rates[i], k = d.up_kernels[i], cout = stage_channels(ctx, i), pad = (k - u) / 2;
            const long Lo = Lc * u;
            float* U = buf[2];
            // B2: leaky_relu(0.1) -> ConvTranspose1d as `u` phases of a 2-tap conv: row u' reads input rows u', u'-1, ...
            bool ups_done = false;
            if (r16 && (fuse_mask & 2) && u == 2 && k == 4) {          // (SI_VOC_FUSE: bit 1 = this kernel, the width bits = the ResBlock kernels)
                // the late upsamplers (128 / 64 input channels) are HBM-bound: a persistent streaming kernel (upsample.hip)
                const GemmW& G = Ly.ups[i];
                UpsampleParams q{};
                q.x16 = x16; q.w = reinterpret_cast<const unsigned short*>(ctx->wdev + G.w); q.bias = reinterpret_cast<const float*>(ctx->wdev + G.bias);
                q.out16 = U16; q.B = Bc; q.Lin = (int)Lc; q.M = (int)((pad + Lo - 1) / u + 1); q.Cin = c; q.N = u * cout; q.taps = G.ntaps;
                q.ooff = (long)pad * cout; q.o_clip_stride = Lo * cout; q.o_clip_elems = Lo * cout;
                q.lens_lin = dLs(i); q.lens_m = dMt(i); q.lens_m_host = hMt(i);
                if (G.has_bias && G.Npad == G.N && G.math == SI_MATH_F16) {
                    rc = si_launch_upsample_stream(ctx, q, st);
                    if (rc < 0) return rc;
                    ups_done = rc == 0;
                }
            }
            if (!ups_done) {
                TapGemmParams p = gemm_params(ctx, Ly.ups[i]);
                p.x = x; p.out = U;
                if (opr) { p.x = nullptr; p.x16 = x16; p.out16 = U16; p.out16_slope = r16 ? 1.f : 0.1f; }
                if (r16) p.out = nullptr;
                p.nseg = Bc; p.Lin = (int)Lc; p.M = (int)((pad + Lo - 1) / u + 1); p.ldx = c; p.x_seg_stride = Lc * c;
                p.dil = -1; p.ldo = u * cout; p.o_seg_stride = Lo * cout; p.ooff = -(long)pad * cout; p.olimit = Lo * cout;
                const bool pre_act = ups_on_gemmcu(i, Lc, c);       // (its producer stored the activated input)
                p.pro_slope = ((opr && !r16) || pre_act) ? 1.f : 0.1f;   // operand-ready inputs are already activated
                p.algo_macs = rows_of(hLs(i), Lc) * (double)(d.up_initial_channel >> i) * (double)(d.up_initial_channel >> (i + 1)) * k;   // Cin*Cout*k*Lin (real widths)
                if (vl) { p.seg_lin = dLs(i); p.seg_m = dMt(i); p.seg_orows = dLs(i + 1); p.olim_mul = cout; p.seg_m_host = hMt(i); }
                // the early upsamplers (N = 2048 / 1024) on the fp16 stream are real GEMMs: the one-tile-per-CU kernel (gemmcu.hip, TC
                // mode) wherever it covers the layer's geometry (whatever the batch: the two forms round differently); SI_VOC_UPSGEMM=0: the tap-GEMM
                int urc = 1;
                if (r16 && ctx->opt_voc_upsgemm && Ly.ups[i].math == SI_MATH_F16 && Ly.ups[i].has_bias) urc = si_launch_gemmcu_tc(ctx, p, st, true);
                if (urc < 0) return urc;
                if (urc > 0 && (rc = si_launch_tapgemm(ctx, Ly.ups[i].math, p, st))) return rc;
            }
            if (!r16 && (rc = si_tap(ctx, upn[i], U, (long)Bc * Lo * cout, st))) return rc;
            // B3: multi-receptive-field fusion: mean over the resblocks, accumulated into xs by the last conv of each
            const bool act_next = ups_on_gemmcu(i + 1, Lo, cout);   // the launch that completes the sum stores the next upsampler's activated input
            for (int j = 0; j < nk; ++j) {
                const ResW& R = Ly.rbs[(size_t)i * nk + j];
                const int rk = d.rb_kernels[j];
                const float* y = U;
                const unsigned short* y16 = U16;
                if (d.resblock_type == 2) {
                    // ResBlock2 (I_ea/hifi_gan/models.py:63-68): per dilation x = x + conv_d(lrelu(x)) -- ONE convolution with the
                    // residual (and, on the block's last one, the 1 / num_kernels scale and the MRF accumulate) in its epilogue
                    for (int n = 0; n < d.num_dil; ++n) {
                        const int dl = d.rb_dilations[j][n];
                        const bool last = (n == d.num_dil - 1);
                        float* ynext = last ? xs : buf[4 + (n & 1)];
                        unsigned short* ynext16 = last ? xs16 : h16[4 + (n & 1)];
                        TapGemmParams q = gemm_params(ctx, R.c1[n]);
                        q.x = y; q.out = ynext; q.res = y;
                        if (opr) {
                            q.x = nullptr; q.x16 = y16;
                            if (r16) { q.res = nullptr; q.res16 = y16; q.out = nullptr; }
                            const bool want16 = r16 || !last || (j == nk - 1 && i + 1 < d.num_ups);
                            if (want16) { q.out16 = ynext16; q.out16_slope = (r16 && !(act_next && last && j == nk - 1)) ? 1.f : 0.1f; }
                        }
                        q.pro_slope = (opr && !r16) ? 1.f : 0.1f;          // operand-ready inputs are already activated
                        q.nseg = Bc; q.Lin = (int)Lo; q.M = (int)Lo; q.ldx = cout; q.x_seg_stride = Lo * cout;
                        q.dil = dl; q.pad = dl * (rk - 1) / 2; q.ldo = cout; q.o_seg_stride = Lo * cout; q.olimit = q.o_seg_stride;
                        if (last) { q.alpha = 1.0f / nk; q.accumulate = (j > 0); q.acc16 = (r16 && j > 0); }
                        seg_conv(q, i + 1);
                        if (vl) q.algo_macs = rows_of(hLs(i + 1), Lo) * (double)cout * cout * rk;
                        if ((rc = si_launch_tapgemm(ctx, R.c1[n].math, q, st))) return rc;
                        y = ynext;
                        y16 = ynext16;
                    }
                    continue;
                }
                if (r16 && (fuse_mask & cout) && ctx->opt_voc_chain && d.num_dil == 3 && !(act_next && j == nk - 1)) {
                    // full-rate stage: the whole resblock (three pairs) as one kernel, residual stream in LDS (reschain.hip)
                    ResChainParams cp{};
                    cp.y16 = U16; cp.out16 = xs16; cp.B = Bc; cp.L = (int)Lo; cp.k = rk; cp.alpha = 1.0f / nk; cp.accumulate = j > 0;
                    cp.lens = dLs(i + 1); cp.lens_host = hLs(i + 1);
                    for (int n = 0; n < 3; ++n) {
                        const TapGemmParams w1 = gemm_params(ctx, R.c1[n]), w2 = gemm_params(ctx, R.c2[n]);
                        cp.w1[n] = static_cast<const unsigned short*>(w1.w); cp.w2[n] = static_cast<const unsigned short*>(w2.w);
                        cp.b1[n] = w1.bias; cp.b2[n] = w2.bias; cp.dil[n] = d.rb_dilations[j][n];
                    }
                    const int crc = si_launch_reschain(ctx, cout, cp, st);
                    if (crc < 0) return crc;
                    if (crc == 0) continue;
                }
                for (int n = 0; n < d.num_dil; ++n) {
                    const int dl = d.rb_dilations[j][n];
                    const bool last_n = (n == d.num_dil - 1);
                    if (r16 && (fuse_mask & cout)) {
                        // narrow stages: conv pair as one kernel, the intermediate stays in LDS (respair.hip)
                        const TapGemmParams w1 = gemm_params(ctx, R.c1[n]), w2 = gemm_params(ctx, R.c2[n]);
                        unsigned short* yn16 = last_n ? xs16 : h16[4 + (n & 1)];
                        const int frc = si_launch_respair(ctx, cout, y16, yn16, w1.w, w2.w, w1.bias, w2.bias, Bc, (int)Lo, rk, dl,
                                                          last_n ? 1.0f / nk : 1.0f, last_n && j > 0, st, dLs(i + 1), hLs(i + 1),
                                                          (act_next && last_n && j == nk - 1) ? 0.1f : 1.f);
                        if (frc < 0) return frc;
                        if (frc == 0) { y16 = yn16; continue; }
                    }
                    float* t = buf[3];
                    TapGemmParams p = gemm_params(ctx, R.c1[n]);
                    p.x = y; p.out = t;
                    if (opr) { p.x = nullptr; p.x16 = y16; p.out = nullptr; p.out16 = t16; p.out16_slope = 0.1f; }
                    p.pro_slope = (opr && !r16) ? 1.f : 0.1f;
                    p.nseg = Bc; p.Lin = (int)Lo; p.M = (int)Lo; p.ldx = cout; p.x_seg_stride = Lo * cout;
                    p.dil = dl; p.pad = dl * (rk - 1) / 2; p.ldo = cout; p.o_seg_stride = Lo * cout; p.olimit = p.o_seg_stride;
                    seg_conv(p, i + 1);
                    if (vl) p.algo_macs = rows_of(hLs(i + 1), Lo) * (double)cout * cout * rk;
                    if ((rc = si_launch_tapgemm(ctx, R.c1[n].math, p, st))) return rc;
                    const bool last = (n == d.num_dil - 1);
                    float* ynext = last ? xs : buf[4 + (n & 1)];
                    unsigned short* ynext16 = last ? xs16 : h16[4 + (n & 1)];
                    TapGemmParams q = gemm_params(ctx, R.c2[n]);
                    q.x = t; q.out = ynext; q.res = y;
                    if (opr) {
                        q.x = nullptr; q.x16 = t16;
                        if (r16) { q.res = nullptr; q.res16 = y16; q.out = nullptr; }
                        // the 16-bit copy is wanted by the next c1 of this block, or -- once the MRF mean is complete --
                        // by the next stage's upsampler; conv_post reads fp32
                        const bool want16 = r16 || !last || (j == nk - 1 && i + 1 < d.num_ups);
                        if (want16) { q.out16 = ynext16; q.out16_slope = (r16 && !(act_next && last && j == nk - 1)) ? 1.f : 0.1f; }
                    }
                    q.nseg = Bc; q.Lin = (int)Lo; q.M = (int)Lo; q.ldx = cout; q.x_seg_stride = Lo * cout;
                    q.dil = 1; q.pad = (rk - 1) / 2; q.ldo = cout; q.o_seg_stride = Lo * cout; q.olimit = q.o_seg_stride;
                    q.pro_slope = opr ? 1.f : 0.1f;               // the intermediate is stored activated in the 16-bit modes
                    if (last) { q.alpha = 1.0f / nk; q.accumulate = (j > 0); q.acc16 = (r16 && j > 0); }
                    seg_conv(q, i + 1);
                    if (vl) q.algo_macs = rows_of(hLs(i + 1), Lo) * (double)cout * cout * rk;
                    if ((rc = si_launch_tapgemm(ctx, R.c2[n].math, q, st))) return rc;
                    y = ynext;
                    y16 = ynext16;
                }
            }
            if (!r16 && (rc = si_tap(ctx, stn[i], xs, (long)Bc * Lo * cout, st))) return rc;
            std::swap(x, xs);
            std::swap(x16, xs16);
            Lc = Lo; c = cout;
        }
        // B4: leaky_relu(0.01) -> conv_post -> tanh
        return si_launch_conv_post(ctx, x, wf(ctx, Ly.post_w), wf(ctx, Ly.post_b), Bc, (int)Lc, c, 7, wav_out + (size_t)b0 * Lwav, st,
                                   r16 ? x16 : nullptr, dLs(d.num_ups), hLs(d.num_ups));
    };

    for (int b0 = 0; b0 < B; b0 += Bc_max) {
        const int Bc = std::min(Bc_max, B - b0);
        if (int rc = run(b0, Bc, ext_ws, buf_ws, h16_ws, st)) return rc;
    }
    return SI_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ mel front-end (f-1)
namespace {

// constants of I_ea/dataset/mel_dump.py:11-20
constexpr int FE_NFFT = 1024, FE_HOP = 441, FE_PAD = 312, FE_NMEL = 80, FE_SR = 22050, FE_NBIN = FE_NFFT / 2 + 1;
constexpr double FE_FMIN = 0.0, FE_FMAX = 8000.0;
// The real DFT as two half-size exact-fp32 GEMMs.  With s_k = x[k] + x[N - k] and d_k = x[k] - x[N - k] (k = 1 .. N/2 - 1; s_0 = x[0],
// s_{N/2} = x[N/2]):  Re X[n] = sum_{k=0}^{N/2} s_k cos(2 pi n k / N),  Im X[n] = -sum_{k=1}^{N/2-1} d_k sin(2 pi n k / N) -- the
// frames are written folded ([s_0 .. s_512 | 15 zeros | 0, d_1 .. d_511]) and each half meets a 513 x 528 / 513 x 512 matrix
// instead of the whole frame a 1026 x 1024 one: half the MACs, the same products (one fp32 add per operand pair before them).
constexpr int FE_KC = 528, FE_KS = 512;                // folded frame: cosine operands (513 padded to a multiple of 16) | sine operands
constexpr int FE_FRAME = FE_KC + FE_KS;                // 1040 floats per folded frame
constexpr int FE_IMOFF = 516;                          // spec row = [re(0..512) | 3 pad | im(0..512) | 3 pad]
constexpr int FE_LDSPEC = 1032;                        // spec row stride (16-byte aligned halves)

// Slaney mel scale (librosa.filters.mel defaults htk=False, norm='slaney', as called at mel_dump.py:66)
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

size_t fe_round(size_t b) { return (b + 255) / 256 * 256; }

int ensure_frontend(si_ctx* ctx) {
    if (ctx->fe_dev) return SI_OK;
    const int npad = si_round_up(FE_NBIN, si_pick_bn(FE_NBIN));
    ctx->fe_dft = 0;                                                   // [npad][FE_KC] cosines, then [npad][FE_KS] (minus) sines
    ctx->fe_hann = ctx->fe_dft + fe_round((size_t)npad * FE_FRAME * 4);
    ctx->fe_basis = ctx->fe_hann + fe_round((size_t)FE_NFFT * 4);
    ctx->fe_lo = ctx->fe_basis + fe_round((size_t)FE_NBIN * FE_NMEL * 4);
    ctx->fe_hi = ctx->fe_lo + fe_round((size_t)FE_NMEL * 4);
    const size_t total = ctx->fe_hi + fe_round((size_t)FE_NMEL * 4);
    std::vector<char> host(total, 0);
    float* dft = reinterpret_cast<float*>(host.data() + ctx->fe_dft);
    float* hann = reinterpret_cast<float*>(host.data() + ctx->fe_hann);
    float* basis_t = reinterpret_cast<float*>(host.data() + ctx->fe_basis);
    int32_t* lo = reinterpret_cast<int32_t*>(host.data() + ctx->fe_lo);
    int32_t* hi = reinterpret_cast<int32_t*>(host.data() + ctx->fe_hi);
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < FE_NFFT; ++k) hann[k] = (float)(0.5 - 0.5 * std::cos(two_pi * k / FE_NFFT));   // torch.hann_window (periodic)
    float* dsin = dft + (size_t)npad * FE_KC;
    for (int n = 0; n < FE_NBIN; ++n)
        for (int k = 0; k <= FE_NFFT / 2; ++k) {
            const double a = two_pi * ((long)n * k % FE_NFFT) / FE_NFFT;   // exact argument reduction
            dft[(size_t)n * FE_KC + k] = (float)std::cos(a);               // columns 0 .. 512 (513 .. 527 stay zero)
            if (k >= 1 && k < FE_NFFT / 2) dsin[(size_t)n * FE_KS + k] = (float)-std::sin(a);   // columns 1 .. 511 (d_0 = 0)
        }
    // triangular filters on the Slaney scale, area-normalised
    std::vector<double> mel_f(FE_NMEL + 2);
    const double m0 = hz_to_mel(FE_FMIN), m1 = hz_to_mel(FE_FMAX);
    for (int i = 0; i < FE_NMEL + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (FE_NMEL + 1));
    for (int i = 0; i < FE_NMEL; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = FE_NBIN, last = -1;
        for (int f = 0; f < FE_NBIN; ++f) {
            const double hz = (double)FE_SR / 2.0 * f / (FE_NBIN - 1);
            const double lower = (hz - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - hz) / (mel_f[i + 2] - mel_f[i + 1]);
            const double w = std::max(0.0, std::min(lower, upper)) * enorm;
            const float wf32 = (float)w;
            basis_t[(size_t)f * FE_NMEL + i] = wf32;
            if (wf32 != 0.f) { first = std::min(first, f); last = f; }
        }
        lo[i] = last < 0 ? 0 : first;
        hi[i] = last < 0 ? 0 : last + 1;
    }
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    char* dev = nullptr;
    SI_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), total));
    hipError_t e = hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(dev); return si_fail_hip(ctx, e, "front-end table upload", __FILE__, __LINE__); }
    ctx->fe_dev = dev;
    return SI_OK;
}

size_t mel_ws_bytes(int B, int N22) {
    const long Tm = (N22 + 2 * FE_PAD - FE_NFFT) / FE_HOP + 1;
    if (Tm < 1) return 0;
    return fe_round((size_t)B * 4) + fe_round((size_t)B * Tm * FE_FRAME * 4) + fe_round((size_t)B * Tm * FE_LDSPEC * 4) + 256 +
           fe_round((size_t)2 * B * 4);                               // ragged batches: the length table
}

}  // namespace

extern "C" {

int si_mel_frames(int n22) { return n22 + 2 * FE_PAD < FE_NFFT ? 0 : (n22 + 2 * FE_PAD - FE_NFFT) / FE_HOP + 1; }

int si_mel_workspace_bytes(si_ctx* ctx, int B, int N22, size_t* out) {
    if (!ctx || !out || B <= 0 || si_mel_frames(N22) < 1) return si_fail(ctx, SI_EINVAL, "si_mel_workspace_bytes: bad argument");
    *out = mel_ws_bytes(B, N22);
    return SI_OK;
}

static int mel_run(si_ctx* ctx, const float* wave22, const int32_t* mask_start, const int32_t* mask_end, int normalize, int B,
                   int N22, float* mel_out, void* workspace, size_t workspace_bytes, si_stream_t stream, const int32_t* host_len);

int si_mel_frontend(si_ctx* ctx, const float* wave22, const int32_t* mask_start, const int32_t* mask_end, int normalize, int B,
                    int N22, float* mel_out, void* workspace, size_t workspace_bytes, si_stream_t stream) {
    return mel_run(ctx, wave22, mask_start, mask_end, normalize, B, N22, mel_out, workspace, workspace_bytes, stream, nullptr);
}

int si_mel_frontend_varlen(si_ctx* ctx, const float* wave22, const int32_t* mask_start, const int32_t* mask_end, const int32_t* sample_len,
                           int normalize, int B, int N22, float* mel_out, void* workspace, size_t workspace_bytes, si_stream_t stream) {
    if (!ctx) return SI_EINVAL;
    if (!sample_len) return si_fail(ctx, SI_EINVAL, "si_mel_frontend_varlen: NULL lengths");
    return mel_run(ctx, wave22, mask_start, mask_end, normalize, B, N22, mel_out, workspace, workspace_bytes, stream, sample_len);
}

// host_len (B, HOST) or null: ragged batch -- clip b holds host_len[b] samples of its row of N22 (peak, reflection and frame count are its own)
static int mel_run(si_ctx* ctx, const float* wave22, const int32_t* mask_start, const int32_t* mask_end, int normalize, int B,
                   int N22, float* mel_out, void* workspace, size_t workspace_bytes, si_stream_t stream, const int32_t* host_len) {
    if (!ctx) return SI_EINVAL;
    if (!wave22 || !mel_out || !workspace || B <= 0) return si_fail(ctx, SI_EINVAL, "si_mel_frontend: NULL / empty argument");
    if ((mask_start == nullptr) != (mask_end == nullptr))
        return si_fail(ctx, SI_EINVAL, "si_mel_frontend: mask_start and mask_end must both be given or both be NULL");
    const int Tm = si_mel_frames(N22);
    if (Tm < 1 || N22 <= FE_PAD) return si_fail(ctx, SI_EINVAL, "clip of %d samples is too short for the %d-sample reflect pad / %d-point STFT", N22, FE_PAD, FE_NFFT);
    if (workspace_bytes < mel_ws_bytes(B, N22))
        return si_fail(ctx, SI_ENOMEM, "workspace of %zu bytes < %zu needed for B=%d N22=%d", workspace_bytes, mel_ws_bytes(B, N22), B, N22);
    int rc = ensure_frontend(ctx);
    if (rc) return rc;
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver W{static_cast<char*>(workspace), workspace_bytes};
    float* peak = W.floats((size_t)B);
    float* frames = W.floats((size_t)B * Tm * FE_FRAME);
    float* spec = W.floats((size_t)B * Tm * FE_LDSPEC);
    const int32_t *d_n = nullptr, *d_tm = nullptr;
    if (host_len) {                                                    // table [samples | frames] per clip
        std::vector<int32_t>& tab = ctx->vl_host;
        tab.assign((size_t)2 * B, 0);
        for (int b = 0; b < B; ++b) {
            if (host_len[b] <= FE_PAD || host_len[b] > N22 || si_mel_frames(host_len[b]) < 1)
                return si_fail(ctx, SI_EINVAL, "ragged batch: clip %d holds %d samples: too short for the mel front-end or longer than its row (%d)", b, host_len[b], N22);
            tab[b] = host_len[b];
            tab[(size_t)B + b] = si_mel_frames(host_len[b]);
        }
        int32_t* d_tab = reinterpret_cast<int32_t*>(W.bytes(tab.size() * 4));
        if (!W.ok) return si_fail(ctx, SI_ENOMEM, "internal: mel workspace carve exceeded its own estimate");
        if ((rc = vl_upload(ctx, d_tab, st))) return rc;
        d_n = d_tab; d_tm = d_tab + B;
    }
    if (!W.ok) return si_fail(ctx, SI_ENOMEM, "internal: mel workspace carve exceeded its own estimate");
    const float* hann = reinterpret_cast<const float*>(ctx->fe_dev + ctx->fe_hann);
    if (normalize && (rc = si_launch_wave_peak(ctx, wave22, mask_start, mask_end, B, N22, peak, st, d_n))) return rc;
    if ((rc = si_launch_mel_frames(ctx, wave22, mask_start, mask_end, peak, hann, B, N22, Tm, FE_HOP, FE_PAD, FE_NFFT, FE_KC, normalize, frames, st, d_n, d_tm)))
        return rc;
    if ((rc = si_tap(ctx, "mel_frames", frames, (long)B * Tm * FE_FRAME, st))) return rc;
    // STFT as two exact-fp32 GEMMs on the folded frames: (B*Tm, 528) x (528, 513) -> re, (B*Tm, 512) x (512, 513) -> im
    const int npad = si_round_up(FE_NBIN, si_pick_bn(FE_NBIN));
    for (int half = 0; half < 2; ++half) {
        TapGemmParams p{};
        p.w = ctx->fe_dev + ctx->fe_dft + (half ? (size_t)npad * FE_KC * 4 : 0); p.w_lo = nullptr; p.bias = nullptr; p.res = nullptr;
        p.x = frames + (half ? FE_KC : 0); p.out = spec + (half ? FE_IMOFF : 0);
        p.nseg = 1; p.Lin = B * Tm; p.M = B * Tm; p.ldx = FE_FRAME; p.x_seg_stride = 0;
        p.Cin = half ? FE_KS : FE_KC; p.N = FE_NBIN; p.Npad = npad; p.ntaps = 1; p.stride = 1; p.dil = 1; p.pad = 0; p.groups = 1;
        p.ldo = FE_LDSPEC; p.o_seg_stride = 0; p.ooff = 0; p.olimit = (long)B * Tm * FE_LDSPEC - (half ? FE_IMOFF : 0);
        p.pro_slope = 1.f; p.act = SI_ACT_NONE; p.alpha = 1.f; p.accumulate = 0;
        if ((rc = si_launch_tapgemm(ctx, SI_MATH_F32, p, st))) return rc;
    }
    return si_launch_mel_project(ctx, spec, FE_LDSPEC, FE_NBIN, FE_IMOFF, reinterpret_cast<const float*>(ctx->fe_dev + ctx->fe_basis),
                                 reinterpret_cast<const int32_t*>(ctx->fe_dev + ctx->fe_lo),
                                 reinterpret_cast<const int32_t*>(ctx->fe_dev + ctx->fe_hi), FE_NMEL, B, Tm, mel_out, st, d_tm);
}

}  // extern "C"

extern "C" {

int si_profile_start(si_ctx* ctx, int max_launches) {
    if (!ctx || max_launches <= 0) return si_fail(ctx, SI_EINVAL, "si_profile_start: bad argument");
    SI_HIP_CHECK(hipSetDevice(ctx->device));
    while (ctx->prof_pool.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        SI_HIP_CHECK(hipEventCreate(&e));
        ctx->prof_pool.push_back(e);
    }
    ctx->prof_recs.clear();
    ctx->prof_names.clear();
    ctx->prof_used = 0;
    ctx->prof_open = -1;
    ctx->prof_on = true;
    return SI_OK;
}

int si_profile_filter(si_ctx* ctx, const char* family) {
    if (!ctx) return SI_EINVAL;
    ctx->prof_filter = family ? family : "";
    return SI_OK;
}

int si_profile_stop(si_ctx* ctx, si_profile_entry* out, int capacity, int* count) {
    if (!ctx || !count) return si_fail(ctx, SI_EINVAL, "si_profile_stop: bad argument");
    ctx->prof_on = false;
    std::vector<si_profile_entry> agg(ctx->prof_names.size());
    for (size_t i = 0; i < agg.size(); ++i) {
        memset(&agg[i], 0, sizeof(si_profile_entry));
        snprintf(agg[i].name, sizeof(agg[i].name), "%s", ctx->prof_names[i].c_str());
    }
    for (const auto& r : ctx->prof_recs) {
        SI_HIP_CHECK(hipEventSynchronize(r.b));
        float ms = 0.f;
        SI_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        si_profile_entry& e = agg[r.name];
        e.launches += 1; e.ms += ms; e.flops += r.flops; e.bytes += r.bytes;
    }
    *count = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < capacity && out; ++i) out[i] = agg[i];
    ctx->prof_recs.clear();
    ctx->prof_used = 0;
    return SI_OK;
}

int si_debug_capture(si_ctx* ctx, const char* name, float* dst, long capacity) {
    if (!ctx || !name) return SI_EINVAL;
    if (!dst || capacity <= 0) ctx->dbg_capture.erase(name);
    else ctx->dbg_capture[name] = {dst, capacity};
    return SI_OK;
}

long si_debug_size(si_ctx* ctx, const char* name) {
    if (!ctx || !name) return SI_EINVAL;
    auto it = ctx->dbg_size.find(name);
    if (it == ctx->dbg_size.end()) return si_fail(ctx, SI_EINVAL, "no intermediate named '%s' in the last forward", name);
    return it->second;
}

}  // extern "C"

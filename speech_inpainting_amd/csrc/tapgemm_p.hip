// tapgemm_p.hip -- persistent, cross-tile-pipelined form of the tap-GEMM convolution kernel (gfx950, wave64, MFMA).
//
// Same contraction, LDS images, MFMA use and epilogue arithmetic as tapgemm.hip; what changes is the OUTER structure.
// In-kernel stamps of tapgemm.hip (bf16x3, 8-wave 256x128 tile) showed a wave spending 33 % of its life in an exposed
// prologue (cold first loads) and epilogue: the epilogue's ~16 us per tile is just the tile's output traffic at the
// chip's fair share of HBM bandwidth, because every CU computes at the same time and then every CU stores at the same
// time.  Here each workgroup is persistent: it walks tiles b, b+G, b+2G, ... and the (tile, chunk, tap) iteration
// space is FLATTENED, so
//   * the weight-slab prefetch (two iterations ahead) and the activation-chunk prefetch (one chunk ahead) simply roll
//     over into the next tile: its "prologue" is in flight during the current tile's last iterations;
//   * the residual of the current tile is prefetched into registers before its last MFMA section;
//   * the epilogue only computes and ISSUES its stores -- nothing waits for them; they drain under the next tile's
//     MFMAs, which spreads every CU's HBM traffic over its whole compute time.
// Convolutions only (ntaps >= 2, BK = 32); Linear layers and odd shapes stay on tapgemm.hip.
//
// STATUS: opt-in experiment (SI_TG_PERSIST=1), parity-green but NOT faster on MI355X: bf16x3 256x128w8 15.3 ms/step
// vs 14.7 for the per-tile grid, fp32 leg 69.1 vs 63.8 ms; the 256x64 / 256x32 variants need 288 B of scratch for the
// residual prefetch registers and are 3x slower.  With two workgroups' worth of waves per SIMD the hardware already
// overlaps one tile's epilogue with its neighbour's main loop, which is what this kernel tried to do by hand.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float p_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int MATH> struct PElem { typedef float type; static constexpr int PAD = 4; };
template <> struct PElem<SI_MATH_BF16> { typedef unsigned short type; static constexpr int PAD = 8; };
template <> struct PElem<SI_MATH_BF16X3> { typedef unsigned short type; static constexpr int PAD = 8; };

template <int BM, int NT> struct PMaxA { static constexpr int value = NT == 512 ? 6 : (BM == 128 ? 10 : 12); };

struct PIter {            // one (tile, chunk, tap) iteration; all members wave-uniform
    int tile, c, t;
    int seg, m0, n0;      // decoded tile
};

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, 2) void tapgemm_p_kernel(const TapGemmParams p, const int ntiles) {
    constexpr int BK = 32;
    constexpr int NT = 64 * WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    typedef typename PElem<MATH>::type elem_t;
    constexpr int LD = BK + PElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    constexpr int V4 = BK / 4;
    constexpr int VB = (MATH == SI_MATH_F32) ? BK / 4 : BK / 8;
    constexpr int MAXB = (BN * VB + NT - 1) / NT;
    constexpr int MAXA = PMaxA<BM, NT>::value;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
    const int g = blockIdx.y;

    const int mtiles = (p.M + BM - 1) / BM;
    const int ntn = (p.N + BN - 1) / BN;
    const int ntaps = p.ntaps;
    const int nchunks = p.Cin / BK;
    const int total = nchunks * ntaps;                             // iterations per tile
    const int stride_tiles = gridDim.x;

    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int dil_lo = p.dil < 0 ? (ntaps - 1) * p.dil : 0;
    const int rowsA = (BM - 1) * p.stride + (ntaps - 1) * adil + 1;
    const int nA = rowsA * V4;

    const size_t a_tile = (size_t)PLANES * rowsA * LD;
    constexpr size_t b_tile = (size_t)PLANES * BN * LD;
    elem_t* As = reinterpret_cast<elem_t*>(smem);                 // [PLANES][rowsA][LD]
    elem_t* Bs = As + a_tile;                                     // [2][PLANES][BN][LD]

    const size_t wplane = (size_t)ntaps * p.Npad * p.Cin;
    const float slope = p.pro_slope;

    auto decode = [&](PIter& d) {                                  // N-tiles of one M-tile are adjacent (L2 reuse)
        const int mt = d.tile / ntn;
        d.seg = mt / mtiles;
        d.m0 = (mt - d.seg * mtiles) * BM;
        d.n0 = (d.tile - mt * ntn) * BN;
    };
    // advance to the next iteration of this workgroup's flattened space; returns false (and leaves d) at the end
    auto advance = [&](PIter& d) -> bool {
        if (d.t + 1 < ntaps) { ++d.t; return true; }
        if (d.c + 1 < nchunks) { d.t = 0; ++d.c; return true; }
        if (d.tile + stride_tiles < ntiles) { d.t = 0; d.c = 0; d.tile += stride_tiles; decode(d); return true; }
        return false;
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    f32x4 ra[MAXA];
    f32x4 rb0[PLANES][MAXB], rb1[PLANES][MAXB];
    float rv[TM][TN][16];                                          // residual of the current tile, prefetched

    auto issueA = [&](const PIter& d, int chunk) {
        const float* xs = p.x + (long)d.seg * p.x_seg_stride + (long)g * p.Cin;
        const int base_in = d.m0 * p.stride - p.pad + dil_lo;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int idx = tid + i * NT;
            ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < nA) {
                const int r = idx / V4, j = idx - r * V4;
                const int grow = base_in + r;
                if (grow >= 0 && grow < p.Lin) ra[i] = *reinterpret_cast<const f32x4*>(xs + (long)grow * p.ldx + chunk * BK + 4 * j);
            }
        }
    };
    auto storeA = [&]() {
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int idx = tid + i * NT;
            if (idx < nA) {
                const int r = idx / V4, j = idx - r * V4;
                f32x4 v = ra[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                if constexpr (MATH == SI_MATH_F32) {
                    *reinterpret_cast<f32x4*>(As + r * LD + 4 * j) = v;
                } else {
                    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                    *reinterpret_cast<bf16x4*>(As + r * LD + 4 * j) = hi;
                    if constexpr (MATH == SI_MATH_BF16X3) {
                        const f32x4 rem = v - __builtin_convertvector(hi, f32x4);
                        *reinterpret_cast<bf16x4*>(As + (size_t)rowsA * LD + r * LD + 4 * j) = __builtin_convertvector(rem, bf16x4);
                    }
                }
            }
        }
    };
    auto issueB = [&](f32x4 (&rb)[PLANES][MAXB], const PIter& d) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            const char* wbase = reinterpret_cast<const char*>(pl == 0 ? p.w : p.w_lo) +
                                sizeof(elem_t) * ((size_t)g * wplane + ((size_t)d.t * p.Npad + d.n0) * p.Cin + d.c * BK);
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int idx = tid + i * NT;
                if (BN * VB % NT == 0 || idx < BN * VB) {
                    const int r = idx / VB, j = idx - r * VB;
                    rb[pl][i] = *reinterpret_cast<const f32x4*>(wbase + sizeof(elem_t) * (size_t)r * p.Cin + 16 * j);
                }
            }
        }
    };
    auto storeB = [&](const f32x4 (&rb)[PLANES][MAXB], elem_t* dst) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int idx = tid + i * NT;
                if (BN * VB % NT == 0 || idx < BN * VB) {
                    const int r = idx / VB, j = idx - r * VB;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst + (size_t)pl * BN * LD + r * LD) + 16 * j) = rb[pl][i];
                }
            }
    };

    auto compute = [&](const elem_t* Bc, int tap) {
        const int toff = tap * p.dil - dil_lo;
        if constexpr (MATH == SI_MATH_F32) {
            const float* ap[TM];
            const float* bp[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ap[i] = As + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * (BK / 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) bp[j] = Bc + (wn0 + j * 32 + l31) * LD + half * (BK / 2);
#pragma unroll
            for (int s4 = 0; s4 < BK / 8; ++s4) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap[i] + 4 * s4);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bp[j] + 4 * s4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        } else {
            const unsigned short* ap[TM];
            const unsigned short* bp[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ap[i] = As + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * 8;
#pragma unroll
            for (int j = 0; j < TN; ++j) bp[j] = Bc + (wn0 + j * 32 + l31) * LD + half * 8;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 ah[TM], bh[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const bf16x8*>(ap[i] + 16 * ks);
#pragma unroll
                for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const bf16x8*>(bp[j] + 16 * ks);
                if constexpr (MATH == SI_MATH_BF16X3) {
                    bf16x8 al[TM], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const bf16x8*>(ap[i] + (size_t)rowsA * LD + 16 * ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(bp[j] + (size_t)BN * LD + 16 * ks);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    // ---- per-tile output addressing (buffer descriptors over the segment: the range check masks rows >= M, the negative
    //      offsets of the ConvTranspose phase layout and, with the voffset forced to 2^31, the columns >= N) ----
    const bool has_res = p.res != nullptr;
    const bool acc_out = p.accumulate != 0;
    const bool gelu = p.act == SI_ACT_GELU;
    const int nbytes = (int)p.olimit * 4;
    const int rstep = p.ldo * 4;
    auto tile_rsrc = [&](const float* base, const PIter& d) {
        const float* bp = base + (long)__builtin_amdgcn_readfirstlane(d.seg) * p.o_seg_stride;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bp), 0, nbytes, 0x00020000);
    };
    auto tile_vb = [&](const PIter& d, int i, int j) -> int {
        const int n = d.n0 + wn0 + j * 32 + l31;
        const int col = g * p.N + n + (int)p.ooff;
        return n < p.N ? ((d.m0 + wm0 + i * 32 + 4 * half) * p.ldo + col) * 4 : (int)0x80000000;
    };
    auto prefetch_res = [&](const PIter& d) {
        const __amdgpu_buffer_rsrc_t rrsrc = tile_rsrc(p.res, d);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int vb = tile_vb(d, i, j);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, vb + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0));
            }
    };
    auto epilogue = [&](const PIter& d) {
        const __amdgpu_buffer_rsrc_t orsrc = tile_rsrc(p.out, d);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = d.n0 + wn0 + j * 32 + l31;
                const float bv = (p.bias && n < p.N) ? p.bias[g * p.N + n] : 0.f;
                const int vb = tile_vb(d, i, j);
                float ov[16];
                if (acc_out) {                                     // read-modify-write of the MRF sum: 2 of 18 launches per stage
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ov[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(orsrc, vb + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (gelu) v = p_gelu_erf(v);
                    if (has_res) v += rv[i][j][r];
                    v *= p.alpha;
                    if (acc_out) v += ov[r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, vb + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0);
                }
            }
    };

    // ---- flattened (tile, chunk, tap) loop ----
    PIter cur{(int)blockIdx.x, 0, 0, 0, 0, 0};
    if (cur.tile >= ntiles) return;                                // uniform per workgroup
    decode(cur);
    PIter nx1 = cur, far = cur;                                    // iterations it+1 and it+2 (clamped at the end)
    bool has1 = advance(nx1);
    far = nx1;
    if (has1) advance(far);

    issueA(cur, 0);
    issueB(rb0, cur);
    issueB(rb1, nx1);
    storeA();
    storeB(rb0, Bs);
    __syncthreads();

    int parity = 0;                                                // LDS weight buffer holding the current iteration
    // one iteration: `rissue` receives the slab of it+2, `rland` holds the slab of it+1
    auto step = [&](f32x4 (&rissue)[PLANES][MAXB], const f32x4 (&rland)[PLANES][MAXB]) -> bool {
        issueB(rissue, far);                                       // unconditional (clamped): keeps vmcnt counts exact
        const bool last_of_tile = cur.t == ntaps - 1 && cur.c == nchunks - 1;
        if (cur.t == 0) {                                          // the chunk after this one (maybe the next tile's first)
            PIter nc = cur;
            bool ok = true;
            if (cur.c + 1 < nchunks) nc.c = cur.c + 1;
            else if (cur.tile + stride_tiles < ntiles) { nc.tile = cur.tile + stride_tiles; nc.c = 0; decode(nc); }
            else ok = false;
            if (ok) issueA(nc, nc.c);
        }
        if (last_of_tile && has_res) prefetch_res(cur);
        compute(Bs + (size_t)parity * b_tile, cur.t);
        if (last_of_tile) {
            epilogue(cur);                                         // stores are issued, never waited for
            zero_acc();
        }
        if (!has1) return false;
        const bool new_chunk = nx1.t == 0;
        if (new_chunk) __syncthreads();                            // every wave is done reading the activation tile
        storeB(rland, Bs + (size_t)(parity ^ 1) * b_tile);
        if (new_chunk) storeA();
        __syncthreads();
        parity ^= 1;
        cur = nx1;
        nx1 = far;
        has1 = !(cur.tile == far.tile && cur.c == far.c && cur.t == far.t);
        advance(far);
        return true;
    };
    for (;;) {
        if (!step(rb0, rb1)) break;
        if (!step(rb1, rb0)) break;
    }
}

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N>
static int p_launch_cfg(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    constexpr int BK = 32;
    constexpr int NT = 64 * WARPS_M * WARPS_N;
    typedef typename PElem<MATH>::type elem_t;
    constexpr int LD = BK + PElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const size_t lds = (size_t)PLANES * ((size_t)rowsA + 2 * BN) * LD * sizeof(elem_t);
    if (rowsA * (BK / 4) > PMaxA<BM, NT>::value * NT || lds > 160 * 1024) return 1;
    auto kern = tapgemm_p_kernel<MATH, BM, BN, WARPS_M, WARPS_N>;
    static size_t lds_set = 0;
    if (lds > 64 * 1024 && lds > lds_set) {
        SI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    const int mtiles = (p.M + BM - 1) / BM;
    const int ntiles = p.nseg * mtiles * ((p.N + BN - 1) / BN);
    // resident workgroups: 256 CUs x (2 for 4-wave tiles, 1 for the 8-wave tile), bounded by what LDS admits
    const int per_cu = NT == 512 ? 1 : (int)std::min<size_t>(2, (160 * 1024) / lds);
    const int grid_x = std::min(ntiles, 256 * std::max(per_cu, 1));
    static const char* const math_names[] = {"f32", "bf16", "bf16x3"};
    char name[48];
    snprintf(name, sizeof(name), "tapgemm_p_%s_%dx%d%s", math_names[MATH], BM, BN, NT == 512 ? "w8" : "");
    const double macs = p.algo_macs > 0 ? p.algo_macs : (double)p.nseg * p.M * p.N * p.groups * (double)p.Cin * p.ntaps;
    double bytes = 4.0 * p.nseg * ((double)p.Lin * p.Cin * p.groups + (double)p.M * p.N * p.groups * (1 + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0))) +
                   (double)p.groups * p.ntaps * p.N * p.Cin * (MATH == SI_MATH_F32 ? 4 : (MATH == SI_MATH_BF16 ? 2 : 4));
    si_prof_begin(ctx, name, 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(kern, dim3(grid_x, p.groups), dim3(NT), lds, st, p, ntiles);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

template <int MATH>
static int p_launch_math(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    const int bn = si_pick_bn(p.N);
    if (p.M <= 256) return 1;
    if (bn == 128) return p_launch_cfg<MATH, 256, 128, 4, 2>(ctx, p, st);
    if (bn == 64) return p_launch_cfg<MATH, 256, 64, 4, 1>(ctx, p, st);
    return p_launch_cfg<MATH, 256, 32, 4, 1>(ctx, p, st);
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller falls back to tapgemm.hip).
int si_launch_tapgemm_p(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st) {
    if (p.ntaps < 2 || p.Cin % 32 != 0 || p.groups != 1) return 1;
    switch (math) {
        case SI_MATH_F32: return p_launch_math<SI_MATH_F32>(ctx, p, st);
        case SI_MATH_BF16: return p_launch_math<SI_MATH_BF16>(ctx, p, st);
        case SI_MATH_BF16X3: return p_launch_math<SI_MATH_BF16X3>(ctx, p, st);
    }
    return 1;
}

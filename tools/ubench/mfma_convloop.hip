// Micro-benchmark (diagnostic): the chain kernel's tap loop in isolation -- 12 accumulators, two fragment sets, runtime tap
// count, optional LDS fragment reads -- to find what keeps its MFMA stream at ~60 % of the dense rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ADDR 1: one swizzled address per tap, the row tiles at immediate offsets (round 3); AGPR 1: accumulators pinned to AccVGPRs
template <int READS, int PAIRLOOP, int STAGGER = 0, int PRIO = 0, int NY = 6, int NW = 2, int ADDR = 0, int AGPR = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, const _Float16* src, int rounds, int kt, int dd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    for (int i = tid; i < 100 * 1024 / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(src)[i & 1023];
    __syncthreads();
    const char* As = smem;
    const char* Wc = smem + 56 * 1024;
    int lin0[NY], preW[NW];
    for (int i = 0; i < NY; ++i) lin0[i] = (wave * 16 * NY + 16 * i + r16 + 32) * 64 + (kg << 4);
    for (int j = 0; j < NW; ++j) preW[j] = (16 * j + r16) * 64 + (kg << 4);
    const int c = (kt - 1) / 2;
    f32x4 acc[NY][NW];
    float s = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < NY; ++i) for (int j = 0; j < NW; ++j) acc[i][j] = f32x4{(float)r, 0.f, 1.f, 2.f};
        auto load = [&](f16x8 (&y)[NY], f16x8 (&w)[NW], int tap) {
            if (!READS && tap > 0) return;
            const int soff = (tap - c) * dd * 64;
            if constexpr (ADDR) {
                const int lin = lin0[0] + soff;
                const char* ap = As + (lin ^ ((lin >> 3) & 32));
                const char* wp = Wc + tap * (NW * 1024) + preW[0];
#pragma unroll
                for (int i = 0; i < NY; ++i) y[i] = *reinterpret_cast<const f16x8*>(ap + i * 1024);
#pragma unroll
                for (int j = 0; j < NW; ++j) w[j] = *reinterpret_cast<const f16x8*>(wp + j * 1024);
                return;
            }
#pragma unroll
            for (int i = 0; i < NY; ++i) { const int lin = lin0[i] + soff; y[i] = *reinterpret_cast<const f16x8*>(As + (lin ^ ((lin >> 3) & 32))); }
#pragma unroll
            for (int j = 0; j < NW; ++j) w[j] = *reinterpret_cast<const f16x8*>(Wc + tap * (NW * 1024) + preW[j]);
        };
        auto mma = [&](const f16x8 (&y)[NY], const f16x8 (&w)[NW]) {
#pragma unroll
            for (int i = 0; i < NY; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(w[j]), "v"(y[i]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
                }
        };
        f16x8 ya[NY], wa[NW], yb[NY], wb[NW];
        load(ya, wa, 0);
        if constexpr (STAGGER > 0) { if (wave >= 4) __builtin_amdgcn_s_sleep(STAGGER); }
        if constexpr (PAIRLOOP) {
            for (int tap = 0; tap < kt; tap += 2) {
                load(yb, wb, tap + 1 < kt ? tap + 1 : kt - 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
                if constexpr (PRIO == 2) { if (wave < 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); }
                mma(ya, wa);
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                load(ya, wa, tap + 2 < kt ? tap + 2 : kt - 1);
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < kt) {
                    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
                    if constexpr (PRIO == 2) { if (wave < 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); }
                    mma(yb, wb);
                    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int tap = 0; tap < kt; ++tap) {
                mma(ya, wa);
                __builtin_amdgcn_sched_barrier(0);
                load(ya, wa, tap + 1 < kt ? tap + 1 : kt - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < NY; ++i) for (int j = 0; j < NW; ++j) s += acc[i][j][0] + acc[i][j][3];
        __syncthreads();
    }
    out[blockIdx.x * 512 + tid] = s;
}

int main() {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<_Float16> h(8192);
    unsigned x = 12345;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = (_Float16)(((x >> 8) & 0xffff) / 65536.0f - 0.5f); }
    _Float16* src; float* out;
    (void)hipMalloc(&src, h.size() * 2); (void)hipMalloc(&out, (size_t)ncu * 512 * 4);
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int kt, int dd, int mfma_per_tap = 12) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        const int R = 20000;
        kern<<<ncu, 512, 100 * 1024>>>(out, src, R / 4, kt, dd);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        kern<<<ncu, 512, 100 * 1024>>>(out, src, R, kt, dd);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / R;
        printf("%-44s k = %2d d = %d: %6.3f us per convolution, %6.1f ns per MFMA and SIMD, %5.0f TFLOP/s\n", name, kt, dd, us, us * 1e3 / (kt * 2 * mfma_per_tap), (double)ncu * 8 * kt * mfma_per_tap * 16384.0 / us * 1e-6);
    };
    for (int kt : {3, 7, 11}) {
        run(k<0, 1>, "6 x 2 fragments (chain), no LDS reads after tap 0", kt, 1);
        run(k<1, 1>, "6 x 2 fragments, LDS reads (8 per 12 MFMAs)", kt, 1);
        run(k<1, 1, 0, 1>, "6 x 2, LDS reads, setprio 1 on MFMA blocks", kt, 1);
        run(k<1, 1, 2, 0>, "6 x 2, LDS reads, second half sleeps 128", kt, 1);
        run(k<1, 0>, "6 x 2, LDS reads, one fragment set", kt, 1);
        run(k<1, 1, 0, 0, 6, 2, 1>, "6 x 2, LDS reads, shared address", kt, 1);
        run(k<1, 1, 0, 0, 6, 2, 1, 1>, "6 x 2, LDS reads, shared address, AGPR acc", kt, 1);
        run(k<0, 1, 0, 0, 6, 2, 1, 1>, "6 x 2, no reads, AGPR acc", kt, 1);
    }
    // the 64 x 64 wave tile of the wide kernels / lingemm: 4 + 4 fragment reads per 16 MFMAs (the read addresses move with
    // the step, so the compiler cannot hoist them as it can in mfma_rate.hip's LDS-fed mode)
    for (int kt : {8, 32}) {
        run(k<0, 1, 0, 0, 4, 4>, "4 x 4 fragments, no LDS reads after step 0", kt, 1, 16);
        run(k<1, 1, 0, 0, 4, 4>, "4 x 4 fragments, LDS reads (8 per 16 MFMAs)", kt, 1, 16);
        run(k<1, 1, 0, 1, 4, 4>, "4 x 4, LDS reads, setprio 1 on MFMA blocks", kt, 1, 16);
        run(k<1, 1, 0, 0, 4, 4, 1>, "4 x 4, LDS reads, shared address", kt, 1, 16);
        run(k<1, 1, 0, 0, 4, 4, 1, 1>, "4 x 4, LDS reads, shared address, AGPR acc", kt, 1, 16);
        run(k<0, 1, 0, 0, 4, 4, 1, 1>, "4 x 4, no reads, AGPR acc", kt, 1, 16);
    }
    return 0;
}

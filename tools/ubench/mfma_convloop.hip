// Micro-benchmark (diagnostic): the chain kernel's tap loop in isolation -- 12 accumulators, two fragment sets, runtime tap
// count, optional LDS fragment reads -- to find what keeps its MFMA stream at ~60 % of the dense rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int READS, int PAIRLOOP, int STAGGER = 0, int PRIO = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, const _Float16* src, int rounds, int kt, int dd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    for (int i = tid; i < 100 * 1024 / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(src)[i & 1023];
    __syncthreads();
    const char* As = smem;
    const char* Wc = smem + 56 * 1024;
    int lin0[6], preW[2];
    for (int i = 0; i < 6; ++i) lin0[i] = (wave * 96 + 16 * i + r16 + 32) * 64 + (kg << 4);
    for (int j = 0; j < 2; ++j) preW[j] = (16 * j + r16) * 64 + (kg << 4);
    const int c = (kt - 1) / 2;
    f32x4 acc[6][2];
    float s = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{(float)r, 0.f, 1.f, 2.f};
        auto load = [&](f16x8 (&y)[6], f16x8 (&w)[2], int tap) {
            if (!READS && tap > 0) return;
            const int soff = (tap - c) * dd * 64;
#pragma unroll
            for (int i = 0; i < 6; ++i) { const int lin = lin0[i] + soff; y[i] = *reinterpret_cast<const f16x8*>(As + (lin ^ ((lin >> 3) & 32))); }
#pragma unroll
            for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const f16x8*>(Wc + tap * 2048 + preW[j]);
        };
        auto mma = [&](const f16x8 (&y)[6], const f16x8 (&w)[2]) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
        };
        f16x8 ya[6], wa[2], yb[6], wb[2];
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
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
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
    auto run = [&](auto kern, const char* name, int kt, int dd) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        const int R = 20000;
        kern<<<ncu, 512, 100 * 1024>>>(out, src, R / 4, kt, dd);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        kern<<<ncu, 512, 100 * 1024>>>(out, src, R, kt, dd);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / R;
        printf("%-44s k = %2d d = %d: %6.3f us per convolution, %6.1f ns per MFMA and SIMD, %5.0f TFLOP/s\n", name, kt, dd, us, us * 1e3 / (kt * 24), (double)ncu * 8 * kt * 12 * 16384.0 / us * 1e-6);
    };
    for (int kt : {3, 7, 11}) {
        run(k<0, 1>, "no LDS reads after tap 0", kt, 1);
        run(k<1, 1>, "LDS reads", kt, 1);
        run(k<1, 1, 0, 1>, "LDS reads, setprio 1 on MFMA blocks", kt, 1);
        run(k<1, 1, 0, 2>, "LDS reads, setprio 3 / 1 by wave half", kt, 1);
        run(k<1, 1, 1, 0>, "LDS reads, second half sleeps 64", kt, 1);
        run(k<1, 1, 2, 0>, "LDS reads, second half sleeps 128", kt, 1);
        run(k<1, 1, 3, 0>, "LDS reads, second half sleeps 192", kt, 1);
        run(k<1, 1, 2, 1>, "LDS reads, sleeps 128 + setprio 1", kt, 1);
        run(k<1, 1, 3, 1>, "LDS reads, sleeps 192 + setprio 1", kt, 1);
    }
    return 0;
}

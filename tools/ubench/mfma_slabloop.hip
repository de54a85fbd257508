// Micro-benchmark (diagnostic): the slab loop of the wide fused ResBlock kernel in isolation -- a 64 x 64 wave tile (4 + 4
// fragment reads per 16 MFMAs), KS k-steps per weight slab, one workgroup barrier per slab -- to price the bubble a slab
// boundary leaves in the MFMA stream and what removes it.
//   PRE 0   as the kernel ran until round 3: the first fragments of a slab are read AFTER the barrier that publishes it
//   PRE 1   the first fragments of slab s + 1 are read BEFORE the barrier that ends slab s (the slab must then have been
//           published one barrier earlier: a deeper weight ring)
//   WST 1   every wave also stores 4 x 16 bytes per lane per slab into the other weight buffer (the staging stores)
//   WST 2   the real thing: slab s + 2 is requested from global memory (an L2-resident weight array, 7 x 32 KB) slot by slot
//           behind the MFMA blocks while slab s + 1 leaves its registers for LDS (respair_wide.hip's hook)
//   WST 3   the same traffic as LDS-DMA (buffer_load ... lds, no registers, no ds_write), waited for before the barrier
//   GRP 1   no workgroup barrier per slab: the waves form two groups of four (one wave of each group per SIMD: waves 0-3 own
//           channel half 0, waves 4-7 half 1) that meet at a counter in LDS (one ds_add per wave, then polling) -- the two
//           waves of a SIMD drift apart instead of being re-aligned every slab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KS, int PRE, int WST, int GRP = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, const _Float16* src, int rounds, int nslab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    for (int i = tid; i < 144 * 1024 / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(src)[i & 1023];
    __syncthreads();
    const char* Ys = smem;                                             // 306 rows x 256 bytes
    char* Ws = smem + 80 * 1024;                                       // 2 x 32 KB
    const int wm0 = GRP ? (wave & 3) * 64 : (wave >> 1) * 64, wn0 = GRP ? (wave >> 2) * 64 : (wave & 1) * 64;
    unsigned* const cnt = reinterpret_cast<unsigned*>(smem + 144 * 1024 - 64) + (wave >> 2) * 4;   // one counter per group
    if (tid < 16) reinterpret_cast<unsigned*>(smem + 144 * 1024 - 64)[tid] = 0;
    __syncthreads();
    unsigned epoch = 0;
    const int preW = (wn0 + r16) * 256 + ((((wn0 + r16) & 7) << 5) ^ (kg << 4));
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 1.f, 2.f};
    f16x8 ya[4], wa[4], yb[4], wb[4];
    auto load = [&](f16x8 (&y)[4], f16x8 (&w)[4], int s, int ks) {
        const int r0 = wm0 + r16 + (s % 11) * 3;
        const int preY = r0 * 256 + (((r0 & 7) << 5) ^ (kg << 4));
        const char* yp = Ys + (preY ^ (ks * 64));
        const char* wp = Ws + (s & 1) * 32768 + (preW ^ (ks * 64));
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const f16x8*>(yp + i * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f16x8*>(wp + j * 4096);
    };
    auto mma = [&](const f16x8 (&y)[4], const f16x8 (&w)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
    };
    const f32x4 stv = {1.f, 2.f, 3.f, 4.f};
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(src), 0, 7 * 32768, 0x00020000);
    u32x4 rw[4];
    if constexpr (WST == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (i * 512 + tid) * 16, 0, 0);
    }
    for (int r = 0; r < rounds; ++r) {
        if (PRE) load(ya, wa, 0, 0);
        for (int s = 0; s < nslab; ++s) {
            if (!PRE) load(ya, wa, s, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ks += 2) {
                load(yb, wb, s, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma(ya, wa);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (WST == 2) {
                    constexpr int SPI = 8 / KS > 0 ? 8 / KS : 1;
                    const int soff = __builtin_amdgcn_readfirstlane(((s + 2) % 7) * 32768);
#pragma unroll
                    for (int q = 0; q < SPI; ++q) {
                        const int i = (ks / 2 * SPI + q) & 3;
                        *reinterpret_cast<u32x4*>(Ws + ((s + 1) & 1) * 32768 + (i * 512 + tid) * 16) = rw[i];
                        rw[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (i * 512 + tid) * 16, soff, 0);
                    }
                } else if constexpr (WST == 3) {
                    constexpr int SPI = 8 / KS > 0 ? 8 / KS : 1;
                    const int soff = __builtin_amdgcn_readfirstlane(((s + 1) % 7) * 32768);
#pragma unroll
                    for (int q = 0; q < SPI; ++q) {
                        const int i = (ks / 2 * SPI + q) & 3;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(Ws + ((s + 1) & 1) * 32768 + (i * 512 + wave * 64) * 16), 16, lane * 16 + (i * 512 + wave * 64) * 16, soff, 0, 0);
                    }
                } else if (WST) {
                    constexpr int SPI = 8 / KS > 0 ? 8 / KS : 1;           // stores per loop trip: 4 per slab
#pragma unroll
                    for (int q = 0; q < SPI; ++q)
                        *reinterpret_cast<f32x4*>(Ws + ((s + 1) & 1) * 32768 + (((ks / 2 * SPI + q) & 3) * 512 + tid) * 16) = stv;
                }
                if (ks + 2 < KS) load(ya, wa, s, ks + 2);
                else if (PRE) load(ya, wa, s + 1, 0);                  // across the barrier
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma(yb, wb);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (GRP) {
                epoch += 4;
                __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's LDS reads / stores are done
                if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - epoch)) < 0)
                    __builtin_amdgcn_s_sleep(1);
            } else {
                if constexpr (WST == 3) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
                __syncthreads();
            }
        }
    }
    float sres = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sres += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = sres;
}

int main() {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<_Float16> h(7 * 16384);
    unsigned x = 12345;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = (_Float16)(((x >> 8) & 0xffff) / 65536.0f - 0.5f); }
    _Float16* src; float* out;
    (void)hipMalloc(&src, h.size() * 2); (void)hipMalloc(&out, (size_t)ncu * 512 * 4);
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int ks, int nslab) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        const int R = 2000;
        kern<<<ncu, 512, 144 * 1024>>>(out, src, R / 4, nslab);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        kern<<<ncu, 512, 144 * 1024>>>(out, src, R, nslab);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / R / nslab;
        printf("%-64s %6.3f us per slab, %6.1f ns per MFMA and SIMD, %5.0f TFLOP/s\n", name, us, us * 1e3 / (ks * 16 * 2), (double)ncu * 8 * ks * 16 * 16384.0 / us * 1e-6);
    };
    run(k<4, 0, 0>, "4 k-steps per slab, first reads after the barrier", 4, 14);
    run(k<4, 1, 0>, "4 k-steps per slab, first reads before the barrier", 4, 14);
    run(k<4, 0, 1>, "4 k-steps per slab, reads after, staging stores", 4, 14);
    run(k<4, 1, 1>, "4 k-steps per slab, reads before, staging stores", 4, 14);
    run(k<4, 0, 2>, "4 k-steps per slab, reads after, slabs staged from L2 through registers", 4, 14);
    run(k<4, 1, 2>, "4 k-steps per slab, reads before, slabs staged from L2 through registers", 4, 14);
    run(k<4, 0, 3>, "4 k-steps per slab, reads after, slabs staged from L2 by LDS-DMA", 4, 14);
    run(k<2, 0, 0>, "2 k-steps per slab, first reads after the barrier", 2, 28);
    run(k<2, 1, 0>, "2 k-steps per slab, first reads before the barrier", 2, 28);
    run(k<2, 1, 1>, "2 k-steps per slab, reads before, staging stores", 2, 28);
    run(k<4, 0, 0, 1>, "4 k-steps per slab, reads after, two 4-wave groups at LDS counters", 4, 14);
    run(k<4, 1, 0, 1>, "4 k-steps per slab, reads before, two 4-wave groups at LDS counters", 4, 14);
    run(k<2, 0, 0, 1>, "2 k-steps per slab, reads after, two 4-wave groups at LDS counters", 2, 28);
    run(k<8, 0, 0>, "8 k-steps per slab, first reads after the barrier", 8, 7);
    run(k<8, 1, 0>, "8 k-steps per slab, first reads before the barrier", 8, 7);
    return 0;
}

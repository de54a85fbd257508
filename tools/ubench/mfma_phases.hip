// Micro-benchmark (diagnostic, not part of the library): does a dense MFMA burst run slower when it alternates with
// VALU-only phases (the shape of the fused ResBlock kernels: convolution | barrier | epilogue | barrier), and does it
// matter whether all CUs burst in step?  hipcc --offload-arch=gfx950 -O3 -o mfma_phases mfma_phases.hip && ./mfma_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// per round: `nm` blocks of 12 MFMAs (16x16x32 f16, 12 accumulators), barrier, `nv` rounds of 48 VALU FMAs, barrier.
// stagger: odd workgroups start with the VALU phase (half a period out of step with the even ones).
__global__ __launch_bounds__(512, 1) void k(float* out, const _Float16* src, int rounds, int nm, int nv, int stagger) {
    const int lane = threadIdx.x & 63;
    f16x8 a[6], b[2];
    for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const f16x8*>(src + 8 * lane + 512 * i);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f16x8*>(src + 8 * lane + 512 * (6 + i));
    f32x4 acc[6][2] = {};
    float v[48];
    for (int i = 0; i < 48; ++i) v[i] = (float)src[lane + i];
    const bool odd = stagger && (blockIdx.x & 1);
    for (int r = 0; r < rounds; ++r) {
        for (int ph = 0; ph < 2; ++ph) {
            const bool mf = (ph == 0) != odd;
            if (mf) {
                for (int m = 0; m < nm; ++m) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
                }
            } else {
                for (int m = 0; m < nv; ++m) {
#pragma unroll
                    for (int i = 0; i < 48; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
    for (int i = 0; i < 48; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<_Float16> h(8192);
    unsigned x = 12345;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = (_Float16)(((x >> 8) & 0xffff) / 65536.0f - 0.5f); }
    _Float16* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, (size_t)ncu * 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int rounds, int nm, int nv, int stagger) {
        k<<<ncu, 512>>>(out, src, rounds / 4, nm, nv, stagger);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<ncu, 512>>>(out, src, rounds, nm, nv, stagger);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3 / rounds;                                    // us per round
    };
    const int R = 4000;
    // nm = 86 blocks of 12 MFMAs per wave ~ one k = 7 convolution of the chain kernel (2 waves per SIMD)
    for (int nm : {3, 7, 11, 43, 344}) {
        const int nv = nm / 10 + 1;
        const double t_m = run(R, nm, 0, 0), t_v = run(R, 0, nv, 0), t_b = run(R, nm, nv, 0), t_s = run(R, nm, nv, 1);
        const double flops = (double)ncu * 8 * nm * 12 * 16384.0;
        printf("bursts of %4d x 12 MFMAs per wave: MFMA only %7.2f us/round (%6.0f TFLOP/s); VALU only %6.2f us; alternating %7.2f us -> MFMA phases at %6.0f TFLOP/s; "
               "odd workgroups half a period out of step %7.2f us -> %6.0f TFLOP/s\n",
               nm, t_m, flops / t_m * 1e-6, t_v, t_b, flops / (t_b - t_v) * 1e-6, t_s, flops / (t_s - t_v) * 1e-6);
    }
    return 0;
}

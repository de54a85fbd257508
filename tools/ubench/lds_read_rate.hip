// Micro-benchmark (diagnostic): what a CU's LDS delivers to ds_read_b128 streams of the 16x16x32 operand pattern
// (conflict-free, 64-byte rows, swizzled), 4 and 8 waves per CU, 8 or 16 independent reads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(512, 1) void k(unsigned* out, const unsigned* src, int iters, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 1024 / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(src)[i & 1023];
    __syncthreads();
    if (wave >= nwaves) return;
    const int r16 = lane & 15, kg = lane >> 4;
    u32x4 acc[INFLIGHT];
    for (int i = 0; i < INFLIGHT; ++i) acc[i] = u32x4{0u, 0u, 0u, 0u};
    int row = wave * 96 + r16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            const int r = (row + 16 * (i % 6) + 3 * it) & 1023;                   // a moving first row, as the taps of a convolution
            const int lin = r * 64 + (kg << 4);
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (lin ^ ((lin >> 3) & 0x30)));
            acc[i] ^= v;
        }
    }
    u32x4 s = acc[0];
    for (int i = 1; i < INFLIGHT; ++i) s ^= acc[i];
    out[blockIdx.x * 512 + tid] = s[0] ^ s[1] ^ s[2] ^ s[3];
}

int main() {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<unsigned> h(4096, 0x3c003c00u);
    unsigned *src, *out;
    (void)hipMalloc(&src, h.size() * 4); (void)hipMalloc(&out, (size_t)ncu * 512 * 4);
    (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](auto kern, int inflight, int nwaves) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        const int iters = 20000;
        kern<<<ncu, 512, 128 * 1024>>>(out, src, iters / 4, nwaves);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        kern<<<ncu, 512, 128 * 1024>>>(out, src, iters, nwaves);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes_per_cu = (double)nwaves * iters * inflight * 1024.0;
        printf("%d waves per CU, %2d ds_read_b128 per trip: %6.1f B/ns per CU (%5.1f TB/s on %d CUs)\n", nwaves, inflight,
               bytes_per_cu / (ms * 1e6), bytes_per_cu * ncu / (ms * 1e-3) * 1e-12, ncu);
    };
    for (int nw : {4, 8}) { run(k<8>, 8, nw); run(k<16>, 16, nw); }
    return 0;
}

// Calibration of the rocprofv3 HBM-traffic counters (FETCH_SIZE / WRITE_SIZE) for the access widths this library uses
// (MI355X_MICROARCH.md, section HBM: FETCH_SIZE reads exactly half of a 16-byte-per-lane coalesced stream on gfx950; other
// widths are "uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel streams a KNOWN
// number of bytes (512 MiB read, 512 MiB written: larger than the 256 MiB Infinity Cache) with one access width:
//     copy16 / copy8 / copy4 : global loads + stores of 16 / 8 / 4 bytes per lane, lane-contiguous
//     rows8                  : 8-byte accesses where 8 lanes cover a 64-byte row segment and rows are 256 bytes apart
//                              (the narrow-tile epilogue shape of round 1)
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and again with `--pmc WRITE_SIZE`; tools/collect_calibration.py
// divides the counters by the known byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__global__ __launch_bounds__(256) void copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
// lane l of a wave: row (l >> 3), 8-byte piece (l & 7) of a 64-byte segment; consecutive rows 256 bytes apart
__global__ __launch_bounds__(256) void rows8_kernel(const u32x2* __restrict__ src, u32x2* __restrict__ dst, long nrows) {
    const int piece = threadIdx.x & 7;
    for (long r = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; r < nrows; r += ((long)gridDim.x * 256) >> 3)
        for (int seg = 0; seg < 4; ++seg) dst[r * 32 + seg * 8 + piece] = src[r * 32 + seg * 8 + piece];
}

int main() {
    const long bytes = 512L << 20;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        copy_kernel<u32x4><<<grid, 256>>>((const u32x4*)a, (u32x4*)b, bytes / 16);
        copy_kernel<u32x2><<<grid, 256>>>((const u32x2*)a, (u32x2*)b, bytes / 8);
        copy_kernel<unsigned><<<grid, 256>>>((const unsigned*)a, (unsigned*)b, bytes / 4);
        rows8_kernel<<<grid, 256>>>((const u32x2*)a, (u32x2*)b, bytes / 256);
    }
    (void)hipDeviceSynchronize();
    printf("bytes_per_launch %ld\n", bytes);
    return 0;
}

// Measurement helper for bench.py (NOT part of the product library): a plain streaming-read kernel that
// finds the box's achievable HBM read ceiling with the same access shape as the scans (16 B per lane,
// nontemporal, non-persistent, one 16 KiB tile per wave, workgroups dispatched in address order), so
// roofline fractions can be quoted against both the 8 TB/s vendor peak and what this chip delivers.
// Built by `make -C quantization_amd/csrc probe` into tools/probe/libqamd_probe.so.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stream_read_kernel(const u32x4 *__restrict__ p, uint64_t n16,
                                                         uint32_t *__restrict__ scratch) {
    const uint64_t wave = ((uint64_t)blockIdx.x * 512 + threadIdx.x) >> 6;
    const uint64_t base = wave * 1024 + (threadIdx.x & 63);  // 1024 x 16 B per wave
    uint32_t acc = 0;
    if (base + 15 * 64 < n16) {
        u32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + base + j * 64);
#pragma unroll
        for (int j = 0; j < 16; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9E3779B9u) scratch[blockIdx.x & 16383] = acc;  // keeps the loads live, ~never stores
}
}  // namespace

// Sums `bytes` of device memory (on the current device) with 16-byte loads; writes at most one u32 per
// workgroup to scratch (>= 64 KiB).  Returns 0 on success, the hipError_t otherwise.
extern "C" __attribute__((visibility("default"))) int qamd_probe_stream_read(const void *dev_ptr, uint64_t bytes,
                                                                             void *scratch, void *stream) {
    if (!dev_ptr || !scratch) return (int)hipErrorInvalidValue;
    const uint64_t n16 = bytes / 16;
    const unsigned grid = (unsigned)((n16 / 1024 + 7) / 8);  // 8 waves per workgroup, one tile per wave
    hipLaunchKernelGGL(stream_read_kernel, dim3(grid ? grid : 1), dim3(512), 0, static_cast<hipStream_t>(stream),
                       static_cast<const u32x4 *>(dev_ptr), n16, static_cast<uint32_t *>(scratch));
    return (int)hipGetLastError();
}

// Which XCD does workgroup b of a 256-workgroup launch (512 threads, 144 KiB of LDS: one workgroup per CU) run on, and on which CU?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_map_probe tools/experiments/xcd_map_probe.hip && /tmp/xcd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void who(uint32_t *out) {
    extern __shared__ uint8_t lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;
        uint32_t xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
    }
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(who), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(who, dim3(256), dim3(512), 144 * 1024, 0, d);
        std::vector<uint32_t> h(512);
        hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
        int same = 0;
        for (int b = 8; b < 256; b++) same += (h[2 * b] & 0xF) == (h[2 * (b - 8)] & 0xF);
        printf("launch %d: blocks b and b - 8 on the same XCD: %d of 248;  XCD of blocks 0..15:", rep, same);
        for (int b = 0; b < 16; b++) printf(" %u", h[2 * b] & 0xF);
        printf("\n  CU ids (hw_id bits 8-11 cu, 13-15 se) of blocks 0, 8, 16, 24: %x %x %x %x\n", h[1], h[17], h[33], h[49]);
    }
    return 0;
}

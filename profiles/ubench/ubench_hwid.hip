// ubench_hwid.hip -- where do the waves of a workgroup land?  Prints (xcc, se, cu, simd, wave slot) per wave for a few blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out)
{
    const int wave = threadIdx.x >> 6;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 0] = hwid;
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = xcc;
    }
    // keep the block alive a little so that blocks overlap
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main()
{
    for (int W : {1, 2, 4}) {
        const int waves = 4 * W, blocks = 1024 / W;
        unsigned* d;
        (void)hipMalloc(&d, blocks * waves * 8);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, d);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(blocks * waves * 2);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("W=%d (block of %d waves):\n", W, waves);
        for (int b : {0, 1, 8, 9}) {
            printf("  block %d:", b);
            for (int w = 0; w < waves; ++w) {
                const unsigned id = h[(b * waves + w) * 2], x = h[(b * waves + w) * 2 + 1];
                printf(" [w%d xcc%u se%u cu%u simd%u slot%u]", w, x & 0xf, (id >> 13) & 7, (id >> 8) & 0xf, (id >> 4) & 3, id & 0xf);
            }
            printf("\n");
        }
        // histogram: blocks per (xcc,se,cu)
        (void)hipFree(d);
    }
    return 0;
}

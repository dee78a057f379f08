// which XCD does workgroup b land on?  (s_getreg HW_REG_XCC_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 15);
}
int main() {
    const int nb = 4096;
    int* d; hipMalloc(&d, nb * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d);
    std::vector<int> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    int mism = 0;
    for (int b = 0; b < nb; ++b) if (h[b] != (b & 7)) ++mism;
    printf("first 32:"); for (int b = 0; b < 32; ++b) printf(" %d", h[b]); printf("\nmismatch vs b&7: %d of %d\n", mism, nb);
    int cnt[16] = {0}; for (int b = 0; b < nb; ++b) cnt[h[b]]++;
    for (int i = 0; i < 16; ++i) if (cnt[i]) printf("xcc %d: %d\n", i, cnt[i]);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("CUs %d\n", p.multiProcessorCount);
    return 0;
}

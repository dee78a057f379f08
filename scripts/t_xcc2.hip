// XCD and start time of every workgroup of a long-running 4096 x 256 launch with 16 KB of LDS (the SpMV kernels' shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256) k(int* xcc, long long* t0, long long* t1, int spin) {
    __shared__ double lds[2048];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    const long long a = wall_clock64();
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) { lds[(threadIdx.x + i) & 2047] = acc; __syncthreads(); acc += lds[(threadIdx.x * 7 + i) & 2047]; __syncthreads(); }
    const long long b = wall_clock64();
    if (threadIdx.x == 0) { xcc[blockIdx.x] = (int)(x & 15) + (acc == 12345.678 ? 100 : 0); t0[blockIdx.x] = a; t1[blockIdx.x] = b; }
}
int main() {
    const int nb = 4096;
    int* d; long long *a, *b;
    hipMalloc(&d, nb * 4); hipMalloc(&a, nb * 8); hipMalloc(&b, nb * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, a, b, 2000);
    std::vector<int> h(nb); std::vector<long long> ha(nb), hb(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), a, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, nb * 8, hipMemcpyDeviceToHost);
    int mism = 0; for (int i = 0; i < nb; ++i) if (h[i] != (i & 7)) ++mism;
    long long m = *std::min_element(ha.begin(), ha.end());
    printf("mismatch vs b&7: %d of %d ; clock rate 100MHz ticks\n", mism, nb);
    for (int i : {0, 8, 64, 128, 256, 512, 1024, 1032, 2040, 2048, 2056, 3000, 4088}) printf("wg %4d xcc %d start %lld end %lld\n", i, h[i], ha[i] - m, hb[i] - m);
    // how many wgs started within the first 10 us (1000 ticks)?
    int early = 0; for (int i = 0; i < nb; ++i) if (ha[i] - m < 1000) ++early;
    printf("started within 10 us: %d\n", early);
    std::vector<long long> s(ha); std::sort(s.begin(), s.end());
    printf("start time of the 1024th / 2048th / 3072th / last wg: %lld %lld %lld %lld\n", s[1023] - m, s[2047] - m, s[3071] - m, s[4095] - m);
    return 0;
}

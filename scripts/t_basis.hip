#include "../polystokes_amd/csrc/ps_common.hpp"
#include <cstdio>
__global__ void k(const int* axes, double* out) {
    const int t = threadIdx.x;
    double v[26];
    ps::basisRow(0.1, 0.2, 0.3, axes[blockIdx.x], v);
    if (t == 0) for (int i = 0; i < 26; ++i) out[blockIdx.x * 26 + i] = v[i];
}
__global__ void k2(int axis, double* out) {
    __shared__ double sa[4][26];
    double a[26];
    bool use = threadIdx.x < 2;
    if (use) ps::basisRow(0.1, 0.2, 0.3, axis, a);
    if (!use) { for (int n = 0; n < 26; ++n) a[n] = 0.; }
    for (int n = 0; n < 26; ++n) sa[threadIdx.x][n] = a[n];
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < 26; ++i) out[i] = sa[1][i];
}
int main() {
    int h[3] = {0, 1, 2}; int* d; double* o; double ho[78];
    hipMalloc(&d, 12); hipMalloc(&o, 78 * 8); hipMemcpy(d, h, 12, hipMemcpyHostToDevice);
    k<<<3, 64>>>(d, o); hipMemcpy(ho, o, 78 * 8, hipMemcpyDeviceToHost);
    for (int a = 0; a < 3; ++a) { printf("axis %d:", a); for (int i = 0; i < 26; ++i) printf(" %g", ho[a * 26 + i]); printf("\n"); }
    k2<<<1, 4>>>(2, o); hipMemcpy(ho, o, 26 * 8, hipMemcpyDeviceToHost);
    printf("k2 axis2:"); for (int i = 0; i < 26; ++i) printf(" %g", ho[i]); printf("\n");
}

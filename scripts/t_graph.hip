// launch-bound loop: 225 tiny dependent kernels, stream launches vs one hipGraph launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(double* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * 1.0000001 + 1e-9; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int n = 114144; double* a; CK(hipMalloc(&a, n * 8)); CK(hipMemset(a, 0, n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int K = 225, reps = 40;
    auto run_stream = [&]() { for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, s, a, n); };
    run_stream(); CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) run_stream();
    CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("stream: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (K * reps));
    hipGraph_t g; hipGraphExec_t ge;
    auto c0 = std::chrono::high_resolution_clock::now();
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    run_stream();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    auto c1 = std::chrono::high_resolution_clock::now();
    printf("capture+instantiate: %.1f us\n", std::chrono::duration<double, std::micro>(c1 - c0).count());
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t1 = std::chrono::high_resolution_clock::now();
    printf("graph: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (K * reps));
    return 0;
}

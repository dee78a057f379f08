// cost of a grid-wide barrier between phases of a persistent kernel (the alternative to one launch per phase):
// every workgroup dirties 16 KB, then release fence + atomic arrive + poll + acquire fence.  Spin loops are capped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned* counter, double* buf, int rounds, int* bailed) {
    const int nb = gridDim.x;
    double* mine = buf + (size_t)blockIdx.x * 2048;
    for (int r = 0; r < rounds; ++r) {
        // phase work: write 16 KB, read a neighbour's 16 KB of the previous round
        const double* other = buf + (size_t)((blockIdx.x + 37) % nb) * 2048;
        double acc = 0;
        for (int i = threadIdx.x; i < 2048; i += 256) acc += other[i];
        for (int i = threadIdx.x; i < 2048; i += 256) mine[i] = acc + r;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * (unsigned)nb;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 20000000L) { *bailed = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
// XCD-hierarchical form (MI355X_MICROARCH.md, price table row "barrier-xcd"): workgroup b runs on XCD b & 7; per-XCD arrival counter, the
// last arriver of an XCD (its leader for this round) adds to the top counter and waits for all 8 XCDs, then publishes the round in the
// XCD's generation word; everybody else polls that word.  Counters are monotonic (no reset), every word on a 128-byte line of its own.
__global__ void __launch_bounds__(256) kx(unsigned* ctl, double* buf, int rounds, int* bailed) {
    const int nb = gridDim.x, x = blockIdx.x & 7, perX = nb >> 3;
    unsigned* cnt = ctl + 32 * x; unsigned* gen = ctl + 32 * (8 + x); unsigned* top = ctl + 32 * 16;
    double* mine = buf + (size_t)blockIdx.x * 2048;
    for (int r = 0; r < rounds; ++r) {
        const double* other = buf + (size_t)((blockIdx.x + 37) % nb) * 2048;
        double acc = 0;
        for (int i = threadIdx.x; i < 2048; i += 256) acc += other[i];
        for (int i = threadIdx.x; i < 2048; i += 256) mine[i] = acc + r;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            if (old == (unsigned)(r + 1) * (unsigned)perX - 1u) {          // the XCD's last arriver
                __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * (unsigned)(r + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000L) { *bailed = 1; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(gen, (unsigned)(r + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000L) { *bailed = 1; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
    }
}
__global__ void tiny(double* buf) { if (threadIdx.x == 0 && blockIdx.x == 0) buf[0] += 1.; }
int main() {
    unsigned* c; double* buf; int* bailed;
    hipMalloc(&c, 4); hipMalloc(&buf, (size_t)2048 * 2048 * 8); hipMalloc(&bailed, 4);
    hipMemset(buf, 0, (size_t)2048 * 2048 * 8); hipMemset(bailed, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nb : {256, 512, 1024, 2048}) {
        const int rounds = 500;
        hipMemset(c, 0, 4);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, c, buf, 10, bailed); hipMemset(c, 0, 4);   // warm up
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, c, buf, rounds, bailed);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int hb; hipMemcpy(&hb, bailed, 4, hipMemcpyDeviceToHost);
        printf("grid %4d x 256: %.2f us per phase (16 KB written + 16 KB read per workgroup, barrier included)%s\n", nb, ms * 1e3 / rounds, hb ? "  [BAILED]" : "");
    }
    {   // the XCD-hierarchical barrier on the same phases
        unsigned* ctl; hipMalloc(&ctl, 32 * 17 * 4);
        for (int nb : {256, 512, 1024, 2048}) {
            const int rounds = 500;
            hipMemset(ctl, 0, 32 * 17 * 4);
            hipLaunchKernelGGL(kx, dim3(nb), dim3(256), 0, 0, ctl, buf, 10, bailed); hipDeviceSynchronize();
            hipMemset(ctl, 0, 32 * 17 * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kx, dim3(nb), dim3(256), 0, 0, ctl, buf, rounds, bailed);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hb; hipMemcpy(&hb, bailed, 4, hipMemcpyDeviceToHost);
            printf("XCD-hierarchical, grid %4d x 256: %.2f us per phase (same phase body)%s\n", nb, ms * 1e3 / rounds, hb ? "  [BAILED]" : "");
        }
        // the phase body alone (no barrier): what the barrier adds is the difference
        for (int nb : {256, 512, 1024, 2048}) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kx, dim3(nb), dim3(256), 0, 0, ctl, buf, 0, bailed);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("empty launch of that grid %4d: %.2f us each\n", nb, ms * 1e3 / 100);
        }
    }
    // the same phases as separate launches
    hipEventRecord(e0, 0);
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, buf);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("500 dependent tiny launches: %.2f us each\n", ms * 1e3 / 500);
    return 0;
}

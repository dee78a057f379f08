// How does the CU's vector-memory path (TA -> TCP) price a 64-lane gather of 8-byte elements by the way its lanes share
// cache lines?  (S and St both run at ~0.85 entries / cycle / CU whatever the row shape; the PMC pass of
// profiles/r03_spmv_issue.md shows ~0.9 TCP tag accesses per gathered entry at ~80 % TCP occupancy.)
// Each wave issues UNR independent buffer_load_dwordx2 (or x4) per iteration; G lanes share one 128-B line:
//   pattern 0: consecutive lanes share (lanes g*G .. g*G+G-1), distinct 8-B elements of the line
//   pattern 1: the sharing lanes are strided (lane % (64/G) is the group)
//   pattern 2: consecutive lanes share AND read the same 8 bytes
//   pattern 3: consecutive lanes share, elements in reverse order
// table: `lines` 128-B lines (32 KB fits L1 .. 1 MB L2 .. 64 MB memory-side cache)
// build: hipcc -O3 --offload-arch=gfx950 scripts/t_gather.hip -o scripts/_bin/t_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int G, int PAT, int W16>
__global__ void __launch_bounds__(256) k_gather(const double* __restrict__ tab, unsigned lineMask, int iters, double* __restrict__ out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tab), 0, (int)((lineMask + 1u) * 128u), 0x00020000);
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int UNR = 8;
    const unsigned grp = PAT == 1 ? lane % (64 / G) : lane / G;
    const unsigned sub = PAT == 1 ? lane / (64 / G) : lane % G;
    unsigned elem = PAT == 2 ? 0u : (PAT == 3 ? (G - 1 - sub) : sub);
    if (W16) elem = (elem * 2) & 15; else elem &= 15;
    double acc = 0.;
    for (int it = 0; it < iters; ++it) {
        double v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned line = hash32((wave * 977u + it) * 64u * UNR + u * 64u + grp) & lineMask;
            const unsigned off = line * 128u + elem * 8u;
            if (W16) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
                v[u] = __builtin_bit_cast(double, u32x2{q.x, q.y}) + __builtin_bit_cast(double, u32x2{q.z, q.w});
            } else {
                v[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    if (acc == 123.456) out[0] = acc;
}

template <int G, int PAT, int W16>
void run(const double* tab, unsigned lines, double* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k_gather<G, PAT, W16>), dim3(grid), dim3(256), 0, 0, tab, lines - 1, iters, out);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_gather<G, PAT, W16>), dim3(grid), dim3(256), 0, 0, tab, lines - 1, iters, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)grid * 4 * iters * 8;                 // wave-level gather instructions
    const double perCU = instr / 256.;
    const double cyc = ms * 1e-3 * 2.4e9 / perCU;                     // at a nominal 2.4 GHz
    printf("{\"G\": %d, \"pattern\": %d, \"bytes_per_lane\": %d, \"table_KB\": %u, \"ms\": %.4f, \"cycles_per_wave_gather_per_CU\": %.2f, \"lane_elems_per_cycle_per_CU\": %.2f}\n",
           G, PAT, W16 ? 16 : 8, lines / 8, ms, cyc, 64. / cyc);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    const unsigned maxLines = 1u << 19;   // 64 MB
    double* tab; double* out;
    CHECK(hipMalloc(&tab, (size_t)maxLines * 128)); CHECK(hipMalloc(&out, 64));
    std::vector<double> h((size_t)maxLines * 16, 1.0);
    CHECK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const unsigned sizes[3] = {128u, 8192u, 1u << 19};   // 16 KB (L1), 1 MB (L2), 64 MB
    for (unsigned lines : sizes) {
        run<1, 0, 0>(tab, lines, out, iters);
        run<2, 0, 0>(tab, lines, out, iters);
        run<4, 0, 0>(tab, lines, out, iters);
        run<8, 0, 0>(tab, lines, out, iters);
        run<16, 0, 0>(tab, lines, out, iters);
        run<2, 1, 0>(tab, lines, out, iters);
        run<4, 1, 0>(tab, lines, out, iters);
        run<16, 1, 0>(tab, lines, out, iters);
        run<4, 2, 0>(tab, lines, out, iters);
        run<16, 2, 0>(tab, lines, out, iters);
        run<4, 3, 0>(tab, lines, out, iters);
        run<1, 0, 1>(tab, lines, out, iters);
        run<2, 0, 1>(tab, lines, out, iters);
        run<4, 0, 1>(tab, lines, out, iters);
        run<8, 0, 1>(tab, lines, out, iters);
    }
    return 0;
}

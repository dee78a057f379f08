// Micro-benchmark (r04, for the design note in DESIGN.md section 9): what is a CU worth when the gathers of a (lattice block, plane) item come
// from an LDS image staged ONCE by 16-byte LDS-DMA (__builtin_amdgcn_global_load_lds, asynchronous, double-buffered) instead of
// from memory?  Synthetic, St-shaped: an item = 1792 rows (seven kinds x 256), G gathers per row out of six 17 x 17 segments of a
// 1792-double window (14 KB); consecutive lanes gather consecutive values with a break every 16 lanes; items overlap by 256 doubles.
//   A: buffer-free global gathers (8 B per lane), one 64-row unit per wave step, two units in flight (as k_spmv_St_ell2)
//   B: the window staged by glds into one of two LDS buffers while the previous item is computed; gathers = ds_read_b64
// Both write one double per row.  usage: t_glds_gather [items] [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int WN = 1792, ROWS = 1792, STEP = 1536, BS = 256;
__device__ inline int slot(int row, int g) {
    const int kind = row >> 8, q = row & 255, j = q >> 4, i = q & 15;
    const int sg = (kind + g) % 6;
    return sg * 289 + (j + ((g >> 1) & 1)) * 17 + i + (g & 1);
}
template <int G>
__global__ void __launch_bounds__(BS) k_global(const double* __restrict__ x, double* __restrict__ y, int items) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const double* w = x + (size_t)item * STEP;
        for (int u = wv; u < ROWS / 64; u += 8) {              // two units in flight per wave
            const int ra = u * 64 + lane, rb = (u + 4) * 64 + lane;
            const bool hb = u + 4 < ROWS / 64;
            double va[G], vb[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { va[g] = w[slot(ra, g)]; vb[g] = hb ? w[slot(rb, g)] : 0.; }
            double sa = 0., sb = 0.;
#pragma unroll
            for (int g = 0; g < G; ++g) { sa += va[g]; sb += vb[g]; }
            y[(size_t)item * ROWS + ra] = sa;
            if (hb) y[(size_t)item * ROWS + rb] = sb;
        }
    }
}
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
template <int G>
__global__ void __launch_bounds__(BS) k_lds(const double* __restrict__ x, double* __restrict__ y, int items) {
    __shared__ __attribute__((aligned(16))) double win[2][WN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto stage = [&](int item, int buf) {                       // 14 wave instructions of 1 KB, dealt to the four waves
        const double* w = x + (size_t)item * STEP;
        for (int k = wv; k < WN / 128; k += 4)
            __builtin_amdgcn_global_load_lds((glb_ptr)(w + k * 128 + lane * 2), (lds_ptr)(&win[buf][k * 128]), 16, 0, 0);
    };
    int item = blockIdx.x, buf = 0;
    if (item < items) stage(item, 0);
    for (; item < items; item += gridDim.x, buf ^= 1) {
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's share of the window (and its stores)
        __builtin_amdgcn_s_barrier();                          // everybody's share; and the other buffer is no longer read
        const int nxt = item + gridDim.x;
        if (nxt < items) stage(nxt, buf ^ 1);
        const double* w = win[buf];
        for (int u = wv; u < ROWS / 64; u += 8) {
            const int ra = u * 64 + lane, rb = (u + 4) * 64 + lane;
            const bool hb = u + 4 < ROWS / 64;
            double sa = 0., sb = 0.;
#pragma unroll
            for (int g = 0; g < G; ++g) { sa += w[slot(ra, g)]; if (hb) sb += w[slot(rb, g)]; }
            y[(size_t)item * ROWS + ra] = sa;
            if (hb) y[(size_t)item * ROWS + rb] = sb;
        }
    }
}
template <int G>
int run(int items) {
    const size_t nx = (size_t)items * STEP + WN, ny = (size_t)items * ROWS;
    double *x, *ya, *yb;
    CK(hipMalloc(&x, nx * 8)); CK(hipMalloc(&ya, ny * 8)); CK(hipMalloc(&yb, ny * 8));
    std::vector<double> hx(nx);
    for (size_t i = 0; i < nx; ++i) hx[i] = (double)((i * 2654435761ull) % 1000003ull) * 1e-3;
    CK(hipMemcpy(x, hx.data(), nx * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {768, 1280, 1536, 2048}) {
        float ma = 0, mb = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_global<G>, dim3(grid), dim3(BS), 0, 0, x, ya, items); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ma, e0, e1));
            CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_lds<G>, dim3(grid), dim3(BS), 0, 0, x, yb, items); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&mb, e0, e1));
        }
        std::vector<double> ha(4096), hb(4096);
        CK(hipMemcpy(ha.data(), ya + ny / 2, 4096 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), yb + ny / 2, 4096 * 8, hipMemcpyDeviceToHost));
        bool same = true; for (int i = 0; i < 4096; ++i) same = same && ha[i] == hb[i];
        const double rows = (double)items * ROWS;
        printf("G %d grid %4d  rows %.1f M  global gathers %.3f ms (%.2f ns/krow)   LDS image %.3f ms (%.2f ns/krow)   ratio %.2f   results %s\n", G, grid, rows / 1e6,
               ma / 10, ma / 10 * 1e6 / (rows / 1e3), mb / 10, mb / 10 * 1e6 / (rows / 1e3), ma / mb, same ? "equal" : "DIFFER");
    }
    CK(hipFree(x)); CK(hipFree(ya)); CK(hipFree(yb));
    return 0;
}
int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 25000;      // 44.8 M rows: the DOFs of the 256^3 cavity
    const int G = argc > 2 ? atoi(argv[2]) : 4;
    if (G == 6) return run<6>(items);
    return run<4>(items);
}

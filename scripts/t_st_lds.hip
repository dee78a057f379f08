// Prototype (r05): the St product of the PCG step (y = -S^T t - 1/2 uInv p;  r -= alpha y;  partials of r.r and
// r.z) with EVERY gather served from an LDS image of t that a workgroup stages once per 16^3 lattice block by LDS-DMA:
//   image = [zero pair][the block's active face rows: one contiguous range of t][its tile's skin rows: one contiguous range]
//           [halo: the 16-byte pairs of t that the block's DOF rows touch outside those two ranges, by per-lane address]
// The stream holds 16-bit image positions instead of windowed columns; blocks with identical streams share one (scripts/st_lds_proto.py
// builds all of it with numpy from the library's CSR — setup in HIP only if this kernel is worth it).
// Built as a shared object; driven from Python (ctypes, device pointers of torch tensors).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
constexpr int TB = 512;                       // 8 waves per workgroup, two workgroups per CU (LDS: <= 80 KB each)
struct BlockDesc { int d0, nUnits, unitBase, colBase, codeBase, a0, lenA, s0, lenS, haloBase, nPairs, pad; };
__device__ inline __amdgpu_buffer_rsrc_t bufRsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
struct Regs { unsigned c0, c1, c2, c3, v0, v1; };
__device__ inline Regs loadUnit(__amdgpu_buffer_rsrc_t rCol, __amdgpu_buffer_rsrc_t rCode, int W, int colByte, int codeByte, unsigned lane) {
    const int cb = colByte + (int)lane * 2 * W;
    Regs r{0u, 0u, 0u, 0u, 0u, 0u};
    if (W == 8) { const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rCol, cb, 0, 0); const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rCode, codeByte + (int)lane * 8, 0, 0); r = Regs{q.x, q.y, q.z, q.w, v.x, v.y}; }
    else if (W == 6) { const u32x3 q = __builtin_amdgcn_raw_buffer_load_b96(rCol, cb, 0, 0); const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rCode, codeByte + (int)lane * 8, 0, 0); r = Regs{q.x, q.y, q.z, 0u, v.x, v.y}; }
    else if (W == 4) { const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rCol, cb, 0, 0); const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rCode, codeByte + (int)lane * 4, 0, 0); r = Regs{q.x, q.y, 0u, 0u, v, 0u}; }
    else if (W == 2) { const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(rCol, cb, 0, 0); const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rCode, codeByte + (int)lane * 4, 0, 0); r = Regs{q, 0u, 0u, 0u, v, 0u}; }
    return r;
}
template <int W>
__device__ inline double rowSum(const Regs& r, const double* img, double scale) {
    const unsigned cw[4] = {r.c0, r.c1, r.c2, r.c3};
    const unsigned vw[2] = {r.v0, r.v1};
    double xv[8];
#pragma unroll
    for (int k = 0; k < W; ++k) xv[k] = img[(cw[k >> 1] >> (16 * (k & 1))) & 0xffffu];
    double s = 0.;
#pragma unroll
    for (int k = 0; k < W; ++k) s += (double)((int)(vw[k >> 2] << (24 - 8 * (k & 3))) >> 24) * scale * xv[k];
    return s;
}
__device__ inline double rowSumW(int W, const Regs& r, const double* img, double scale) {
    if (W == 8) return rowSum<8>(r, img, scale);
    if (W == 6) return rowSum<6>(r, img, scale);
    if (W == 4) return rowSum<4>(r, img, scale);
    if (W == 2) return rowSum<2>(r, img, scale);
    return 0.;
}
__device__ inline double waveReduceSum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// STAGE: 0 = LDS-DMA (asynchronous), 1 = through registers (global_load + ds_write)
template <int STAGE>
__global__ void __launch_bounds__(TB) k_st_lds(const BlockDesc* __restrict__ blocks, int nBlocks, const int4* __restrict__ units, const uint16_t* __restrict__ ecol,
                                               const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes, const int32_t* __restrict__ haloPairs,
                                               const double* __restrict__ t, int tLen, const double* __restrict__ p, const uint8_t* __restrict__ uCode,
                                               const double* __restrict__ uDict, double* __restrict__ r, const float* __restrict__ dinvF, int rows, double scale,
                                               double alpha, double* __restrict__ rPart, int imgCap) {
    extern __shared__ __attribute__((aligned(16))) double img[];
    __shared__ double dict[256];
    __shared__ double wsum[2][TB / 64];
    if (threadIdx.x < 256) dict[threadIdx.x] = uDict[threadIdx.x];
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rP = bufRsrc(p, (size_t)rows * 8), rUc = bufRsrc(uCode, (size_t)rows),
                                 rR = bufRsrc(r, (size_t)rows * 8), rD = bufRsrc(dinvF, (size_t)rows * 4);
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double dacc = 0., dacc2 = 0.;
    for (int b = blockIdx.x; b < nBlocks; b += gridDim.x) {
        const BlockDesc bd = blocks[b];
        // ---- stage the image: pieces of 64 lanes x 16 bytes = 128 doubles, dealt to the eight waves
        const int nA = bd.lenA >> 7, nS = bd.lenS >> 7, nH = (bd.nPairs + 63) >> 6;     // lenA, lenS: multiples of 128 doubles (padded by the host)
        if (threadIdx.x < 2) img[threadIdx.x] = 0.;
        for (int k = wv; k < nA + nS + nH; k += TB / 64) {
            const double* src;
            int dst;
            if (k < nA) { src = t + bd.a0 + k * 128 + lane * 2; dst = 2 + k * 128; }
            else if (k < nA + nS) { src = t + bd.s0 + (k - nA) * 128 + lane * 2; dst = 2 + bd.lenA + (k - nA) * 128; }
            else {
                const int q = (k - nA - nS) * 64 + (int)lane;
                const int pr = q < bd.nPairs ? haloPairs[bd.haloBase + q] : 0;
                src = t + 2 * (size_t)pr; dst = 2 + bd.lenA + bd.lenS + (k - nA - nS) * 128;
            }
            if (STAGE == 0) __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(img + dst), 16, 0, 0);
            else { const double2 v = *reinterpret_cast<const double2*>(src); *reinterpret_cast<double2*>(img + dst + lane * 2) = v; }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
        __syncthreads();
        // ---- the block's units: wave w takes units w, w + 8, ... two in flight
        for (int u = wv; u < bd.nUnits; u += 2 * (TB / 64)) {
            const int ub = u + TB / 64;
            const int4 da = units[bd.unitBase + u];
            const int4 db = ub < bd.nUnits ? units[bd.unitBase + ub] : make_int4(0, 0, 0, 0);
            const int Wa = da.w >> 8, Wb = db.w >> 8, na = da.w & 255, nb = db.w & 255;
            const Regs sa = loadUnit(rCol, rCode, Wa, bd.colBase + da.x, bd.codeBase + da.y, lane), sb = loadUnit(rCol, rCode, Wb, bd.colBase + db.x, bd.codeBase + db.y, lane);
            const bool liveA = (int)lane < na, liveB = (int)lane < nb;
            const unsigned rowA = liveA ? (unsigned)(bd.d0 + da.z) + lane : 0x1fffffffu, rowB = liveB ? (unsigned)(bd.d0 + db.z) + lane : 0x1fffffffu;
            const double eA = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rP, (int)(rowA * 8u), 0, 2));
            const double eB = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rP, (int)(rowB * 8u), 0, 2));
            const int ucA = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)rowA, 0, 2), ucB = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)rowB, 0, 2);
            const double crA = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rR, (int)(rowA * 8u), 0, 2));
            const double crB = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rR, (int)(rowB * 8u), 0, 2));
            const float fdA = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)(rowA * 4u), 0, 2));
            const float fdB = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)(rowB * 4u), 0, 2));
            const double a = rowSumW(Wa, sa, img, scale), bsum = rowSumW(Wb, sb, img, scale);
            double yA = -a; yA -= 0.5 * dict[ucA] * eA;
            double yB = -bsum; yB -= 0.5 * dict[ucB] * eB;
            const double rvA = liveA ? crA - alpha * yA : 0., rvB = liveB ? crB - alpha * yB : 0.;
            dacc += rvA * rvA; dacc += rvB * rvB;
            dacc2 += rvA * ((double)fdA * rvA); dacc2 += rvB * ((double)fdB * rvB);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, rvA), rR, (int)(rowA * 8u), 0, 2);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, rvB), rR, (int)(rowB * 8u), 0, 2);
        }
        __syncthreads();                                       // the image is reused
    }
    const double s0 = waveReduceSum(dacc), s1 = waveReduceSum(dacc2);
    if (lane == 0) { wsum[0][wv] = s0; wsum[1][wv] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0., c = 0.;
        for (int i = 0; i < TB / 64; ++i) { a += wsum[0][i]; c += wsum[1][i]; }
        rPart[blockIdx.x] = a; rPart[gridDim.x + blockIdx.x] = c;
    }
}
extern "C" int st_lds_launch(int stage, int grid, int ldsBytes, const void* blocks, int nBlocks, const void* units, const void* ecol, const void* ecode, unsigned colBytes,
                             unsigned codeBytes, const void* haloPairs, const void* t, int tLen, const void* p, const void* uCode, const void* uDict, void* r,
                             const void* dinvF, int rows, double scale, double alpha, void* rPart, int reps, float* msOut) {
    auto k = stage == 0 ? k_st_lds<0> : k_st_lds<1>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(k, dim3(grid), dim3(TB), ldsBytes, 0, (const BlockDesc*)blocks, nBlocks, (const int4*)units, (const uint16_t*)ecol, (const int8_t*)ecode, colBytes, codeBytes,
                           (const int32_t*)haloPairs, (const double*)t, tLen, (const double*)p, (const uint8_t*)uCode, (const double*)uDict, (double*)r, (const float*)dinvF, rows,
                           scale, alpha, (double*)rPart, ldsBytes / 8);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2;
    hipEventElapsedTime(msOut, e0, e1);
    *msOut /= reps;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

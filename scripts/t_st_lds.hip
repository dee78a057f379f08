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
// STAGE: 0 = LDS-DMA into ONE image per workgroup (512 threads, two workgroups per CU take turns), 1 = the same through registers (global_load +
// ds_write), 2 = LDS-DMA, TWO images per workgroup of 1024 threads (one per CU): the next item's image is in flight while this one is computed
// MODE 0: the St product of the PCG step (image of t; rows = DOFs).  MODE 1: the S product, out = (row < nA ? dt dict[mcCode] : 1) (S x)[row], partials of
// sum over the active rows of s (s sc) (image of x = p; rows = face rows; arguments reused: tLen = nA, uCode = mcCode, r = out, alpha = dt)
// NU: 64-row units a wave has in flight
template <int STAGE, int MODE, int NU, int NT>
__global__ void __launch_bounds__(NT) k_st_lds(const BlockDesc* __restrict__ blocks, int nBlocks, const int4* __restrict__ units, const uint16_t* __restrict__ ecol,
                                               const int8_t* __restrict__ ecode, unsigned colBytes, unsigned codeBytes, const int32_t* __restrict__ haloPairs,
                                               const double* __restrict__ t, int tLen, const double* __restrict__ p, const uint8_t* __restrict__ uCode,
                                               const double* __restrict__ uDict, double* __restrict__ r, const float* __restrict__ dinvF, int rows, double scale,
                                               double alpha, double* __restrict__ rPart, int imgCap) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double dict[256];
    __shared__ double wsum[2][NT / 64];
    constexpr int NW = NT / 64;
    if (threadIdx.x < 256) dict[threadIdx.x] = uDict[threadIdx.x];
    const __amdgpu_buffer_rsrc_t rCol = bufRsrc(ecol, colBytes), rCode = bufRsrc(ecode, codeBytes), rP = bufRsrc(p, (size_t)rows * 8), rUc = bufRsrc(uCode, (size_t)rows),
                                 rR = bufRsrc(r, (size_t)rows * 8), rD = bufRsrc(dinvF, (size_t)rows * 4), rMc = bufRsrc(uCode, (size_t)tLen);
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double dacc = 0., dacc2 = 0.;
    auto stage = [&](const BlockDesc& bd, double* img) {       // pieces of 64 lanes x 16 bytes = 128 doubles, dealt to the waves
        const int nA = bd.lenA >> 7, nS = bd.lenS >> 7, nH = (bd.nPairs + 63) >> 6;     // lenA, lenS: multiples of 128 doubles (padded by the host)
        for (int k = wv; k < nA + nS + nH; k += NW) {
            const double* src;
            int dst;
            if (k < nA) { src = t + bd.a0 + k * 128 + lane * 2; dst = 2 + k * 128; }
            else if (k < nA + nS) { src = t + bd.s0 + (k - nA) * 128 + lane * 2; dst = 2 + bd.lenA + (k - nA) * 128; }
            else {
                const int q = (k - nA - nS) * 64 + (int)lane;
                const int pr = q < bd.nPairs ? haloPairs[bd.haloBase + q] : 0;
                src = t + 2 * (size_t)pr; dst = 2 + bd.lenA + bd.lenS + (k - nA - nS) * 128;
            }
            if (STAGE != 1) __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(img + dst), 16, 0, 0);
            else { const double2 v = *reinterpret_cast<const double2*>(src); *reinterpret_cast<double2*>(img + dst + lane * 2) = v; }
        }
    };
    if (threadIdx.x < 2) { lds[threadIdx.x] = 0.; if (STAGE == 2) lds[imgCap + threadIdx.x] = 0.; }
    int b = blockIdx.x, buf = 0;
    BlockDesc bd{}, nbd{};
    if (b < nBlocks) { bd = blocks[b]; if (STAGE == 2) stage(bd, lds); }
    for (; b < nBlocks; b += gridDim.x, buf ^= (STAGE == 2 ? 1 : 0)) {
        double* img = lds + buf * imgCap;
        if (STAGE != 2) stage(bd, img);
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's share of the image
        __syncthreads();                                       // everybody's; and (STAGE 2) nobody reads the other image any more
        const int nb = b + gridDim.x;
        if (nb < nBlocks) { nbd = blocks[nb]; if (STAGE == 2) stage(nbd, lds + (buf ^ 1) * imgCap); }
        // ---- the item's units: wave w takes units w, w + NW, ... NU of them in flight
        for (int u = wv; u < bd.nUnits; u += NU * NW) {
            int W[NU], nr[NU];
            unsigned row[NU];
            Regs sg[NU];
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const int uq = u + q * NW;
                const int4 d = uq < bd.nUnits ? units[bd.unitBase + uq] : make_int4(0, 0, 0, 0);
                W[q] = d.w >> 8; nr[q] = d.w & 255;
                sg[q] = loadUnit(rCol, rCode, W[q], bd.colBase + d.x, bd.codeBase + d.y, lane);
                row[q] = (int)lane < nr[q] ? (unsigned)(bd.d0 + d.z) + lane : 0x1fffffffu;
            }
            if (MODE == 0) {
                double e[NU], cr[NU];
                int uc[NU];
                float fd[NU];
#pragma unroll
                for (int q = 0; q < NU; ++q) {
                    e[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rP, (int)(row[q] * 8u), 0, 2));
                    uc[q] = (int)__builtin_amdgcn_raw_buffer_load_b8(rUc, (int)row[q], 0, 2);
                    cr[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rR, (int)(row[q] * 8u), 0, 2));
                    fd[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)(row[q] * 4u), 0, 2));
                }
#pragma unroll
                for (int q = 0; q < NU; ++q) {
                    const double a = rowSumW(W[q], sg[q], img, scale);
                    double y = -a; y -= 0.5 * dict[uc[q]] * e[q];
                    const double rv = (int)lane < nr[q] ? cr[q] - alpha * y : 0.;
                    dacc += rv * rv;
                    dacc2 += rv * ((double)fd[q] * rv);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, rv), rR, (int)(row[q] * 8u), 0, 2);
                }
            } else {
                int mc[NU];
#pragma unroll
                for (int q = 0; q < NU; ++q) mc[q] = (int)__builtin_amdgcn_raw_buffer_load_b8(rMc, (int)row[q], 0, 2);   // tLen = nA: past it 0 without an access
#pragma unroll
                for (int q = 0; q < NU; ++q) {
                    const double a = rowSumW(W[q], sg[q], img, scale);
                    const double sc = (int)row[q] < tLen ? alpha * dict[mc[q]] : 1.;
                    dacc += (int)row[q] < tLen ? a * (a * sc) : 0.;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a * sc), rR, (int)(row[q] * 8u), 0, 2);
                }
            }
        }
        if (STAGE != 2) __syncthreads();                       // the image is reused
        bd = nbd;
    }
    const double s0 = waveReduceSum(dacc), s1 = waveReduceSum(dacc2);
    if (lane == 0) { wsum[0][wv] = s0; wsum[1][wv] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0., c = 0.;
        for (int i = 0; i < NW; ++i) { a += wsum[0][i]; c += wsum[1][i]; }
        rPart[blockIdx.x] = a; rPart[gridDim.x + blockIdx.x] = c;
    }
}
extern "C" int st_lds_launch(int variant, int grid, int ldsBytes, const void* blocks, int nBlocks, const void* units, const void* ecol, const void* ecode, unsigned colBytes,
                             unsigned codeBytes, const void* haloPairs, const void* t, int tLen, const void* p, const void* uCode, const void* uDict, void* r,
                             const void* dinvF, int rows, double scale, double alpha, void* rPart, int reps, float* msOut) {
    // variant = stage + 4 * mode + 8 * (units in flight == 4)
    const int stage = variant & 3, mode = (variant >> 2) & 1, nu4 = (variant >> 3) & 1;
    typedef void (*K)(const BlockDesc*, int, const int4*, const uint16_t*, const int8_t*, unsigned, unsigned, const int32_t*, const double*, int, const double*, const uint8_t*,
                      const double*, double*, const float*, int, double, double, double*, int);
    K k = nullptr;
    int nt = stage == 2 ? 1024 : 512;
#define PICK(S_, M_, N_, T_) if (stage == S_ && mode == M_ && nu4 == (N_ == 4)) k = k_st_lds<S_, M_, N_, T_>;
    PICK(0, 0, 2, 512) PICK(1, 0, 2, 512) PICK(2, 0, 2, 1024) PICK(0, 0, 4, 512) PICK(1, 0, 4, 512) PICK(2, 0, 4, 1024)
    PICK(0, 1, 2, 512) PICK(1, 1, 2, 512) PICK(2, 1, 2, 1024) PICK(0, 1, 4, 512) PICK(1, 1, 4, 512) PICK(2, 1, 4, 1024)
#undef PICK
    if (!k) return -4;
    const int imgCap = ldsBytes / 8;
    const int dyn = stage == 2 ? 2 * ldsBytes : ldsBytes;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(k, dim3(grid), dim3(nt), dyn, 0, (const BlockDesc*)blocks, nBlocks, (const int4*)units, (const uint16_t*)ecol, (const int8_t*)ecode, colBytes, codeBytes,
                           (const int32_t*)haloPairs, (const double*)t, tLen, (const double*)p, (const uint8_t*)uCode, (const double*)uDict, (double*)r, (const float*)dinvF, rows,
                           scale, alpha, (double*)rPart, imgCap);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2;
    hipEventElapsedTime(msOut, e0, e1);
    *msOut /= reps;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

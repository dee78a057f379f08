// Per-tile dense blocks of the reduced (26-DOF divergence-free quadratic) model:
//   centre of mass            Solver.cpp:328-372, 1274-1324
//   least-squares fit N,rhs   Solver.cpp:374-417, 1330-1399      (c_fit = fullPivLu(N).solve(rhs))
//   reduced mass  Mr          Solver.cpp:419-441, 1405-1482
//   reduced viscosity K       Solver.cpp:468-490, 1484-1694      (interior stresses only)
//   B = Mr/dt + 2K, B^-1, rhs_r = Mr c_fit     AssembleBlocks.cpp:147-244, 356-367
//
// All of N, Mr, K are sums of rank-1 outer products a_f (x) b_f over the faces of a region: one work
// item = <=FB_CHUNK positions of the region's face bounding box; a 256-thread block stages 128 face
// vectors at a time in LDS and accumulates the 26x26 block; partial blocks are then summed per region
// in a fixed order (deterministic, no atomics).
#include "ps_context.hpp"

using namespace ps;

namespace {

constexpr int BS = 256;
constexpr int FBATCH = 128;
constexpr int OUTW = PS_RD * PS_RD + PS_RD;   // 676 block entries + 26 rhs entries

enum { MODE_MASS = 0, MODE_LSQ = 1, MODE_VISC = 2 };

struct TileArgs {
    Grid g;
    double dx, rho;
    const int32_t* lab[7];
    const int32_t* reg[7];
    const float* vel[3];
    const float* visc;
    int viscUniform; float viscValue;   // a constant field: its samples without loads (bit-identical: ps_context::upload)
    const double* COM;
    int3 off;                       // global index of the local cell (0, 0, 0) (ps_kernels_tiles.hpp: rowOffset)
    const int32_t* bbox;
    const int32_t* itemRegion;
    const int32_t* itemAxis;
    const int32_t* itemStart;
};

__device__ inline int labAt(const TileArgs& A, int s, const int3 d, int i, int j, int k) {
    return oob3(d, i, j, k) ? PS_UNASSIGNED : A.lab[s][lin3(d, i, j, k)];
}
__device__ inline int regAt(const TileArgs& A, int s, const int3 d, int i, int j, int k) {
    return oob3(d, i, j, k) ? PS_UNASSIGNED : A.reg[s][lin3(d, i, j, k)];
}

__device__ inline float viscSample(const TileArgs& A, float px, float py, float pz) {
    if (A.viscUniform) return A.viscValue;
    // same restatement of SIM_RawField::getValue as ps_grid.hip::sampleCenterField
    const int n[3] = {A.g.nx, A.g.ny, A.g.nz};
    const float p[3] = {px, py, pz};
    int i0[3], i1[3];
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = p[a] - 0.5f;
        if (u < 0.f) u = 0.f;
        if (u > (float)(n[a] - 1)) u = (float)(n[a] - 1);
        int b = (int)u;
        if (b >= n[a] - 1) { b = n[a] - 1; i0[a] = b; i1[a] = b; t[a] = 0.f; }
        else { i0[a] = b; i1[a] = b + 1; t[a] = u - (float)b; }
    }
    const int64_t sy = A.g.nx, sz = (int64_t)A.g.nx * A.g.ny;
    auto at = [&](int i, int j, int k) { return A.visc[i + j * sy + k * sz]; };
    auto L = [](float a, float b, float tt) { return a + (b - a) * tt; };
    const float c00 = L(at(i0[0], i0[1], i0[2]), at(i1[0], i0[1], i0[2]), t[0]);
    const float c10 = L(at(i0[0], i1[1], i0[2]), at(i1[0], i1[1], i0[2]), t[0]);
    const float c01 = L(at(i0[0], i0[1], i1[2]), at(i1[0], i0[1], i1[2]), t[0]);
    const float c11 = L(at(i0[0], i1[1], i1[2]), at(i1[0], i1[1], i1[2]), t[0]);
    return L(L(c00, c10, t[1]), L(c01, c11, t[1]), t[2]);
}

__device__ inline void faceOffset(const TileArgs& A, int axis, int i, int j, int k, int region, double* o) {
    double p[3] = {(double)(i + A.off.x), (double)(j + A.off.y), (double)(k + A.off.z)};   // global indices: ps_kernels_tiles.hpp, rowOffset
    p[axis] -= 0.5;
#pragma unroll
    for (int q = 0; q < 3; ++q) { p[q] *= A.dx; p[q] -= A.COM[(int64_t)region * 3 + q]; }
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

// g = sum_i contribution_i * C(adjacent face_i): the row-vector side of the viscosity outer products
// of one reduced face (Solver.cpp:1538-1683).
// part: 0 = every term; 1 = the cell terms and the first of the two edge axes, 2 = the second edge axis — the MFMA kernel splits the
// up to 20 basis rows of a face over its two half-blocks (the sum of the two parts differs from part 0 by rounding only).
// FA (the face's axis) is a template parameter: after unrolling, the axis of every adjacent face is a compile-time constant and its
// basis row is added entry by entry where it has entries (ps_common.hpp: basisAccum) — the sums and their order are those of
// basisRow + 26 multiply-adds per adjacent face, a third of the instructions (7.8 -> 5.5 ms of the 256^3 setup).  Folding the rows through the 30
// moments of the face offsets was measured too (5.3 ms) and rejected: it reorders sums that cancel heavily, and on the stiff spheres the
// velocities at tol 1e-8 then sit 1.3e-4 from the oracle's instead of 4e-6 (tests/test_gpu_parity.py: the tolerance ladder).
template <int FA>
__device__ void viscosityRowT(const TileArgs& A, int i, int j, int k, double* gv, int part) {
#pragma unroll
    for (int n = 0; n < PS_RD; ++n) gv[n] = 0.;
    const int3 cd = A.g.dims(0);
    const int3 fd = A.g.dims(1 + FA);
    const double dx2 = A.dx * A.dx;
    // cell-centred stresses
    if (part != 2) {
#pragma unroll
        for (int divDir = 0; divDir < 2; ++divDir) {
            int3 c = make_int3(i, j, k);
            addc(c, FA, divDir - 1);
            if (!isReducedL(labAt(A, 0, cd, c.x, c.y, c.z))) continue;
            if (comp(c, FA) < 0 || comp(c, FA) >= comp(fd, FA)) continue;
            const double divSign = divDir == 0 ? -1. : 1.;
            const double visc = (double)viscSample(A, (float)c.x + 0.5f, (float)c.y + 0.5f, (float)c.z + 0.5f);
#pragma unroll
            for (int gradDir = 0; gradDir < 2; ++gradDir) {
                int3 af = c;
                addc(af, FA, gradDir);
                const double gradSign = gradDir == 0 ? -1. : 1.;
                const double contribution = -1. * divSign * gradSign * visc / dx2;
                const int adj = regAt(A, 1 + FA, fd, af.x, af.y, af.z);
                if (adj < 0) continue;
                double o[3];
                faceOffset(A, FA, af.x, af.y, af.z, adj, o);
                basisAccum<FA>(o[0], o[1], o[2], contribution, gv);
            }
        }
    }
    // edge-centred stresses (pure REDUCED edges only)
    int nth = 0;
#pragma unroll
    for (int edgeAxis = 0; edgeAxis < 3; ++edgeAxis) {
        if (edgeAxis == FA) continue;
        ++nth;                                         // 1: the first edge axis of this face, 2: the second
        if (part != 0 && part != nth) continue;
        const int3 ed = A.g.dims(4 + edgeAxis);
#pragma unroll
        for (int divDir = 0; divDir < 2; ++divDir) {
            const double divSign = divDir == 0 ? -1. : 1.;
            int3 e = make_int3(i, j, k);
            addc(e, 3 - FA - edgeAxis, divDir);
            if (labAt(A, 4 + edgeAxis, ed, e.x, e.y, e.z) != PS_REDUCED) continue;
            const float ox = edgeAxis == 0 ? 0.5f : 0.f, oy = edgeAxis == 1 ? 0.5f : 0.f, oz = edgeAxis == 2 ? 0.5f : 0.f;
            const float visc = viscSample(A, (float)e.x + ox, (float)e.y + oy, (float)e.z + oz);
#pragma unroll
            for (int gradAxis = 0; gradAxis < 3; ++gradAxis) {
                if (gradAxis == edgeAxis) continue;
                const int adjFaceAxis = 3 - gradAxis - edgeAxis;
                const int3 ad = A.g.dims(1 + adjFaceAxis);
#pragma unroll
                for (int gradDir = 0; gradDir < 2; ++gradDir) {
                    int3 af = e;
                    addc(af, gradAxis, gradDir - 1);
                    const double gradSign = gradDir == 0 ? -1. : 1.;
                    const double contribution = -0.5 * divSign * gradSign * visc / dx2;
                    const int adj = regAt(A, 1 + adjFaceAxis, ad, af.x, af.y, af.z);
                    if (adj < 0) continue;
                    double o[3];
                    faceOffset(A, adjFaceAxis, af.x, af.y, af.z, adj, o);
                    if (adjFaceAxis == 0) basisAccum<0>(o[0], o[1], o[2], contribution, gv);
                    else if (adjFaceAxis == 1) basisAccum<1>(o[0], o[1], o[2], contribution, gv);
                    else basisAccum<2>(o[0], o[1], o[2], contribution, gv);
                }
            }
        }
    }
}
__device__ void viscosityRow(const TileArgs& A, int faceAxis, int i, int j, int k, double* gv, int part = 0) {
    if (faceAxis == 0) viscosityRowT<0>(A, i, j, k, gv, part);
    else if (faceAxis == 1) viscosityRowT<1>(A, i, j, k, gv, part);
    else viscosityRowT<2>(A, i, j, k, gv, part);
}

template <int MODE>
__global__ void __launch_bounds__(BS) k_region_outer(TileArgs A, double* __restrict__ partial) {
    __shared__ double sa[FBATCH][PS_RD];
    __shared__ double sb[MODE == MODE_VISC ? FBATCH : 1][PS_RD];
    __shared__ double su[FBATCH];
    const int item = blockIdx.x;
    const int r = A.itemRegion[item], axis = A.itemAxis[item], start = A.itemStart[item];
    const int bx0 = A.bbox[r * 6 + 0], by0 = A.bbox[r * 6 + 1], bz0 = A.bbox[r * 6 + 2];
    int ex = A.bbox[r * 6 + 3] - bx0 + 1, ey = A.bbox[r * 6 + 4] - by0 + 1, ez = A.bbox[r * 6 + 5] - bz0 + 1;
    if (axis == 0) ex++; else if (axis == 1) ey++; else ez++;
    const int total = ex * ey * ez;
    const int end = min(start + FB_CHUNK, total);
    const int3 fd = A.g.dims(1 + axis), cd = A.g.dims(0);

    double acc[3] = {0., 0., 0.};
    double accR = 0.;
    const int t = threadIdx.x;
    int em[3], en[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { const int e = t + q * BS; em[q] = e / PS_RD; en[q] = e % PS_RD; }

    for (int base = start; base < end; base += FBATCH) {
        if (t < FBATCH) {
            const int pos = base + t;
            bool use = false;
            double a[PS_RD];
            double uval = 0.;
            if (pos < end) {
                const int li = pos % ex, lj = (pos / ex) % ey, lk = pos / (ex * ey);
                const int i = bx0 + li, j = by0 + lj, k = bz0 + lk;
                if (regAt(A, 1 + axis, fd, i, j, k) == r) {
                    int3 hi = make_int3(i, j, k), lo = hi;
                    addc(lo, axis, -1);
                    const int lhi = labAt(A, 0, cd, hi.x, hi.y, hi.z), llo = labAt(A, 0, cd, lo.x, lo.y, lo.z);
                    if (MODE == MODE_MASS) use = (lhi == PS_REDUCED) || (llo == PS_REDUCED && isActiveL(lhi));
                    else if (MODE == MODE_LSQ) use = (lhi == PS_REDUCED && isActiveL(llo)) || (llo == PS_REDUCED && isActiveL(lhi));
                    else use = true;
                    if (use) {
                        double o[3];
                        faceOffset(A, axis, i, j, k, r, o);
                        basisRow(o[0], o[1], o[2], axis, a);
                        if (MODE == MODE_LSQ) uval = (double)A.vel[axis][lin3(fd, i, j, k)];
                        if (MODE == MODE_VISC) {
                            double gv[PS_RD];
                            viscosityRow(A, axis, i, j, k, gv);
#pragma unroll
                            for (int n = 0; n < PS_RD; ++n) sb[t][n] = gv[n];
                        }
                    }
                }
            }
            if (!use) {
#pragma unroll
                for (int n = 0; n < PS_RD; ++n) a[n] = 0.;
                if (MODE == MODE_VISC) {
#pragma unroll
                    for (int n = 0; n < PS_RD; ++n) sb[t][n] = 0.;
                }
            }
#pragma unroll
            for (int n = 0; n < PS_RD; ++n) sa[t][n] = a[n];
            su[t] = uval;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (t + q * BS < PS_RD * PS_RD) {
                double s = acc[q];
                if (MODE == MODE_VISC) {
                    for (int f = 0; f < FBATCH; ++f) s += sa[f][em[q]] * sb[f][en[q]];
                } else if (MODE == MODE_MASS) {
                    for (int f = 0; f < FBATCH; ++f) s += (A.rho * sa[f][em[q]]) * sa[f][en[q]];
                } else {
                    for (int f = 0; f < FBATCH; ++f) s += sa[f][em[q]] * sa[f][en[q]];
                }
                acc[q] = s;
            }
        }
        if (MODE == MODE_LSQ && t < PS_RD) {
            double s = accR;
            for (int f = 0; f < FBATCH; ++f) s += su[f] * sa[f][t];
            accR = s;
        }
        __syncthreads();
    }
    double* out = partial + (int64_t)item * OUTW;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (t + q * BS < PS_RD * PS_RD) out[t + q * BS] = acc[q];
    if (t < PS_RD) out[PS_RD * PS_RD + t] = accR;
}


// ---- MFMA variant -------------------------------------------------------------------------------------
// The 26x26 (+1 rhs column) block of a region is  M = sum_f a_f b_f^T : a GEMM with the faces as the K dimension.
// v_mfma_f64_16x16x4_f64: A operand lane l holds A[row l&15][k = l>>4], B operand lane l holds B[k = l>>4][col l&15],
// result reg q of lane l is D[row (l>>4) + 4q][col l&15] (cdna_hip_programming.md, f64 layout).
// M is padded to 32x32 = 2x2 MFMA tiles; wave w of the block owns tile (w>>1, w&1) and walks all 128 staged faces
// (32 k-steps), so no cross-wave reduction is needed.  Threads t and t+128 produce a_f and b_f of face t.
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int MPAD = 32;

template <int MODE>
__global__ void __launch_bounds__(BS) k_region_outer_mfma(TileArgs A, double* __restrict__ partial) {
    // rows padded to 33 doubles: with 32 every lane of a wave stores its row's entry n into the SAME bank pair (row stride 256 B =
    // all 64 banks) — a 32-way conflict on each of the 32 stores per face, which was most of this kernel's time
    __shared__ double sa[FBATCH][MPAD + 1];
    __shared__ double sb[FBATCH][MPAD + 1];
    const int item = blockIdx.x;
    const int r = A.itemRegion[item], axis = A.itemAxis[item], start = A.itemStart[item];
    const int bx0 = A.bbox[r * 6 + 0], by0 = A.bbox[r * 6 + 1], bz0 = A.bbox[r * 6 + 2];
    int ex = A.bbox[r * 6 + 3] - bx0 + 1, ey = A.bbox[r * 6 + 4] - by0 + 1, ez = A.bbox[r * 6 + 5] - bz0 + 1;
    if (axis == 0) ex++; else if (axis == 1) ey++; else ez++;
    const int total = ex * ey * ez;
    const int end = min(start + FB_CHUNK, total);
    const int3 fd = A.g.dims(1 + axis), cd = A.g.dims(0);
    const int t = threadIdx.x;
    const int ft = t & (FBATCH - 1);      // staged face handled by this thread
    const bool doB = t >= FBATCH;         // second half of the block produces the b vectors
    const int wave = t >> 6, lane = t & 63;
    const int ti = wave >> 1, tj = wave & 1;
    double4_t acc = {0., 0., 0., 0.};
    // MODE_LSQ sums over the faces between the region and its ACTIVE surroundings only — a sixth of the positions of a tile's box:
    // the used faces are packed (in position order: the sum is the same sequence of faces, the zero rows between them gone) until 128
    // are staged, and only then multiplied.  2.35 -> 0.6 ms at 256^3.
    __shared__ int swc[4];
    int fill = 0;                                          // block-uniform: staged faces not yet multiplied
    auto flush = [&](int rowsUsed) {
        if (ft >= rowsUsed) {                              // rows beyond the staged faces of the last k-step: zero
            double* z = doB ? sb[ft] : sa[ft];
#pragma unroll
            for (int n = 0; n < MPAD; ++n) z[n] = 0.;
        }
        __syncthreads();
        for (int ks = 0; ks < (rowsUsed + 3) / 4; ++ks) {
            const int f = 4 * ks + (lane >> 4);
            const double av = sa[f][16 * ti + (lane & 15)];
            const double bv = sb[f][16 * tj + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    };

    for (int base = start; base < end; base += FBATCH) {
        const int pos = base + ft;
        bool use = false;
        int i = 0, j = 0, k = 0;
        if (pos < end) {
            const int li = pos % ex, lj = (pos / ex) % ey, lk = pos / (ex * ey);
            i = bx0 + li; j = by0 + lj; k = bz0 + lk;
            if (regAt(A, 1 + axis, fd, i, j, k) == r) {
                int3 hi = make_int3(i, j, k), lo = hi;
                addc(lo, axis, -1);
                const int lhi = labAt(A, 0, cd, hi.x, hi.y, hi.z), llo = labAt(A, 0, cd, lo.x, lo.y, lo.z);
                if (MODE == MODE_MASS) use = (lhi == PS_REDUCED) || (llo == PS_REDUCED && isActiveL(lhi));
                else if (MODE == MODE_LSQ) use = (lhi == PS_REDUCED && isActiveL(llo)) || (llo == PS_REDUCED && isActiveL(lhi));
                else use = true;
            }
        }
        double vec[MPAD];
#pragma unroll
        for (int n = 0; n < MPAD; ++n) vec[n] = 0.;
        double gv1[MODE == MODE_VISC ? PS_RD : 1];      // MODE_VISC, first half-block: its share of b_f (cell terms + first edge axis)
        if (use) {
            if (!doB || MODE != MODE_VISC) {
                double o[3];
                faceOffset(A, axis, i, j, k, r, o);
                basisRow(o[0], o[1], o[2], axis, vec);
                if (MODE == MODE_MASS && !doB) {
#pragma unroll
                    for (int n = 0; n < PS_RD; ++n) vec[n] *= A.rho;
                }
                if (MODE == MODE_LSQ && doB) vec[PS_RD] = (double)A.vel[axis][lin3(fd, i, j, k)];   // rhs column: sum_f C_f u_f
                if (MODE == MODE_VISC) viscosityRow(A, axis, i, j, k, gv1, 1);
            } else {
                viscosityRow(A, axis, i, j, k, vec, 2);
            }
        } else if (MODE == MODE_VISC) {
#pragma unroll
            for (int n = 0; n < PS_RD; ++n) gv1[n] = 0.;
        }
        if (MODE == MODE_LSQ) {
            // rank of this face among the used ones of the batch (both halves of the block see the same 128 positions)
            const unsigned long long bal = __ballot(use);
            if (lane == 0) swc[wave] = __popcll(bal);
            __syncthreads();
            const int inBatch = swc[0] + swc[1];
            const int rank = __popcll(bal & ((1ull << lane) - 1ull)) + ((wave & 1) ? swc[wave - 1] : 0);
            if (fill + inBatch > FBATCH) { flush(fill); fill = 0; }     // (block-uniform) — flush() ends with a barrier: swc is free again
            else __syncthreads();
            if (use) {
                double* dst = doB ? sb[fill + rank] : sa[fill + rank];
#pragma unroll
                for (int n = 0; n < MPAD; ++n) dst[n] = vec[n];
            }
            fill += inBatch;
            continue;
        }
        double* dst = doB ? sb[ft] : sa[ft];
#pragma unroll
        for (int n = 0; n < MPAD; ++n) dst[n] = vec[n];
        __syncthreads();
        if (MODE == MODE_VISC) {                         // b_f = the two halves' shares
            if (!doB) {
#pragma unroll
                for (int n = 0; n < PS_RD; ++n) sb[ft][n] += gv1[n];
            }
            __syncthreads();
        }
        // 32 k-steps of 4 faces; lane l: A[m = 16 ti + (l&15)][face 4 ks + (l>>4)], B[face][n = 16 tj + (l&15)]
#pragma unroll 8
        for (int ks = 0; ks < FBATCH / 4; ++ks) {
            const int f = 4 * ks + (lane >> 4);
            const double av = sa[f][16 * ti + (lane & 15)];
            const double bv = sb[f][16 * tj + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (MODE == MODE_LSQ && fill > 0) flush(fill);
    double* out = partial + (int64_t)item * OUTW;
    const int n = 16 * tj + (lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = 16 * ti + (lane >> 4) + 4 * q;
        if (m < PS_RD) {
            if (n < PS_RD) out[m * PS_RD + n] = acc[q];
            else if (n == PS_RD) out[PS_RD * PS_RD + m] = acc[q];   // LSQ right-hand side (zero in the other modes)
        }
    }
}

// out[r] = sum of the region's item partials, in item order
__global__ void k_region_sum(const double* __restrict__ partial, const int32_t* __restrict__ itemPtr,
                             double* __restrict__ out676, double* __restrict__ out26) {
    const int r = blockIdx.x;
    for (int e = threadIdx.x; e < OUTW; e += blockDim.x) {
        double s = 0.;
        for (int it = itemPtr[r]; it < itemPtr[r + 1]; ++it) s += partial[(int64_t)it * OUTW + e];
        if (e < PS_RD * PS_RD) out676[(int64_t)r * PS_RD * PS_RD + e] = s;
        else if (out26) out26[(int64_t)r * PS_RD + (e - PS_RD * PS_RD)] = s;
    }
}

// ---- classes of identical tiles -----------------------------------------------------------------------------------------
// Mr (Solver.cpp:1405-1482) and K (:1484-1694) of a tile depend on nothing but the pattern of labels / region membership of the cells,
// faces and edges in the tile's box and the one-sample ring around it, the viscosity samples there, and the face offsets from the
// tile's centre of mass — which the pattern fixes.  A periodic tile structure has a handful of such patterns (125 among the 4096 tiles
// of the 256^3 cavity; the submerged tiles of a free-surface scene share one): the dense sums are formed once per pattern and
// copied.  (1) a 128-bit signature per region over the box + ring; (2) a device hash table keeps the smallest region index per
// signature; (3) every other region compares its pattern WORD FOR WORD with that representative's and, if equal, takes its block —
// a collision merely keeps the region on its own.  The copies differ from a tile's own sums by the rounding of `index * dx - COM` at
// another position (1e-14 relative; the oracle comparison allows 1e-12).  The least-squares fit reads velocities: not shared.
struct TileWord { unsigned long long geom; unsigned visc; bool foreign; };
__device__ inline TileWord tileWord(const TileArgs& A, int r, int i, int j, int k) {
    TileWord w{0ull, 0u, false};
    const int3 cd = A.g.dims(0);
    if (!oob3(cd, i, j, k)) {
        const int64_t c = lin3(cd, i, j, k);
        const int rg = A.reg[0][c];
        w.geom |= (unsigned long long)(A.lab[0][c] & 0xff) | ((unsigned long long)(rg == r ? 1 : (rg >= 0 ? 2 : 0)) << 8);
        if (!A.viscUniform) w.visc = __float_as_uint(A.visc[c]);
    } else w.geom |= 0xffull;
#pragma unroll
    for (int s = 1; s < 7; ++s) {
        const int3 d = A.g.dims(s);
        unsigned long long code;
        if (!oob3(d, i, j, k)) {
            const int64_t c = lin3(d, i, j, k);
            const int rg = A.reg[s][c];
            const int rc = rg == r ? 1 : (rg >= 0 ? 2 : 0);
            if (rc == 2 && s <= 3) w.foreign = true;     // a face of ANOTHER tile inside my ring: its offsets use that tile's centre — keep me on my own
            code = (unsigned long long)(A.lab[s][c] & 0xf) | ((unsigned long long)rc << 4);
        } else code = 0x3full;
        w.geom |= code << (10 + 6 * (s - 1));
    }
    return w;
}
__device__ inline unsigned long long tmix(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
constexpr long long TILE_SIG_MAX_POS = 1 << 16;   // boxes beyond this (the single region of the non-tiled mode) are not compared
__global__ void __launch_bounds__(BS) k_tile_signature(TileArgs A, unsigned long long* __restrict__ sig, int32_t* __restrict__ unique) {
    const int r = blockIdx.x;
    const int bx0 = A.bbox[r * 6 + 0] - 1, by0 = A.bbox[r * 6 + 1] - 1, bz0 = A.bbox[r * 6 + 2] - 1;
    const int ex = A.bbox[r * 6 + 3] - bx0 + 3, ey = A.bbox[r * 6 + 4] - by0 + 3, ez = A.bbox[r * 6 + 5] - bz0 + 3;   // cells -1 .. max + 2
    const long long total = (long long)ex * ey * ez;
    unsigned long long h1 = 0, h2 = 0;
    bool foreign = total > TILE_SIG_MAX_POS;
    if (!foreign)
        for (int pos = threadIdx.x; pos < (int)total; pos += BS) {
            const TileWord w = tileWord(A, r, bx0 + pos % ex, by0 + (pos / ex) % ey, bz0 + pos / (ex * ey));
            foreign |= w.foreign;
            h1 += tmix(((unsigned long long)pos << 46) ^ w.geom);
            h2 += tmix((((unsigned long long)pos * 0x9e3779b97f4a7c15ull) ^ w.geom) + ((unsigned long long)w.visc << 17) + 0x632be59bd9b4e019ull);
        }
    __shared__ unsigned long long sh[2][BS / 64];
    __shared__ int sf[BS / 64];
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_down(h1, o, 64); h2 += __shfl_down(h2, o, 64); }
    const int anyF = __any(foreign) ? 1 : 0;
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = h1; sh[1][threadIdx.x >> 6] = h2; sf[threadIdx.x >> 6] = anyF; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0; int f = 0;
        for (int q = 0; q < BS / 64; ++q) { a += sh[0][q]; b += sh[1][q]; f |= sf[q]; }
        a = tmix(a ^ (((unsigned long long)ex << 42) | ((unsigned long long)ey << 21) | (unsigned long long)ez));
        sig[2 * r] = a == 0xffffffffffffffffull ? 0ull : a;
        sig[2 * r + 1] = b;
        unique[r] = f;
    }
}
__global__ void k_tile_rep_insert(const unsigned long long* __restrict__ sig, const int32_t* __restrict__ unique, int R, unsigned long long* __restrict__ keys,
                                  int32_t* __restrict__ vals, unsigned mask) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R || unique[r]) return;
    const unsigned long long h = sig[2 * r];
    unsigned slot = (unsigned)(h >> 20) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe, slot = (slot + 1) & mask) {
        unsigned long long cur = keys[slot];
        if (cur == 0xffffffffffffffffull) { cur = atomicCAS(&keys[slot], 0xffffffffffffffffull, h); if (cur == 0xffffffffffffffffull) cur = h; }
        if (cur == h) { atomicMin(&vals[slot], r); return; }
    }
}
// rep[r] = the region whose blocks r takes (itself unless its pattern equals, word for word, that of the smallest region with its signature)
__global__ void __launch_bounds__(BS) k_tile_rep_verify(TileArgs A, const unsigned long long* __restrict__ sig, const int32_t* __restrict__ unique,
                                                        const unsigned long long* __restrict__ keys, const int32_t* __restrict__ vals, unsigned mask,
                                                        int32_t* __restrict__ rep) {
    const int r = blockIdx.x;
    int cand = r;
    if (!unique[r]) {
        const unsigned long long h = sig[2 * r];
        unsigned slot = (unsigned)(h >> 20) & mask;
        while (keys[slot] != h) slot = (slot + 1) & mask;     // present: inserted above
        cand = vals[slot];
    }
    bool same = cand != r && sig[2 * cand + 1] == sig[2 * r + 1];
    if (same) {
        const int ax0 = A.bbox[r * 6 + 0] - 1, ay0 = A.bbox[r * 6 + 1] - 1, az0 = A.bbox[r * 6 + 2] - 1;
        const int ex = A.bbox[r * 6 + 3] - ax0 + 3, ey = A.bbox[r * 6 + 4] - ay0 + 3, ez = A.bbox[r * 6 + 5] - az0 + 3;
        const int cx0 = A.bbox[cand * 6 + 0] - 1, cy0 = A.bbox[cand * 6 + 1] - 1, cz0 = A.bbox[cand * 6 + 2] - 1;
        same = ex == A.bbox[cand * 6 + 3] - cx0 + 3 && ey == A.bbox[cand * 6 + 4] - cy0 + 3 && ez == A.bbox[cand * 6 + 5] - cz0 + 3;
        if (same)
            for (int pos = threadIdx.x; pos < ex * ey * ez; pos += BS) {
                const int li = pos % ex, lj = (pos / ex) % ey, lk = pos / (ex * ey);
                const TileWord a = tileWord(A, r, ax0 + li, ay0 + lj, az0 + lk), b = tileWord(A, cand, cx0 + li, cy0 + lj, cz0 + lk);
                same = same && a.geom == b.geom && a.visc == b.visc && !a.foreign && !b.foreign;
            }
    }
    __shared__ int ok[BS / 64];
    const int all = __all(same) ? 1 : 0;
    if ((threadIdx.x & 63) == 0) ok[threadIdx.x >> 6] = all;
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 1;
        for (int q = 0; q < BS / 64; ++q) a &= ok[q];
        rep[r] = (cand != r && a) ? cand : r;
    }
}
// the blocks of the regions that share a representative's (26 x 26 entries each; a region that is its own is left alone)
__global__ void __launch_bounds__(BS) k_tile_replicate(const int32_t* __restrict__ rep, double* __restrict__ blk) {
    const int r = blockIdx.x, q = rep[r];
    if (q == r) return;
    for (int e = threadIdx.x; e < PS_RD * PS_RD; e += BS) blk[(int64_t)r * PS_RD * PS_RD + e] = blk[(int64_t)q * PS_RD * PS_RD + e];
}
// out[list[q]] = sum of the item partials of the q-th listed region, in item order
__global__ void k_region_sum_list(const double* __restrict__ partial, const int32_t* __restrict__ itemPtr, const int32_t* __restrict__ list,
                                  double* __restrict__ out676) {
    const int q = blockIdx.x, r = list[q];
    for (int e = threadIdx.x; e < PS_RD * PS_RD; e += blockDim.x) {
        double s = 0.;
        for (int it = itemPtr[q]; it < itemPtr[q + 1]; ++it) s += partial[(int64_t)it * OUTW + e];
        out676[(int64_t)r * PS_RD * PS_RD + e] = s;
    }
}

// Solver.cpp:1274-1324 + :355-371.  Exact integer sums (order independent), COM = sum * (dx / count).
__global__ void k_com(Grid g, double dx, int3 off, const int32_t* __restrict__ lab, const int32_t* __restrict__ reg,
                      const int32_t* __restrict__ bbox, double* __restrict__ COM) {
    const int r = blockIdx.x;
    const int3 d = g.dims(0);
    const int bx0 = bbox[r * 6 + 0], by0 = bbox[r * 6 + 1], bz0 = bbox[r * 6 + 2];
    const int ex = bbox[r * 6 + 3] - bx0 + 1, ey = bbox[r * 6 + 4] - by0 + 1, ez = bbox[r * 6 + 5] - bz0 + 1;
    const int64_t total = (int64_t)ex * ey * ez;
    unsigned long long sx = 0, sy = 0, sz = 0, cnt = 0;
    for (int64_t pos = threadIdx.x; pos < total; pos += blockDim.x) {
        const int i = bx0 + (int)(pos % ex), j = by0 + (int)((pos / ex) % ey), k = bz0 + (int)(pos / ((int64_t)ex * ey));
        const int64_t c = lin3(d, i, j, k);
        if (isReducedL(lab[c]) && reg[c] == r) { sx += i + off.x; sy += j + off.y; sz += k + off.z; cnt++; }
    }
    __shared__ unsigned long long sm[4][BS];
    sm[0][threadIdx.x] = sx; sm[1][threadIdx.x] = sy; sm[2][threadIdx.x] = sz; sm[3][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = BS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int q = 0; q < 4; ++q) sm[q][threadIdx.x] += sm[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        const double s = dx / (double)sm[3][0];
        COM[(int64_t)r * 3 + threadIdx.x] = (double)sm[threadIdx.x][0] * s;
    }
}

// c_fit = N.fullPivLu().solve(rhs)  (Solver.cpp:415).  Eigen FullPivLU semantics: column-major
// first-maximum pivot search, rank threshold eps*26*|maxpivot|, kernel components zero.  One wave / region.
__global__ void __launch_bounds__(64) k_lsq_solve(const double* __restrict__ N, const double* __restrict__ rhs, double* __restrict__ cfit) {
    __shared__ double lu[PS_RD][PS_RD + 1];
    __shared__ double c[PS_RD];
    __shared__ int rt[PS_RD], ct[PS_RD];
    __shared__ int s_pr, s_pc, s_nz;
    __shared__ double s_big, s_maxpivot;
    const int r = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < PS_RD * PS_RD; e += 64) lu[e / PS_RD][e % PS_RD] = N[(int64_t)r * PS_RD * PS_RD + e];
    if (lane < PS_RD) c[lane] = rhs[(int64_t)r * PS_RD + lane];
    if (lane == 0) { s_nz = PS_RD; s_maxpivot = 0.; }
    __syncthreads();
    for (int k = 0; k < PS_RD; ++k) {
        // pivot search over rows/cols >= k, column-major order, first maximum wins
        double best = -1.;
        int bestIdx = 0x7fffffff;
        const int m = PS_RD - k;
        for (int e = lane; e < m * m; e += 64) {
            const int jj = k + e / m, ii = k + e % m;
            const double a = fabs(lu[ii][jj]);
            const int idx = jj * PS_RD + ii;
            if (a > best || (a == best && idx < bestIdx)) { best = a; bestIdx = idx; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bestIdx, o, 64);
            if (ob > best || (ob == best && oi < bestIdx)) { best = ob; bestIdx = oi; }
        }
        if (lane == 0) { s_big = best; s_pc = bestIdx / PS_RD; s_pr = bestIdx % PS_RD; }
        __syncthreads();
        if (s_big == 0.) {
            if (lane == 0) { s_nz = k; for (int i = k; i < PS_RD; ++i) { rt[i] = i; ct[i] = i; } }
            __syncthreads();
            break;
        }
        const int pr = s_pr, pc = s_pc;
        if (lane == 0) { if (s_big > s_maxpivot) s_maxpivot = s_big; rt[k] = pr; ct[k] = pc; }
        if (lane < PS_RD && pr != k) { const double tv = lu[k][lane]; lu[k][lane] = lu[pr][lane]; lu[pr][lane] = tv; }
        __syncthreads();
        if (lane < PS_RD && pc != k) { const double tv = lu[lane][k]; lu[lane][k] = lu[lane][pc]; lu[lane][pc] = tv; }
        __syncthreads();
        if (lane > k && lane < PS_RD) lu[lane][k] /= lu[k][k];
        __syncthreads();
        if (lane > k && lane < PS_RD) {
            const double f = lu[lane][k];
            for (int j = k + 1; j < PS_RD; ++j) lu[lane][j] -= f * lu[k][j];
        }
        __syncthreads();
    }
    if (lane == 0) {
        const int n = PS_RD;
        const double thresh = fabs(s_maxpivot) * (2.220446049250313e-16 * (double)n);
        int rank = 0;
        for (int i = 0; i < s_nz; ++i) rank += (fabs(lu[i][i]) > thresh);
        double x[PS_RD];
        for (int i = 0; i < n; ++i) x[i] = 0.;
        if (rank > 0) {
            for (int k = 0; k < n; ++k) { const double tv = c[k]; c[k] = c[rt[k]]; c[rt[k]] = tv; }
            for (int i = 0; i < n; ++i) { double s = c[i]; for (int j = 0; j < i; ++j) s -= lu[i][j] * c[j]; c[i] = s; }
            for (int i = rank - 1; i >= 0; --i) {
                double s = c[i];
                for (int j = i + 1; j < rank; ++j) s -= lu[i][j] * c[j];
                c[i] = s / lu[i][i];
            }
            int colperm[PS_RD];
            for (int i = 0; i < n; ++i) colperm[i] = i;
            for (int k = 0; k < n; ++k) { const int tv = colperm[k]; colperm[k] = colperm[ct[k]]; colperm[ct[k]] = tv; }
            for (int i = 0; i < rank; ++i) x[colperm[i]] = c[i];
        }
        for (int i = 0; i < n; ++i) cfit[(int64_t)r * PS_RD + i] = x[i];
    }
}

// B = invDt*Mr + 2K; Binv = B^-1 (Eigen PartialPivLU inverse, AssembleBlocks.cpp:208-209);
// rhs_r = Mr * c_fit (AssembleBlocks.cpp:356-367).  One wave per region.
__global__ void __launch_bounds__(64) k_binv(double invDt, const double* __restrict__ Mr, const double* __restrict__ Kv,
                                             const double* __restrict__ cfit, double* __restrict__ Binv, double* __restrict__ rhsR) {
    __shared__ double lu[PS_RD][PS_RD + 1];
    __shared__ int perm[PS_RD];
    __shared__ int s_pr;
    __shared__ double s_big;
    const int r = blockIdx.x, lane = threadIdx.x;
    const int64_t o = (int64_t)r * PS_RD * PS_RD;
    for (int e = lane; e < PS_RD * PS_RD; e += 64) lu[e / PS_RD][e % PS_RD] = invDt * Mr[o + e] + 2. * Kv[o + e];
    if (lane < PS_RD) {
        perm[lane] = lane;
        double s = 0;
        for (int n = 0; n < PS_RD; ++n) s += Mr[o + lane * PS_RD + n] * cfit[(int64_t)r * PS_RD + n];
        rhsR[(int64_t)r * PS_RD + lane] = s;
    }
    __syncthreads();
    for (int k = 0; k < PS_RD; ++k) {
        if (lane == 0) {
            int pr = k;
            double biggest = fabs(lu[k][k]);
            for (int i = k + 1; i < PS_RD; ++i) if (fabs(lu[i][k]) > biggest) { biggest = fabs(lu[i][k]); pr = i; }
            s_pr = pr; s_big = biggest;
            if (biggest != 0. && pr != k) { const int tv = perm[k]; perm[k] = perm[pr]; perm[pr] = tv; }
        }
        __syncthreads();
        if (s_big == 0.) continue;
        const int pr = s_pr;
        if (lane < PS_RD && pr != k) { const double tv = lu[k][lane]; lu[k][lane] = lu[pr][lane]; lu[pr][lane] = tv; }
        __syncthreads();
        if (lane > k && lane < PS_RD) {
            lu[lane][k] /= lu[k][k];
            const double f = lu[lane][k];
            for (int j = k + 1; j < PS_RD; ++j) lu[lane][j] -= f * lu[k][j];
        }
        __syncthreads();
    }
    if (lane < PS_RD) {
        double y[PS_RD];
        for (int i = 0; i < PS_RD; ++i) y[i] = (perm[i] == lane) ? 1. : 0.;
        for (int i = 0; i < PS_RD; ++i) { double s = y[i]; for (int j = 0; j < i; ++j) s -= lu[i][j] * y[j]; y[i] = s; }
        for (int i = PS_RD - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < PS_RD; ++j) s -= lu[i][j] * y[j]; y[i] = s / lu[i][i]; }
        for (int i = 0; i < PS_RD; ++i) Binv[o + i * PS_RD + lane] = y[i];
    }
}

TileArgs makeArgs(ps_context* c) {
    TileArgs A;
    A.g = c->g; A.dx = c->dx; A.rho = c->rho;
    for (int s = 0; s < 7; ++s) { A.lab[s] = c->labels[s].p; A.reg[s] = c->reducedIdx[s].p; }
    for (int a = 0; a < 3; ++a) A.vel[a] = c->vel[a].p;
    A.visc = c->viscosity.p;
    A.viscUniform = c->viscUniform ? 1 : 0; A.viscValue = c->viscUniformValue;
    A.COM = c->COM.p;
    A.off = make_int3(c->gOff[0], c->gOff[1], c->gOff[2]);
    A.bbox = c->bbox.p;
    A.itemRegion = c->fbItemRegion.p; A.itemAxis = c->fbItemAxis.p; A.itemStart = c->fbItemStart.p;
    return A;
}

template <int MODE>
void runOuter(ps_context* c, double* out676, double* out26) {
    if (c->regionCount == 0 || c->fbItems == 0) return;
    c->partials.alloc((size_t)c->fbItems * OUTW);
    TileArgs A = makeArgs(c);
    static const bool useMfma = !(PS_ENV("PS_TILE_VALU") && atoi(PS_ENV("PS_TILE_VALU")) != 0);
    // Which blocks are shared (PS_TILE_CLASS_MASK; bit 0: Mr, bit 1: K).  Default: K only.  A copied block differs from the tile's own sums
    // by the rounding of `index * dx - COM` at the representative's position (Mr 3e-16, K 4e-14 relative to the oracle's either way).  On
    // the stiff 48^3 spheres at tol 1e-8 — where a 1e-15 change of B moves the velocities by 1e-5..1e-4, the AMP sensitivity of
    // DESIGN.md section 4 — the distance to the oracle's velocities is 6.8e-5 with own sums, 7.3e-5 with K shared and 1.04e-4 with Mr
    // shared (scripts/ladder_tile_classes.py): Mr carries the tile's rigid modes, which the large pressure-stress terms cancel against.
    // So Mr (2.3 of the 7.8 ms at 256^3) stays per tile and the tolerance ladder of tests/test_gpu_parity.py keeps its bounds.
    static const int clsMask = PS_ENV("PS_TILE_CLASS_MASK") ? atoi(PS_ENV("PS_TILE_CLASS_MASK")) : 2;
    if (MODE != MODE_LSQ && ((MODE == MODE_MASS ? 1 : 2) & clsMask) && c->tileReps > 0 && c->tileReps < c->regionCount) {   // one sum per class of identical tiles, copied to the others
        A.itemRegion = c->repItemRegion.p; A.itemAxis = c->repItemAxis.p; A.itemStart = c->repItemStart.p;
        if (useMfma) hipLaunchKernelGGL(k_region_outer_mfma<MODE>, dim3((unsigned)c->repItems), dim3(BS), 0, c->stream, A, c->partials.p);
        else hipLaunchKernelGGL(k_region_outer<MODE>, dim3((unsigned)c->repItems), dim3(BS), 0, c->stream, A, c->partials.p);
        hipLaunchKernelGGL(k_region_sum_list, dim3((unsigned)c->tileReps), dim3(BS), 0, c->stream, c->partials.p, c->repRegionItemPtr.p, c->repList.p, out676);
        hipLaunchKernelGGL(k_tile_replicate, dim3((unsigned)c->regionCount), dim3(BS), 0, c->stream, (const int32_t*)c->tileRep.p, out676);
        return;
    }
    if (useMfma) hipLaunchKernelGGL(k_region_outer_mfma<MODE>, dim3((unsigned)c->fbItems), dim3(BS), 0, c->stream, A, c->partials.p);
    else hipLaunchKernelGGL(k_region_outer<MODE>, dim3((unsigned)c->fbItems), dim3(BS), 0, c->stream, A, c->partials.p);
    hipLaunchKernelGGL(k_region_sum, dim3((unsigned)c->regionCount), dim3(BS), 0, c->stream, c->partials.p, c->fbRegionItemPtr.p, out676, out26);
}

}  // namespace

// Solver.cpp:328-372.  Also builds the per-region face-box work table used by all dense reductions.
void ps_context::computeCenterOfMasses() {
    computeRegionBoxes();
    const int64_t R = regionCount;
    COM.alloc((size_t)R * 3); cfit.alloc((size_t)R * PS_RD);
    Mr.alloc((size_t)R * PS_RD * PS_RD); Kv.alloc((size_t)R * PS_RD * PS_RD); Binv.alloc((size_t)R * PS_RD * PS_RD);
    rhsR.alloc((size_t)R * PS_RD);
    regionScratch.alloc((size_t)R * OUTW);
    if (R == 0) { fbItems = 0; return; }
    if (slabEnabled) {   // a tile is owned iff all of its cells are in my box; cuts are tile aligned
        std::vector<int32_t> ro((size_t)R, 1);
        for (int64_t r = 0; r < R; ++r) {
            bool inside = true, outside = false;
            for (int a = 0; a < 3; ++a) {
                const int mn = hbbox[(size_t)r * 6 + a], mx = hbbox[(size_t)r * 6 + 3 + a];
                inside = inside && mn >= brick.lo[a] && mx < brick.hi[a];
                outside = outside || mx < brick.lo[a] || mn >= brick.hi[a];
            }
            if (!inside && !outside) throw Error("a reduced region straddles a cut of the decomposition (cuts must be tile aligned; doTile required)");
            ro[(size_t)r] = inside ? 1 : 0;
        }
        regionOwned.alloc((size_t)R);
        HIP_CHECK(hipMemcpyAsync(regionOwned.p, ro.data(), (size_t)R * 4, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    // (host tables kept with the context — hostTab* — so that their uploads need no synchronisation: the vectors outlive the copies)
    std::vector<int32_t>& iR = hostTab[0]; std::vector<int32_t>& iA = hostTab[1]; std::vector<int32_t>& iS = hostTab[2]; std::vector<int32_t>& ptr = hostTab[3];
    iR.clear(); iA.clear(); iS.clear(); ptr.assign((size_t)R + 1, 0);
    for (int64_t r = 0; r < R; ++r) {
        ptr[(size_t)r] = (int32_t)iR.size();
        for (int a = 0; a < 3; ++a) {
            int64_t e[3];
            for (int q = 0; q < 3; ++q) e[q] = (int64_t)hbbox[(size_t)r * 6 + 3 + q] - hbbox[(size_t)r * 6 + q] + 1;
            e[a] += 1;
            const int64_t total = e[0] * e[1] * e[2];
            if (total > 0x7fffffff) throw Error("region face box too large");
            for (int64_t st = 0; st < total; st += FB_CHUNK) { iR.push_back((int32_t)r); iA.push_back(a); iS.push_back((int32_t)st); }
        }
    }
    ptr[(size_t)R] = (int32_t)iR.size();
    fbItems = (int64_t)iR.size();
    fbItemRegion.alloc(iR.size()); fbItemAxis.alloc(iR.size()); fbItemStart.alloc(iR.size()); fbRegionItemPtr.alloc(ptr.size());
    {   // skin-row enumeration items: the union face box of each region, FB_CHUNK positions per item
        std::vector<int32_t>& sR = hostTab[4]; std::vector<int32_t>& sS = hostTab[5];
        sR.clear(); sS.clear();
        sbRegionItemPtrHost.assign((size_t)R + 1, 0);
        for (int64_t r = 0; r < R; ++r) {
            sbRegionItemPtrHost[(size_t)r] = (int32_t)sR.size();
            int64_t total = 1;
            for (int q = 0; q < 3; ++q) total *= (int64_t)hbbox[(size_t)r * 6 + 3 + q] - hbbox[(size_t)r * 6 + q] + 2;
            if (total > 0x7fffffff) throw Error("region face box too large");
            for (int64_t st = 0; st < total; st += FB_CHUNK) { sR.push_back((int32_t)r); sS.push_back((int32_t)st); }
        }
        sbRegionItemPtrHost[(size_t)R] = (int32_t)sR.size();
        sbItems = (int64_t)sR.size();
        sbItemRegion.alloc(sR.size()); sbItemStart.alloc(sS.size()); sbItemCount.alloc(sR.size() + 1);
        HIP_CHECK(hipMemcpyAsync(sbItemRegion.p, sR.data(), sR.size() * 4, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(sbItemStart.p, sS.data(), sS.size() * 4, hipMemcpyHostToDevice, stream));
    }
    HIP_CHECK(hipMemcpyAsync(fbItemRegion.p, iR.data(), iR.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(fbItemAxis.p, iA.data(), iA.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(fbItemStart.p, iS.data(), iS.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(fbRegionItemPtr.p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_com, dim3((unsigned)R), dim3(BS), 0, stream, g, dx, make_int3(gOff[0], gOff[1], gOff[2]), labels[0].p, reducedIdx[0].p, bbox.p, COM.p);
    buildTileClasses();
}

// Classes of identical tiles (see k_tile_signature): tileRep[r], and the face-box items of the representatives alone.
void ps_context::buildTileClasses() {
    tileReps = 0; repItems = 0;
    const int64_t R = regionCount;
    static const bool off = PS_ENV("PS_NO_TILE_CLASSES") && atoi(PS_ENV("PS_NO_TILE_CLASSES")) != 0;   // A/B: every tile sums its own blocks
    if (off || R < 2) return;
    TileArgs A = makeArgs(this);
    unsigned cap = 1024;
    while (cap < 4u * (unsigned)R) cap <<= 1;
    tileSig.alloc((size_t)R * 2); tileUnique.alloc((size_t)R); tileRep.alloc((size_t)R);
    DevBuf<unsigned long long>& keys = scrKeys; DevBuf<int32_t>& vals = scrVals;
    keys.alloc(cap); vals.alloc(cap);
    HIP_CHECK(hipMemsetAsync(keys.p, 0xff, (size_t)cap * 8, stream));
    HIP_CHECK(hipMemsetAsync(vals.p, 0x7f, (size_t)cap * 4, stream));
    hipLaunchKernelGGL(k_tile_signature, dim3((unsigned)R), dim3(BS), 0, stream, A, tileSig.p, tileUnique.p);
    hipLaunchKernelGGL(k_tile_rep_insert, dim3(gridFor(R, BS)), dim3(BS), 0, stream, (const unsigned long long*)tileSig.p, (const int32_t*)tileUnique.p, (int)R, keys.p, vals.p, cap - 1);
    hipLaunchKernelGGL(k_tile_rep_verify, dim3((unsigned)R), dim3(BS), 0, stream, A, (const unsigned long long*)tileSig.p, (const int32_t*)tileUnique.p,
                       (const unsigned long long*)keys.p, (const int32_t*)vals.p, cap - 1, tileRep.p);
    std::vector<int32_t>& rep = hostTab[11];
    rep.assign((size_t)R, 0);
    HIP_CHECK(hipMemcpyAsync(rep.data(), tileRep.p, (size_t)R * 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int32_t>& iR = hostTab[12]; std::vector<int32_t>& iA = hostTab[13]; std::vector<int32_t>& iS = hostTab[14];
    std::vector<int32_t>& ptr = hostTab[15]; std::vector<int32_t>& list = hostTab[16];
    iR.clear(); iA.clear(); iS.clear(); ptr.clear(); list.clear();
    for (int64_t r = 0; r < R; ++r) {
        if (rep[(size_t)r] != (int32_t)r) continue;
        list.push_back((int32_t)r);
        ptr.push_back((int32_t)iR.size());
        for (int a = 0; a < 3; ++a) {                 // the same items, in the same order, as computeCenterOfMasses builds for every region
            int64_t e[3];
            for (int q = 0; q < 3; ++q) e[q] = (int64_t)hbbox[(size_t)r * 6 + 3 + q] - hbbox[(size_t)r * 6 + q] + 1;
            e[a] += 1;
            const int64_t total = e[0] * e[1] * e[2];
            for (int64_t st = 0; st < total; st += FB_CHUNK) { iR.push_back((int32_t)r); iA.push_back(a); iS.push_back((int32_t)st); }
        }
    }
    ptr.push_back((int32_t)iR.size());
    if ((int64_t)list.size() == R) return;            // nothing repeats: the plain path
    tileReps = (int64_t)list.size(); repItems = (int64_t)iR.size();
    repItemRegion.alloc(iR.size()); repItemAxis.alloc(iA.size()); repItemStart.alloc(iS.size()); repRegionItemPtr.alloc(ptr.size()); repList.alloc(list.size());
    HIP_CHECK(hipMemcpyAsync(repItemRegion.p, iR.data(), iR.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(repItemAxis.p, iA.data(), iA.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(repItemStart.p, iS.data(), iS.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(repRegionItemPtr.p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(repList.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, stream));
    if (PS_ENV_VERBOSE()) std::fprintf(stderr, "[polystokes] tile classes: %lld of %lld tiles sum their own Mr / K\n", (long long)tileReps, (long long)R);
}

void ps_context::computeLeastSquaresFits() {
    if (regionCount == 0) return;
    double* N = regionScratch.p;                                   // R*676
    double* rhs = regionScratch.p + (size_t)regionCount * PS_RD * PS_RD;   // R*26
    runOuter<MODE_LSQ>(this, N, rhs);
    hipLaunchKernelGGL(k_lsq_solve, dim3((unsigned)regionCount), dim3(64), 0, stream, N, rhs, cfit.p);
}
void ps_context::computeReducedMassMatrices() { runOuter<MODE_MASS>(this, Mr.p, nullptr); }
void ps_context::computeReducedViscosityMatricesInteriorOnly() { runOuter<MODE_VISC>(this, Kv.p, nullptr); }

void ps_context::assembleReducedBlocks() {
    if (regionCount == 0) return;
    hipLaunchKernelGGL(k_binv, dim3((unsigned)regionCount), dim3(64), 0, stream, invDt, Mr.p, Kv.p, cfit.p, Binv.p, rhsR.p);
}

// Shared helpers for the HIP kernels of libmfem_amd (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include "../../include/mfem_amd.h"

namespace mfa
{

constexpr int MAX_D1D = 10;
constexpr int MAX_Q1D = 14;
constexpr int MAX_DQ  = 140;   // D1D*Q1D limit of the by-value basis tables

// 1D basis tables passed BY VALUE in the kernel-argument segment: every access with a
// compile-time (unrolled) index becomes a scalar load into SGPRs, and v_fma_f64 takes the
// SGPR pair directly -- no LDS or VGPR traffic for B/G in the contraction loops.
// Layout: b[q*D + d] (q-major, so a thread's row q is contiguous).
template <int N>
struct BasisT
{
   double b[N];
   double g[N];
};

// The same with the transposed tables bt[d*Q + q], gt[d*Q + q] beside them (the specialised apply kernels:
// D*Q <= 90, 2880 B of the 4 KB kernel-argument segment): a contraction phase whose unrolled loops walk q
// fastest reads contiguous scalar runs too -- with strided scalar operands the (9,Q) kernels ran out of SGPRs.
template <int N>
struct BasisTT
{
   double b[N];
   double g[N];
   double bt[N];
   double gt[N];
};

// Elements per workgroup of the specialised apply kernels (Q*Q*NBZ ~ 200..256 threads; Q = 8: two
// one-wave elements -- measured 6 % faster than four for the collocated and vector kernels, equal
// otherwise; halving the other sizes was equal or slower, profiles/r1/sweep_nbz_half.txt).
// Also the blocking factor of the HBM-friendly pa_data layout below.
// (MFA_NBZ<Q>: tuning overrides -- every source file must see the same values.)
#ifndef MFA_NBZ3
#define MFA_NBZ3 16
#endif
#ifndef MFA_NBZ4
#define MFA_NBZ4 16
#endif
#ifndef MFA_NBZ5
#define MFA_NBZ5 10
#endif
#ifndef MFA_NBZ6
#define MFA_NBZ6 7
#endif
#ifndef MFA_NBZ7
#define MFA_NBZ7 5
#endif
#ifndef MFA_NBZ8
#define MFA_NBZ8 2
#endif
// (per operator where the two disagree: the mass kernel's Q = 3 batch, profiles/r3/nbz3_probe.txt)
#ifndef MFA_NBZ3M
#define MFA_NBZ3M 28
#endif
#ifndef MFA_NBZ8M
#define MFA_NBZ8M MFA_NBZ8
#endif
__host__ __device__ constexpr int nbz_for(int kind, int Q)
{
   if (kind == 0 && Q == 3) { return MFA_NBZ3M; }
   if (kind == 0 && Q == 8) { return MFA_NBZ8M; }
   return Q <= 2 ? 32 : Q == 3 ? MFA_NBZ3 : Q == 4 ? MFA_NBZ4 : Q == 5 ? MFA_NBZ5 : Q == 6 ? MFA_NBZ6 :
          Q == 7 ? MFA_NBZ7 : Q == 8 ? MFA_NBZ8 : Q == 9 ? 3 : 2;
}

// Blocked pa_data layout (layout 1).  The reference stores D as (q, component, element)
// (fem/integ/bilininteg_diffusion_kernels.cpp:257); a thread sheet (qx,qy) walking qz then
// reads Q^2-long segments that start at arbitrary 8-byte offsets.  Layout 1 stores, per
// batch of NBZ = nbz_for(Q) elements, [qz][component pair][element in batch][qy][qx][2]
// (diffusion; mass: [qz][element in batch][qy][qx]), so that one load instruction of the
// whole workgroup reads ONE contiguous, 16-byte-per-lane (mass: 8) run of 4 KB (2 KB).
// Same doubles, same count (plus padding of the last batch): the algorithmic bytes do
// not change.
__host__ __device__ inline int64_t blocked_index(int kind, int Q, int64_t e, int q, int c)
{
   const int NBZ = nbz_for(kind, Q), QQ = Q*Q;
   const int64_t batch = e/NBZ;
   const int tz = (int)(e % NBZ), qz = q/QQ, qq = q % QQ;
   if (kind) { return ((((batch*Q + qz)*3 + (c >> 1))*NBZ + tz)*QQ + qq)*2 + (c & 1); }
   return ((batch*Q + qz)*NBZ + tz)*(int64_t)QQ + qq;
}

__host__ __device__ inline int64_t blocked_size(int kind, int Q, int64_t ne)
{
   const int NBZ = nbz_for(kind, Q);
   return ((ne + NBZ - 1)/NBZ)*NBZ*(int64_t)(kind ? 6 : 1)*Q*Q*Q;
}

// Named ranges around the stages of the path, the role of MFEM_PERF_SCOPE (general/annotation.hpp:21-31, Caliper
// there): with MFEM_AMD_TRACE=1 in the environment they are roctx ranges (`rocprofv3 --marker-trace --kernel-trace`
// shows which kernels belong to which stage); otherwise they cost one load and a branch.  The roctx library is
// looked up at run time (librocprofiler-sdk-roctx, else libroctx64): nothing is linked.
void trace_push(const char *name);
void trace_pop();
struct TraceRange
{
   explicit TraceRange(const char *name) { trace_push(name); }
   ~TraceRange() { trace_pop(); }
   TraceRange(const TraceRange&) = delete;
   TraceRange &operator=(const TraceRange&) = delete;
};

void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);
bool report_kernels_enabled();
void report_fallback(const char *name, int D1D, int Q1D);

#define MFA_CHECK_LAUNCH(name) \
   do { if (int rc_ = ::mfa::check_hip(hipGetLastError(), name)) { return rc_; } } while (0)

#define MFA_REQUIRE(cond, ...) \
   do { if (!(cond)) { ::mfa::set_error(__VA_ARGS__); return 1; } } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// XCD-aware block remap: hardware deals workgroups round-robin over the 8 XCDs, so blocks
// b and b+8 share an L2.  Map the launch index to a logical block id such that each XCD
// owns one contiguous range of logical blocks (neighbouring elements share L-dofs, so
// their x gathers and the straddling cache lines of pa_data hit the same L2).
// Bijective for any nb (guide: cdna_hip_programming.md, 256^2 template notes).
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nb)
{
   const int64_t q = nb >> 3, r = nb & 7;
   const int64_t xcd = bid & 7, k = bid >> 3;
   const int64_t base = (xcd < r) ? xcd*(q + 1) : r*(q + 1) + (xcd - r)*q;
   return base + k;
}

// Wavefront (64 lanes) sum, result valid in lane 0.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off, 64); }
   return v;
}

// Vector H1 spaces: arguments of the component loop of k_pa_apply (pa_apply.hip)
struct VecArgs
{
   int vd;                        // components
   int byvdim;                    // L-vector ordering: 0 byNODES (x[c*ndofs + i]), 1 byVDIM (x[i*vd + c])
   int64_t ndofs;                 // scalar L-dofs
   const unsigned char *marker;   // optional, vd*ndofs bytes indexed like x: nonzero reads as 0.0 (essential dofs)
   int64_t d_comp_stride;         // 0: every component uses the same pa_data; else blocked layout: doubles between
                                  // the components' arrays; MFEM layout: nonzero = (Q^3,NC,vd,NE) as the reference
   int64_t d_block;               // MFEM layout: doubles per (element, component) block; 0 = NC*Q^3.  The reference's
                                  // vector diffusion allocates dim*dim = 9 planes per block and fills 6
   double *ydirect;               // scalar gather kernels, optional: the L-vector rows of the element-interior dofs, element e's
                                  // (D-2)^3 rows at ydirect + e*(D-2)^3 in lexicographic order -- written here instead of
                                  // to the E-vector (mfem_amd_interior_rows); the E-vector is then written in the packed
                                  // form below (packed_pos)
};

// Packed E-vector (used together with the direct interior rows).  Once the element-interior entries go straight to the
// L-vector, the reference's (D,D,D,NE) E-vector is written with holes -- runs of D-2 missing doubles in every inner row --
// so every 32/64-B sector of it still travels to memory and back, and an L-row block of the transpose-gather (MFEM numbers
// vertex, edge, face dofs class after class) finds the entries it wants one or two to a 128-B line.  The packed form stores
// the NS = D^3 - (D-2)^3 surviving entries of an element contiguously and class by class -- 8 vertex entries, then the 12
// edges (D-2 each), then the 6 faces ((D-2)^2 each) -- at e*NS + packed_pos(dz,dy,dx); the transpose-gather reads it
// through `indices` remapped once at set-up (mfem_amd_pack_indices): same values, same order of additions, same bits.
// Order inside a class: every row (dz,dy) of the element kernel's store phase owns one contiguous run of D-2 entries (an
// x-edge, a row of a z- or y-face) plus its two end entries.  Returns -1 for an element-interior entry.
__host__ __device__ inline int packed_pos(const int D, const int dz, const int dy, const int dx)
{
   const int DI = D - 2, E0 = 8, F0 = 8 + 12*DI;
   const bool nz = dz >= 1 && dz <= DI, ny = dy >= 1 && dy <= DI, nx = dx >= 1 && dx <= DI;
   const int bz = dz ? 1 : 0, by = dy ? 1 : 0, bx = dx ? 1 : 0;       // side, where the coordinate is on the boundary
   const int iz = dz - 1, iy = dy - 1, ix = dx - 1;
   const int nin = (nz ? 1 : 0) + (ny ? 1 : 0) + (nx ? 1 : 0);
   if (nin == 0) { return bz*4 + by*2 + bx; }
   if (nin == 1)
   {
      if (nx) { return E0 + (bz*2 + by)*DI + ix; }
      if (ny) { return E0 + 4*DI + (bz*2 + bx)*DI + iy; }
      return E0 + 8*DI + (by*2 + bx)*DI + iz;
   }
   if (nin == 2)
   {
      if (!nz) { return F0 + (bz*DI + iy)*DI + ix; }
      if (!ny) { return F0 + 2*DI*DI + (by*DI + iz)*DI + ix; }
      return F0 + 4*DI*DI + (bx*DI + iz)*DI + iy;
   }
   return -1;
}
__host__ __device__ constexpr int packed_size(const int D) { return D*D*D - (D - 2)*(D - 2)*(D - 2); }

// Workgroup sum for blockDim.x == 256 (4 waves); result valid in thread 0.
// Fixed tree: lanes by shuffle, then waves 0..3 added in order.
__device__ __forceinline__ double block_sum_256(double v, double *s4)
{
   v = wave_sum(v);
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   if (lane == 0) { s4[w] = v; }
   __syncthreads();
   double r = 0.0;
   if (threadIdx.x == 0) { r = ((s4[0] + s4[1]) + s4[2]) + s4[3]; }
   return r;
}

} // namespace mfa

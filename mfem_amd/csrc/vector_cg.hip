// Vector kernels, deterministic reductions and the device-resident Jacobi-PCG (gfx950).
//
// Replaces linalg/vector.cpp:203-240 (operator=), :436-473 (add), :570 (subtract),
// :1079-1152 (operator*) + general/reducers.hpp:451-593 (reduce), linalg/solvers.cpp:401-453
// (OperatorJacobiSmoother) and linalg/solvers.cpp:869-1050 (CGSolver::Mult).
//
// The reference runs one kernel per BLAS-1 call (22 vector passes per PCG iteration) and
// synchronises the host twice per iteration to read the dot products.  Here one iteration
// is: kernel A {r -= alpha z; z = dinv.r; partial r.z}, a one-workgroup scalar kernel,
// kernel B {x += alpha d; d = z + beta d}, the fused gather+apply kernel, the
// transpose-gather kernel with the d.Ad partial folded into its epilogue, and a second
// scalar kernel -- 11 vector passes; alpha/beta/(Br,r) and the stop flag stay on the device.
// All reductions have a fixed shape that depends on n only (wavefront shuffles, fixed-order
// LDS combine, fixed-order final pass), so repeated runs are bit-identical.
#include "common.h"
#include <cmath>
#include <vector>

namespace mfa
{

constexpr int RED_BLOCKS_MAX = 8192;       // partials per reduction

int pa_apply_L2E_vec(int kind, int layout, int D, int Q, int64_t ne, const double *B, const double *G,
                     const double *Dq, const int *gmap, const double *x, double *yE, VecArgs va, hipStream_t s);
int scatter_dot_vec(int64_t ndofs, int nd, int vd, int byvdim, const int *offsets, const int *indices,
                    const double *xE, double *y, const unsigned char *marker, const double *xin,
                    const double *w, double *partials, int nblocks, const int *skip, hipStream_t s);
int pa_apply_L2E(int kind, int layout, int D, int Q, int64_t ne, const double *B, const double *G,
                 const double *Dq, const int *gmap, const double *x, double *yE, hipStream_t s, double *ydir);
int scatter_list(int64_t nlist, const int *rows, int64_t nrows, const int *offsets, const int *indices, const double *xE,
                 double *y, const unsigned char *marker, const double *xin, const int *skip, hipStream_t s);
int scatter_dot(int64_t ndofs, const int *offsets, const int *indices, const double *xE,
                double *y, const unsigned char *marker, const double *xin, const double *w,
                double *partials, int nblocks, const int *skip, hipStream_t s, int64_t ndirect);

static inline int red_blocks(int64_t n)
{
   int64_t nb = (n + 256*4 - 1)/(256*4);     // >= 4 elements per thread before adding blocks
   if (nb < 1) { nb = 1; }
   if (nb > RED_BLOCKS_MAX) { nb = RED_BLOCKS_MAX; }
   return (int)nb;
}

__global__ void __launch_bounds__(256)
k_axpby(const int64_t n, const double a, const double *x, const double b, const double *y,
        double *z)
{
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
   {
      z[i] = y ? a*x[i] + b*y[i] : a*x[i];
   }
}

__global__ void __launch_bounds__(256)
k_fill(const int64_t n, const double v, double *__restrict__ x)
{
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256) { x[i] = v; }
}

__global__ void __launch_bounds__(256)
k_dot_partial(const int64_t n, const double *__restrict__ x, const double *__restrict__ y,
              double *__restrict__ partials)
{
   __shared__ double s4[4];
   double acc = 0.0;
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
   {
      acc += x[i]*y[i];
   }
   const double r = block_sum_256(acc, s4);
   if (threadIdx.x == 0) { partials[blockIdx.x] = r; }
}

// Sum of nb (<= RED_BLOCKS_MAX) partials by one workgroup, fixed order: thread t adds partials
// t, t+256, ... then the fixed block tree.
__device__ __forceinline__ double final_sum(const double *partials, const int nb, double *s4)
{
   // eight independent loads in flight per thread (one workgroup reads up to 8192 partials:
   // a chain of dependent loads here costs ~12 us per reduction); the additions keep the
   // order t, t+256, t+512, ...
   double acc = 0.0;
   for (int i0 = threadIdx.x; i0 < nb; i0 += 256*8)
   {
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { const int i = i0 + k*256; t[k] = partials[i < nb ? i : nb - 1]; }
#pragma unroll
      for (int k = 0; k < 8; k++) { if (i0 + k*256 < nb) { acc += t[k]; } }
   }
   return block_sum_256(acc, s4);
}

__global__ void __launch_bounds__(256)
k_dot_final(const int nb, const double *__restrict__ partials, double *__restrict__ result)
{
   __shared__ double s4[4];
   const double r = final_sum(partials, nb, s4);
   if (threadIdx.x == 0) { *result = r; }
}

__global__ void __launch_bounds__(256)
k_jacobi_setup(const int64_t n, const double *__restrict__ diag, const double damping,
               double *__restrict__ dinv)
{
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
   {
      dinv[i] = damping/diag[i];
   }
}

__global__ void __launch_bounds__(256)
k_mul(const int64_t n, const double *__restrict__ a, const double *__restrict__ b,
      double *__restrict__ z)
{
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
   {
      z[i] = a[i]*b[i];
   }
}

#ifndef MFA_PCG_UNROLL
#define MFA_PCG_UNROLL 2
#endif
// ---- device-resident PCG state -------------------------------------------------------
struct PcgState
{
   double nom, den, betanom, alpha, beta, r0, nom0;
   int iter;        // iterations completed (CGSolver final_iter when done)
   int done;        // stop flag: kernels become no-ops
   int converged;
   int max_iter;
   double rel_tol, abs_tol;
   int hist_n;      // entries of the (B r, r) history written so far
   int k;           // k_pcg_beta_scalars launches so far (no-op launches after the end included)
   int upd;         // value of k at the last launch that followed a kernel A that ran: kernel B owes x its update
};

// den = (d, A d) after the first/each operator application (linalg/solvers.cpp:931-949, 1010-1025);
// nom <- betanom at the end of an iteration.  One thread.
__device__ __forceinline__ void pcg_den_step(PcgState *st, const double den, const int first)
{
   st->den = den;
   if (!first) { st->nom = st->betanom; }
   if (den == 0.0 || den != den)                           // (linalg/solvers.cpp:940-948, 1019-1023)
   {
      st->done = 1;
      if (!first) { st->iter += 1; }                       // the reference has already done ++i
   }
   else { st->alpha = st->nom/den; }
}

// r = b; x = 0; z = dinv.r (or r); d = z; partial d.r
__global__ void __launch_bounds__(256)
k_pcg_init(const int64_t n, const double *__restrict__ b, const double *__restrict__ dinv,
           double *__restrict__ x, double *__restrict__ r, double *__restrict__ d,
           double *__restrict__ partials)
{
   __shared__ double s4[4];
   double acc = 0.0;
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
   {
      const double ri = b[i];
      const double zi = dinv ? dinv[i]*ri : ri;
      x[i] = 0.0; r[i] = ri; d[i] = zi;
      acc += zi*ri;
   }
   const double s = block_sum_256(acc, s4);
   if (threadIdx.x == 0) { partials[blockIdx.x] = s; }
}

// nom0 = nom = (d, r); r0; immediate-convergence / indefinite checks
// (linalg/solvers.cpp:895-928).
__global__ void __launch_bounds__(256)
k_pcg_init_scalars(const int nb, const double *__restrict__ partials, PcgState *st,
                   double *__restrict__ hist, const double *__restrict__ gsum)
{
   __shared__ double s4[4];
   const double nom = gsum ? *gsum : final_sum(partials, nb, s4);   // gsum: all-reduced sum
   if (threadIdx.x == 0)
   {
      st->nom = st->nom0 = st->betanom = nom;
      st->iter = 0; st->converged = 0; st->done = 0;
      st->hist_n = 1; st->k = 0; st->upd = -1;
      if (hist) { hist[0] = nom; }
      const double r0 = fmax(nom*st->rel_tol*st->rel_tol, st->abs_tol*st->abs_tol);
      st->r0 = r0;
      if (!(nom >= 0.0)) { st->done = 1; }                    // negative or NaN
      else if (nom <= r0) { st->converged = 1; st->done = 1; }
   }
}

// den = (d, A d) after the first/each operator application (linalg/solvers.cpp:931-949,
// 1010-1025); nom <- betanom at the end of an iteration.
__global__ void __launch_bounds__(256)
k_pcg_den_scalars(const int nb, const double *__restrict__ partials, PcgState *st, const int first,
                  const double *__restrict__ gsum)
{
   __shared__ double s4[4];
   if (st->done) { return; }
   const double den = gsum ? *gsum : final_sum(partials, nb, s4);
   if (threadIdx.x == 0) { pcg_den_step(st, den, first); }
}

// betanom, convergence test, iteration count, beta (linalg/solvers.cpp:963-1000).  One thread.
__device__ __forceinline__ void pcg_beta_step(PcgState *st, const double betanom, double *__restrict__ hist)
{
   const int i = st->iter + 1;        // the iteration whose update just ran
   st->betanom = betanom;
   st->upd = st->k;
   // a negative or NaN (B r, r) ends the solve BEFORE the reference monitors it (linalg/solvers.cpp:969-978: the break
   // precedes the print / Monitor call), so it is not part of the history
   if (!(betanom >= 0.0)) { st->iter = i; st->done = 1; return; }
   st->hist_n = i + 1;
   if (hist) { hist[i] = betanom; }
   if (betanom <= st->r0) { st->iter = i; st->converged = 1; st->done = 1; }
   else if (i + 1 > st->max_iter) { st->iter = st->max_iter; st->done = 1; }
   else { st->iter = i; st->beta = betanom/st->nom; }
}

// kernel A: r -= alpha z; z = dinv.r (or r); partial (r, z).  (x += alpha d rides with kernel B, which reads d
// anyway: one pass over d less per iteration, the same x bit for bit.)
// NT (large solves, pcg_nt): the vectors that are touched once per iteration (r, dinv here; x in kernel B) go past
// the caches, so that the two the next kernels read again -- z, then d -- are what the 256 MiB Infinity Cache holds
template <bool NT> __device__ __forceinline__ double ld_stream(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st_stream(double *p, const double v)
{
   if (NT) { __builtin_nontemporal_store(v, p); }
   else { *p = v; }
}

template <bool NT>
__global__ void __launch_bounds__(256)
k_pcg_update(const int64_t n, const PcgState *__restrict__ st, const double *__restrict__ dinv,
             double *__restrict__ r, double *__restrict__ z, double *__restrict__ partials)
{
   __shared__ double s4[4];
   if (st->done) { return; }
   const double alpha = st->alpha;
   double acc = 0.0;
   // U grid strides per trip, every load of the trip requested before the first value is used (the plain loop keeps one
   // element per thread in flight: load, wait, store).  Same elements per thread in the same order: same partial sums.
   constexpr int U = MFA_PCG_UNROLL;
   const int64_t stride = (int64_t)gridDim.x*256;
   int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   for (; i + (U - 1)*stride < n; i += U*stride)
   {
      double rr[U], zz[U], dd[U];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         rr[u] = ld_stream<NT>(&r[i + u*stride]); zz[u] = z[i + u*stride];
         dd[u] = dinv ? ld_stream<NT>(&dinv[i + u*stride]) : 1.0;
      }
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         const double ri = rr[u] + (-alpha)*zz[u];
         const double zi = dinv ? dd[u]*ri : ri;
         st_stream<NT>(&r[i + u*stride], ri); z[i + u*stride] = zi;
         acc += ri*zi;
      }
   }
   for (; i < n; i += stride)
   {
      const double ri = ld_stream<NT>(&r[i]) + (-alpha)*z[i];
      const double zi = dinv ? ld_stream<NT>(&dinv[i])*ri : ri;
      st_stream<NT>(&r[i], ri); z[i] = zi;
      acc += ri*zi;
   }
   const double s = block_sum_256(acc, s4);
   if (threadIdx.x == 0) { partials[blockIdx.x] = s; }
}

__global__ void __launch_bounds__(256)
k_pcg_beta_scalars(const int nb, const double *__restrict__ partials, PcgState *st,
                   double *__restrict__ hist, const double *__restrict__ gsum)
{
   __shared__ double s4[4];
   if (threadIdx.x == 0) { st->k = st->k + 1; }        // counts every launch: kernel B tells its iteration by it
   if (st->done) { return; }
   const double betanom = gsum ? *gsum : final_sum(partials, nb, s4);
   if (threadIdx.x == 0) { pcg_beta_step(st, betanom, hist); }
}

// kernel B: x += alpha d (the update of the iteration whose kernel A just ran: the reference updates x before it
// tests convergence, linalg/solvers.cpp:957-958, so the last iteration's update is applied as well); then, unless
// the solve has ended, d = z + beta d.  Launches after the end change nothing.
template <bool NT>
__global__ void __launch_bounds__(256)
k_pcg_dir(const int64_t n, const PcgState *__restrict__ st, const double *__restrict__ z,
          double *__restrict__ d, double *__restrict__ x)
{
   const bool xup = st->k == st->upd;             // kernel A of this iteration ran (no-op launches after the end: k moved on)
   if (!xup) { return; }
   const double alpha = st->alpha, beta = st->beta;
   if (st->done)
   {
      for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
      {
         x[i] = x[i] + alpha*d[i];
      }
      return;
   }
   constexpr int U = MFA_PCG_UNROLL;                  // (as in kernel A)
   const int64_t stride = (int64_t)gridDim.x*256;
   int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   for (; i + (U - 1)*stride < n; i += U*stride)
   {
      double dd[U], xx[U], zz[U];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         dd[u] = d[i + u*stride]; xx[u] = ld_stream<NT>(&x[i + u*stride]); zz[u] = ld_stream<NT>(&z[i + u*stride]);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         st_stream<NT>(&x[i + u*stride], xx[u] + alpha*dd[u]);
         d[i + u*stride] = zz[u] + beta*dd[u];
      }
   }
   for (; i < n; i += stride)
   {
      const double di = d[i];
      st_stream<NT>(&x[i], ld_stream<NT>(&x[i]) + alpha*di);
      d[i] = ld_stream<NT>(&z[i]) + beta*di;          // (z is dead after this read: the next apply overwrites it)
   }
}

static inline unsigned stream_grid(int64_t n)
{
   int64_t nb = (n + 255)/256;
   if (nb > 256*8) { nb = 256*8; }        // 8 workgroups per CU, grid-stride the rest
   if (nb < 1) { nb = 1; }
   return (unsigned)nb;
}

} // namespace mfa

using namespace mfa;

extern "C" int mfem_amd_vec_axpby(int64_t n, double a, const double *x, double b,
                                  const double *y, double *z, void *stream)
{
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && x && z && (y || b == 0.0), "vec_axpby: bad arguments");
   k_axpby<<<dim3(stream_grid(n)), dim3(256), 0, as_stream(stream)>>>(n, a, x, b, b == 0.0 ? nullptr : y, z);
   MFA_CHECK_LAUNCH("k_axpby");
   return 0;
}

extern "C" int mfem_amd_vec_fill(int64_t n, double value, double *x, void *stream)
{
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && x, "vec_fill: bad arguments");
   k_fill<<<dim3(stream_grid(n)), dim3(256), 0, as_stream(stream)>>>(n, value, x);
   MFA_CHECK_LAUNCH("k_fill");
   return 0;
}

extern "C" int64_t mfem_amd_reduce_workspace_bytes(void)
{
   return (int64_t)sizeof(double)*RED_BLOCKS_MAX;
}

extern "C" int mfem_amd_vec_dot(int64_t n, const double *x, const double *y, double *result,
                                void *ws, void *stream)
{
   MFA_REQUIRE(n >= 0 && result && ws && (n == 0 || (x && y)), "vec_dot: bad arguments");
   hipStream_t s = as_stream(stream);
   const int nb = red_blocks(n);
   k_dot_partial<<<dim3(nb), dim3(256), 0, s>>>(n, x, y, (double*)ws);
   MFA_CHECK_LAUNCH("k_dot_partial");
   k_dot_final<<<dim3(1), dim3(256), 0, s>>>(nb, (const double*)ws, result);
   MFA_CHECK_LAUNCH("k_dot_final");
   return 0;
}

// ---- mfem_amd_vec_dot for callers that cannot allocate ------------------------------------------------------
// The partial sums of the reduction live in a file-scope device array instead of a caller-provided workspace.
// Every GPU holds its own copy of the array at its own address: the address is looked up once per device, with
// that device current (mfem_amd_init, or the first call on the device), and only that pointer is ever handed to
// kernels -- the array's name is a host-side stand-in in host code, not a device address.
static __device__ __align__(256) double scratch[RED_BLOCKS_MAX];     // 64 KiB per GPU, held until the process ends
static double *g_scratch_ptr[64] = {nullptr};

namespace mfa
{
int dot_scratch_setup(int dev)
{
   if (dev < 0 || dev >= 64) { set_error("vec_dot_noalloc: device index %d out of range [0, 64)", dev); return 1; }
   if (g_scratch_ptr[dev]) { return 0; }
   int cur = -1;
   if (int rc = check_hip(hipGetDevice(&cur), "hipGetDevice")) { return rc; }
   if (cur != dev) { set_error("vec_dot_noalloc: device %d must be current (current: %d)", dev, cur); return 1; }
   double *p = nullptr;
   if (int rc = check_hip(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)")) { return rc; }
   if (!p) { set_error("vec_dot_noalloc: hipGetSymbolAddress returned a null address"); return 1; }
   g_scratch_ptr[dev] = p;
   return 0;
}
}

extern "C" int mfem_amd_vec_dot_noalloc(int64_t n, const double *x, const double *y, double *result, void *stream)
{
   MFA_REQUIRE(n >= 0 && result && (n == 0 || (x && y)), "vec_dot_noalloc: bad arguments");
   MFA_REQUIRE(n <= MFEM_AMD_DOT_NOALLOC_MAX_N, "vec_dot_noalloc: n = %lld exceeds the maximum of %lld entries",
               (long long)n, (long long)MFEM_AMD_DOT_NOALLOC_MAX_N);
   int dev = -1;
   if (int rc = check_hip(hipGetDevice(&dev), "hipGetDevice")) { return rc; }
   if (int rc = dot_scratch_setup(dev)) { return rc; }
   double *part = g_scratch_ptr[dev];
   hipStream_t s = as_stream(stream);
   // the same two kernels, grid and summation order as mfem_amd_vec_dot: bit-identical result.  Every partial
   // that the final pass reads (nb of them) is written by this call first, so nothing has to start at zero.
   const int nb = red_blocks(n);
   k_dot_partial<<<dim3(nb), dim3(256), 0, s>>>(n, x, y, part);
   MFA_CHECK_LAUNCH("k_dot_partial");
   k_dot_final<<<dim3(1), dim3(256), 0, s>>>(nb, part, result);
   MFA_CHECK_LAUNCH("k_dot_final");
   return 0;
}

extern "C" int mfem_amd_jacobi_setup(int64_t n, const double *diag, double damping,
                                     int64_t ness, const int *ess, double *dinv, void *stream)
{
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && diag && dinv && (ness == 0 || ess), "jacobi_setup: bad arguments");
   k_jacobi_setup<<<dim3(stream_grid(n)), dim3(256), 0, as_stream(stream)>>>(n, diag, damping, dinv);
   MFA_CHECK_LAUNCH("k_jacobi_setup");
   if (ness > 0) { return mfem_amd_set_subvector(ness, ess, nullptr, damping, dinv, stream); }
   return 0;
}

// y[vdof(i,c)] = scale[c]*x[i]: a scalar-space vector spread over the components of a vector space
struct RepScale { double s[16]; };
__global__ void __launch_bounds__(256)
k_replicate_vdim(const int64_t ndofs, const int vd, const int byvdim, const RepScale sc,
                 const double *__restrict__ x, double *__restrict__ y)
{
   const int64_t n = ndofs*vd;
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < n; i += (int64_t)gridDim.x*256)
   {
      const int64_t dof = byvdim ? i/vd : i % ndofs;
      const int c = (int)(byvdim ? i % vd : i/ndofs);
      y[i] = sc.s[c]*x[dof];
   }
}

extern "C" int mfem_amd_vec_replicate_vdim(int64_t ndofs, int vdim, int byvdim, const double *scale,
                                           const double *x, double *y, void *stream)
{
   if (ndofs == 0) { return 0; }
   MFA_REQUIRE(ndofs > 0 && vdim >= 1 && vdim <= 16 && x && y, "vec_replicate_vdim: bad arguments (vdim must be in 1..16)");
   RepScale sc;
   for (int c = 0; c < 16; c++) { sc.s[c] = (scale && c < vdim) ? scale[c] : 1.0; }
   k_replicate_vdim<<<dim3(stream_grid(ndofs*vdim)), dim3(256), 0, as_stream(stream)>>>(ndofs, vdim, byvdim ? 1 : 0, sc, x, y);
   MFA_CHECK_LAUNCH("k_replicate_vdim");
   return 0;
}

extern "C" int mfem_amd_jacobi_apply(int64_t n, const double *dinv, const double *r, double *z,
                                     void *stream)
{
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && dinv && r && z, "jacobi_apply: bad arguments");
   k_mul<<<dim3(stream_grid(n)), dim3(256), 0, as_stream(stream)>>>(n, dinv, r, z);
   MFA_CHECK_LAUNCH("k_mul");
   return 0;
}

// y = A x for the constrained PA operator; optionally the partial sums of (w, y).
// With a communicator this is RAPOperator(P, A, P) (linalg/operator.cpp:433) on vectors that
// live on the local L-dofs with non-owned (ghost) entries kept at zero: forward = P fills the
// ghost entries of x from their owners, reverse = P^T adds the ghost entries of y into the
// owners and zeroes them.  The (w, y) partials are taken BEFORE the reverse exchange, over
// all local dofs: sum_ranks (P w)^T A_L (P w) = w^T P^T A P w.
static bool pcg_nt(int64_t n);

// The element kernel over the elements [e0, e1) of a scalar space (e0 on a batch boundary of the blocked layout): the
// tables, pa_data, the (packed) E-vector and the direct rows of that range.
static int apply_range(const mfem_amd_pa_operator *op, const int *gm, const double *x, double *y, int64_t nd, int64_t nrows,
                       int64_t e0, int64_t e1, hipStream_t s)
{
   if (e1 <= e0) { return 0; }
   const int D = op->D1D, Q = op->Q1D;
   const int64_t D3 = (int64_t)D*D*D, NI = (int64_t)(D - 2)*(D - 2)*(D - 2), Q3 = (int64_t)Q*Q*Q;
   const int64_t pa_off = op->pa_layout ? blocked_size(op->kind, Q, e0) : e0*(op->kind == 2 ? 9 : (op->kind ? 6 : 1))*Q3;
   return pa_apply_L2E(op->kind, op->pa_layout, D, Q, e1 - e0, op->B, op->G, op->pa_data + pa_off, gm + e0*D3, x,
                       op->yE + e0*(nd ? D3 - NI : D3), s, nd ? y + nrows + e0*NI : nullptr);
}

// a second stream and four events per device for the exchanges that run beside the kernels (created on first use)
struct OverlapObjects { hipStream_t side; hipEvent_t x_ready, ghosts_in, ghost_rows_summed, sent; };
static int overlap_objects(OverlapObjects *oo)
{
   static OverlapObjects obj[64] = {};
   int dev = 0;
   if (int rc = check_hip(hipGetDevice(&dev), "hipGetDevice")) { return rc; }
   if (dev < 0 || dev >= 64) { set_error("pa operator: device index %d out of range", dev); return 1; }
   if (!obj[dev].side)
   {
      OverlapObjects o = {};
      if (int rc = check_hip(hipStreamCreateWithFlags(&o.side, hipStreamNonBlocking), "hipStreamCreateWithFlags")) { return rc; }
      for (hipEvent_t *e : {&o.x_ready, &o.ghosts_in, &o.ghost_rows_summed, &o.sent})
      {
         if (int rc = check_hip(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreateWithFlags")) { return rc; }
      }
      obj[dev] = o;
   }
   *oo = obj[dev];
   return 0;
}

static int constrained_mult(const mfem_amd_pa_operator *op, const mfem_amd_comm *comm,
                            double *x, double *y, const double *w, double *partials, int nb,
                            const int *skip, hipStream_t s)
{
   const TraceRange range(comm ? "mfem_amd: P^T A P apply" : "mfem_amd: constrained PA apply");
   const bool ess = op->ness > 0;
   MFA_REQUIRE(!ess || op->ess_marker, "pa operator: ess_marker is NULL");
   MFA_REQUIRE(op->yE && op->offsets && op->indices && op->pa_data, "pa operator: null array");
   if (op->vdim > 0)
   {
      // vector space: the essential marker is applied in the gather, rows are vdofs
      MFA_REQUIRE(op->gather_map, "pa operator: gather_map is NULL");
      if (comm)
      {
         if (int rc = comm->forward(comm->ctx, x, (void*)s)) { set_error("comm forward exchange failed (%d)", rc); return rc; }
      }
      const VecArgs va = {op->vdim, op->byvdim ? 1 : 0, op->ndofs, ess ? op->ess_marker : nullptr, op->pa_comp_stride, 0};
      if (int rc = pa_apply_L2E_vec(op->kind, op->pa_layout, op->D1D, op->Q1D, op->ne, op->B, op->G, op->pa_data,
                                    op->gather_map, x, op->yE, va, s)) { return rc; }
      if (int rc = scatter_dot_vec(op->ndofs, op->D1D*op->D1D*op->D1D, op->vdim, op->byvdim ? 1 : 0, op->offsets, op->indices,
                                   op->yE, y, ess ? op->ess_marker : nullptr, x, w, partials, nb, skip, s)) { return rc; }
      if (comm)
      {
         if (int rc = comm->reverse(comm->ctx, y, (void*)s)) { set_error("comm reverse exchange failed (%d)", rc); return rc; }
      }
      return 0;
   }
   const int *gm = ess ? op->masked_gather_map : op->gather_map;
   MFA_REQUIRE(gm, "pa operator: %s is NULL", ess ? "masked_gather_map" : "gather_map");
   // With a communicator and the local elements listed interior first (op->e_split of them read no ghost entry of x), the
   // owner -> ghost exchange runs on a second stream while the element kernel works on those elements; the interface
   // elements wait for its event.  (The reference overlaps the local copy only, fem/pfespace.cpp:5441.)  The interior range
   // is launched BEFORE the exchange is called: a transport that blocks the host still finds the device busy.
   static const bool overlap_ok = !(getenv("MFEM_AMD_OVERLAP") && atoi(getenv("MFEM_AMD_OVERLAP")) == 0);
   const bool overlap = comm && overlap_ok && op->e_split > 0 && op->e_split < op->ne;
   MFA_REQUIRE(op->e_split >= 0 && op->e_split <= op->ne, "pa operator: e_split out of range");
   MFA_REQUIRE(!overlap || !op->pa_layout || op->e_split % nbz_for(op->kind, op->Q1D) == 0,
               "pa operator: e_split must be a multiple of the blocked layout's batch (%d elements)", nbz_for(op->kind, op->Q1D));
   if (comm && !overlap)
   {
      if (int rc = comm->forward(comm->ctx, x, (void*)s)) { set_error("comm forward exchange failed (%d)", rc); return rc; }
   }
   // NOTE: the element kernel is not gated by `skip`: launched after convergence it rewrites the work E-vector and, with
   // direct rows, the element-interior rows y[ndofs - direct_rows, ndofs) -- the gated transpose-gather leaves the
   // rest of y untouched.  `skip` is only passed by the solver, whose y is its work vector z (dead after the end: x has
   // had its last update); a caller that wants y preserved after `skip` must not enable direct rows.
   // Element-interior rows (the last direct_rows ones, mfem_amd_interior_rows) are written by the element kernel itself, the
   // rest of the element goes to the packed E-vector, and the transpose-gather sums the rows before them.  With the packed
   // form this pays wherever the property holds (profiles/r3/direct_policy.txt, C++ drivers on the reference's numbering:
   // BP3 p = 2 / 3 +4 / +7 %, p = 4..8 +14..21 %, 1e6 dofs +9 %, 1e8 dofs +16 %, BP1 +8..40 %, inside the solver too) --
   // round 2's holed layout paid at p >= 4 below 3.3e7 dofs only.  MFEM_AMD_DIRECT=0 never (the plain two-kernel form).
   const char *pol = getenv("MFEM_AMD_DIRECT");
   const int policy = pol ? atoi(pol) : 1;
   const bool pays = policy != 0;
   // (the E-vector is then the packed one, read through op->packed_indices: no direct rows without that table)
   const int64_t nd = (op->direct_rows > 0 && op->packed_indices && op->kind != 2 && pays) ? op->direct_rows : 0;
   MFA_REQUIRE(nd >= 0 && nd <= op->ndofs, "pa operator: direct_rows out of range");
   const int64_t nrows = op->ndofs - nd;
   if (nd && !w) { nb = (int)((nrows + 1023)/1024 > 0 ? (nrows + 1023)/1024 : 1); }
   // The other direction: with the rank's ghost rows listed (op->reverse_overlap) and a transport that offers the two
   // halves of P^T, those rows are summed first and travel to their owners on the second stream while the
   // transpose-gather sums all rows (it writes the same bits into the ghost rows again); the owners add what arrived
   // once both are done.  The full pass is launched BEFORE the transport is called, as above.
   static const bool reverse_ok = !(getenv("MFEM_AMD_OVERLAP_REVERSE") && atoi(getenv("MFEM_AMD_OVERLAP_REVERSE")) == 0);
   const bool rsplit = comm && overlap_ok && reverse_ok && op->reverse_overlap && comm->reverse_begin && comm->reverse_end;
   MFA_REQUIRE(!rsplit || (op->n_ghost_rows >= 0 && (op->ghost_rows || op->n_ghost_rows == 0)), "pa operator: reverse_overlap without ghost_rows");
   OverlapObjects oo = {};
   if (overlap || rsplit) { if (int rc = overlap_objects(&oo)) { return rc; } }
   if (overlap)
   {
      if (int rc = check_hip(hipEventRecord(oo.x_ready, s), "hipEventRecord")) { return rc; }
      if (int rc = check_hip(hipStreamWaitEvent(oo.side, oo.x_ready, 0), "hipStreamWaitEvent")) { return rc; }
      if (int rc = apply_range(op, gm, x, y, nd, nrows, 0, op->e_split, s)) { return rc; }
      if (int rc = comm->forward(comm->ctx, x, (void*)oo.side)) { set_error("comm forward exchange failed (%d)", rc); return rc; }
      if (int rc = check_hip(hipEventRecord(oo.ghosts_in, oo.side), "hipEventRecord")) { return rc; }
      if (int rc = check_hip(hipStreamWaitEvent(s, oo.ghosts_in, 0), "hipStreamWaitEvent")) { return rc; }
      if (int rc = apply_range(op, gm, x, y, nd, nrows, op->e_split, op->ne, s)) { return rc; }
   }
   else if (int rc = apply_range(op, gm, x, y, nd, nrows, 0, op->ne, s)) { return rc; }
   const int *ind = nd ? op->packed_indices : op->indices;
   const unsigned char *mk = ess ? op->ess_marker : nullptr;
   if (rsplit)
   {
      if (int rc = scatter_list(op->n_ghost_rows, op->ghost_rows, nrows, op->offsets, ind, op->yE, y, mk, x, skip, s)) { return rc; }
      if (int rc = check_hip(hipEventRecord(oo.ghost_rows_summed, s), "hipEventRecord")) { return rc; }
   }
   if (int rc = scatter_dot(nrows, op->offsets, ind, op->yE, y, mk, x, w, partials, nb, skip, s, nd)) { return rc; }
   if (rsplit)
   {
      if (int rc = check_hip(hipStreamWaitEvent(oo.side, oo.ghost_rows_summed, 0), "hipStreamWaitEvent")) { return rc; }
      if (int rc = comm->reverse_begin(comm->ctx, y, (void*)oo.side)) { set_error("comm reverse exchange (begin) failed (%d)", rc); return rc; }
      if (int rc = check_hip(hipEventRecord(oo.sent, oo.side), "hipEventRecord")) { return rc; }
      if (int rc = check_hip(hipStreamWaitEvent(s, oo.sent, 0), "hipStreamWaitEvent")) { return rc; }
      if (int rc = comm->reverse_end(comm->ctx, y, (void*)s)) { set_error("comm reverse exchange (end) failed (%d)", rc); return rc; }
   }
   else if (comm)
   {
      if (int rc = comm->reverse(comm->ctx, y, (void*)s)) { set_error("comm reverse exchange failed (%d)", rc); return rc; }
   }
   return 0;
}

extern "C" int mfem_amd_pa_constrained_mult(const mfem_amd_pa_operator *op, const double *x,
                                            double *y, void *stream)
{
   MFA_REQUIRE(op && x && y, "pa_constrained_mult: bad arguments");
   MFA_REQUIRE(x != y, "pa_constrained_mult: x and y must not alias");
   const int64_t n = op->ndofs*(op->vdim > 0 ? op->vdim : 1);
   const unsigned nb = (unsigned)((n + 255)/256);
   if (n == 0) { return 0; }
   return constrained_mult(op, nullptr, const_cast<double*>(x), y, nullptr, nullptr, (int)nb, nullptr, as_stream(stream));
}

extern "C" int mfem_amd_pa_apply_L2L(int kind, int pa_layout, int D1D, int Q1D, int64_t ne, int64_t ndofs,
                                     const double *B, const double *G, const double *D, const int *gather_map,
                                     const int *offsets, const int *indices, const int *packed_indices,
                                     int64_t direct_rows, const double *x, double *yE, double *y, void *stream)
{
   MFA_REQUIRE(kind == 0 || kind == 1, "pa_apply_L2L: kind must be 0 (mass) or 1 (diffusion)");
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_apply_L2L: pa_layout must be 0 (MFEM) or 1 (blocked)");
   MFA_REQUIRE(ne >= 0 && ndofs >= 0 && direct_rows >= 0 && direct_rows <= ndofs, "pa_apply_L2L: bad sizes");
   MFA_REQUIRE(x && y && yE && x != y && offsets && indices && gather_map, "pa_apply_L2L: bad arguments");
   MFA_REQUIRE(direct_rows == 0 || (D1D >= 3 && direct_rows == ne*(int64_t)(D1D - 2)*(D1D - 2)*(D1D - 2)),
               "pa_apply_L2L: direct_rows must be 0 or NE*(D1D-2)^3 (mfem_amd_interior_rows)");
   MFA_REQUIRE(direct_rows == 0 || packed_indices, "pa_apply_L2L: direct_rows needs packed_indices (mfem_amd_pack_indices)");
   if (ndofs == 0) { return 0; }
   mfem_amd_pa_operator op = {};
   op.kind = kind; op.pa_layout = pa_layout; op.D1D = D1D; op.Q1D = Q1D; op.ne = ne; op.ndofs = ndofs;
   op.B = B; op.G = G; op.pa_data = D; op.gather_map = gather_map; op.offsets = offsets; op.indices = indices;
   op.yE = yE; op.direct_rows = direct_rows; op.packed_indices = packed_indices;
   const int64_t nb = (ndofs + 1023)/1024;
   return constrained_mult(&op, nullptr, const_cast<double*>(x), y, nullptr, nullptr, (int)nb, nullptr, as_stream(stream));
}

extern "C" int mfem_amd_pa_par_mult(const mfem_amd_pa_operator *op, const mfem_amd_comm *comm,
                                    double *x, double *y, void *stream)
{
   MFA_REQUIRE(op && comm && comm->forward && comm->reverse && x && y && x != y, "pa_par_mult: bad arguments");
   const unsigned nb = (unsigned)((op->ndofs + 255)/256);
   return constrained_mult(op, comm, x, y, nullptr, nullptr, (int)(nb ? nb : 1), nullptr, as_stream(stream));
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1)/a*a; }

extern "C" int64_t mfem_amd_pcg_workspace_bytes(int64_t n, int max_iter)
{
   // r, d, z | partials | hist | state | global-sum slot
   return 3*align_up(n*8, 256) + align_up(RED_BLOCKS_MAX*8, 256) +
          align_up(((int64_t)max_iter + 2)*8, 256) + 256 + 256;
}

// the streaming policy of the solver's vector kernels: MFEM_AMD_PCG_NT=0|1 forces it; by default on once the solver's
// six vectors (x, r, d, z, dinv, b) cannot all stay in the Infinity Cache anyway
static bool pcg_nt(int64_t n)
{
   const char *e = getenv("MFEM_AMD_PCG_NT");
   if (e && (e[0] == '0' || e[0] == '1')) { return e[0] == '1'; }
   return n*(int64_t)sizeof(double)*6 > ((int64_t)256 << 20);
}

static int pcg_impl(const mfem_amd_pa_operator *op, const mfem_amd_comm *comm, const double *dinv,
                    const double *b, double *x, double rel_tol, double abs_tol, int max_iter,
                    int check_every, void *workspace, mfem_amd_pcg_result *result, double *hist,
                    hipStream_t s)
{
   MFA_REQUIRE(op && b && x && workspace && result, "pcg_pa: bad arguments");
   MFA_REQUIRE(max_iter >= 0, "pcg_pa: max_iter < 0");
   MFA_REQUIRE(!comm || (comm->allreduce_sum && comm->forward && comm->reverse), "pcg_pa_par: incomplete communicator");
   if (check_every < 1) { check_every = 1; }
   const int64_t n = op->ndofs*(op->vdim > 0 ? op->vdim : 1);   // unknowns (vdofs of a vector space)
   char *ws = (char*)workspace;
   double *r = (double*)ws; ws += align_up(n*8, 256);
   double *d = (double*)ws; ws += align_up(n*8, 256);
   double *z = (double*)ws; ws += align_up(n*8, 256);
   double *partials = (double*)ws; ws += align_up(RED_BLOCKS_MAX*8, 256);
   double *dhist = (double*)ws; ws += align_up(((int64_t)max_iter + 2)*8, 256);
   PcgState *st = (PcgState*)ws; ws += 256;
   double *gsum = (double*)ws;
   const double *gs = comm ? gsum : nullptr;

   PcgState h = {};
   h.max_iter = max_iter; h.rel_tol = rel_tol; h.abs_tol = abs_tol;
   if (int rc = check_hip(hipMemcpyAsync(st, &h, sizeof h, hipMemcpyHostToDevice, s), "hipMemcpyAsync(state)")) { return rc; }

   const int nb = red_blocks(n);
   const unsigned sg = stream_grid(n);
   // local partials -> one local sum -> all-reduce (InnerProduct(comm,...), linalg/vector.hpp:773-779)
   auto global_sum = [&]() -> int
   {
      if (!comm) { return 0; }
      k_dot_final<<<dim3(1), dim3(256), 0, s>>>(nb, partials, gsum);
      if (int rc = check_hip(hipGetLastError(), "k_dot_final")) { return rc; }
      if (int rc = comm->allreduce_sum(comm->ctx, gsum, 1, (void*)s)) { set_error("comm all-reduce failed (%d)", rc); return rc; }
      return 0;
   };

   const TraceRange solve_range(comm ? "mfem_amd: Jacobi-PCG (partitioned)" : "mfem_amd: Jacobi-PCG");
   k_pcg_init<<<dim3(nb), dim3(256), 0, s>>>(n, b, dinv, x, r, d, partials);
   MFA_CHECK_LAUNCH("k_pcg_init");
   if (int rc = global_sum()) { return rc; }
   k_pcg_init_scalars<<<dim3(1), dim3(256), 0, s>>>(nb, partials, st, dhist, gs);
   MFA_CHECK_LAUNCH("k_pcg_init_scalars");
   // z = A d; den = (z, d)
   if (int rc = constrained_mult(op, comm, d, z, d, partials, nb, &st->done, s)) { return rc; }
   if (int rc = global_sum()) { return rc; }
   k_pcg_den_scalars<<<dim3(1), dim3(256), 0, s>>>(nb, partials, st, 1, gs);
   MFA_CHECK_LAUNCH("k_pcg_den_scalars");

   // The reference's loop body runs at least once, even for max_iter == 0.
   // (Finishing the two reductions inside the kernels that produce them, by the last workgroup to arrive, and
   // folding the scalar steps into the prologues of the kernels that consume them were both measured and
   // rejected: profiles/r2/pcg_fused_finish_probe.txt.)
   const int total = max_iter > 0 ? max_iter : 1;
   const bool nt = pcg_nt(n);
   PcgState cur;
   int launched = 0;
   while (true)
   {
      if (int rc = check_hip(hipMemcpyAsync(&cur, st, sizeof cur, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(poll)")) { return rc; }
      if (int rc = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize")) { return rc; }
      if (cur.done || launched >= total) { break; }
      int batch = check_every;
      if (batch > total - launched) { batch = total - launched; }
      const TraceRange batch_range("mfem_amd: PCG iterations (one host poll)");
      for (int k = 0; k < batch; k++)
      {
         if (nt) { k_pcg_update<true><<<dim3(nb), dim3(256), 0, s>>>(n, st, dinv, r, z, partials); }
         else { k_pcg_update<false><<<dim3(nb), dim3(256), 0, s>>>(n, st, dinv, r, z, partials); }
         if (int rc = global_sum()) { return rc; }
         k_pcg_beta_scalars<<<dim3(1), dim3(256), 0, s>>>(nb, partials, st, dhist, gs);
         if (nt) { k_pcg_dir<true><<<dim3(sg), dim3(256), 0, s>>>(n, st, z, d, x); }
         else { k_pcg_dir<false><<<dim3(sg), dim3(256), 0, s>>>(n, st, z, d, x); }
         MFA_CHECK_LAUNCH("pcg iteration kernels");
         if (int rc = constrained_mult(op, comm, d, z, d, partials, nb, &st->done, s)) { return rc; }
         if (int rc = global_sum()) { return rc; }
         k_pcg_den_scalars<<<dim3(1), dim3(256), 0, s>>>(nb, partials, st, 0, gs);
         MFA_CHECK_LAUNCH("k_pcg_den_scalars");
      }
      launched += batch;
   }
   // Reference bookkeeping (linalg/solvers.cpp:952-953,995-998): if the loop ran out of
   // iterations without converging, final_iter = max_iter.
   result->iterations = cur.done ? cur.iter : max_iter;
   result->converged = cur.converged;
   result->initial_nom = cur.nom0;
   result->final_nom = cur.betanom;
   // a breakdown (zero or NaN curvature) ends the solve one count past the last residual the reference prints
   result->history_count = cur.hist_n < result->iterations + 1 ? cur.hist_n : result->iterations + 1;
   if (hist)
   {
      const int cnt = result->history_count;
      if (int rc = check_hip(hipMemcpyAsync(hist, dhist, sizeof(double)*cnt, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(hist)")) { return rc; }
      if (int rc = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize")) { return rc; }
   }
   return 0;
}

extern "C" int mfem_amd_pcg_pa(const mfem_amd_pa_operator *op, const double *dinv,
                               const double *b, double *x, double rel_tol, double abs_tol,
                               int max_iter, int check_every, void *workspace,
                               mfem_amd_pcg_result *result, double *hist, void *stream)
{
   return pcg_impl(op, nullptr, dinv, b, x, rel_tol, abs_tol, max_iter, check_every, workspace, result,
                   hist, as_stream(stream));
}

extern "C" int mfem_amd_pcg_pa_par(const mfem_amd_pa_operator *op, const mfem_amd_comm *comm,
                                   const double *dinv, const double *b, double *x, double rel_tol,
                                   double abs_tol, int max_iter, int check_every, void *workspace,
                                   mfem_amd_pcg_result *result, double *hist, void *stream)
{
   MFA_REQUIRE(comm, "pcg_pa_par: comm is NULL");
   return pcg_impl(op, comm, dinv, b, x, rel_tol, abs_tol, max_iter, check_every, workspace, result,
                   hist, as_stream(stream));
}

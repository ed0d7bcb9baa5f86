// ElementRestriction L<->E maps and the ConstrainedOperator index kernels (gfx950).
//
// Replaces fem/restriction.cpp:109-129 (Mult), :152-180 (TAddMultTranspose<ADD>),
// :196-221 (AbsMultTranspose) and the index lambdas of linalg/operator.cpp:567-629.
// These are pure HBM-bound index streams: one thread per entry, coalesced int32 index
// reads, indexed fp64 reads served by L2 / Infinity Cache.  The E->L direction is the
// reference's deterministic transpose-offset gather (sequential left-to-right sum from 0
// over indices[offsets[i]..offsets[i+1])), so results are bit-reproducible -- no atomics.
#include "common.h"
#include <cstdlib>

namespace mfa
{

__global__ void __launch_bounds__(256)
k_gather(const int64_t n, const int *__restrict__ gmap, const double *__restrict__ x,
         double *__restrict__ yE)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int g = gmap[i];
   const double v = x[g >= 0 ? g : -1 - g];
   yE[i] = g >= 0 ? v : -v;
}

__global__ void __launch_bounds__(256)
k_gather_vdim(const int64_t n, const int nd, const int vd, const int byvdim,
              const int64_t ndofs, const int *__restrict__ gmap,
              const double *__restrict__ x, double *__restrict__ yE)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int g = gmap[i];
   const int64_t j = g >= 0 ? g : -1 - g;
   const int64_t e = i/nd, l = i % nd;
   for (int c = 0; c < vd; c++)
   {
      const double v = byvdim ? x[c + (int64_t)vd*j] : x[j + c*ndofs];
      yE[l + nd*(c + (int64_t)vd*e)] = g >= 0 ? v : -v;
   }
}

// MODE 0: y = sum, 1: y += sum, 2: y = sum of absolute-index entries (no signs).
// DOT: additionally accumulate sum_i w[i]*y[i] into per-block partials (CG: d.Ad folded
// into the transpose-gather epilogue).  marker/xin: rows with marker[i] != 0 take xin[i]
// (ConstrainedOperator y[ess] = x[ess], linalg/operator.cpp:624-629).  skip: optional
// device flag; non-zero -> whole kernel is a no-op (converged PCG).
//
// The kernel is bound by the latency of its dependent loads (offsets -> indices -> values),
// not by bytes, so every thread works on several rows at once, 256 rows apart (each load
// instruction stays coalesced across the wave): all offsets first, then the first NB index
// entries of every row, then their values -- independent loads in flight together -- before
// anything is added.  The additions stay sequential and left to right per row,
// starting from 0: the reference's order, bit for bit.
// NB index/value pairs of each of SKK rows in flight per thread (NB*SKK = 16 either way: 62 VGPRs, 8 waves/SIMD);
// entries beyond NB (edges/vertices at high order) follow one by one.
template <int MODE, bool DOT, int NB, int SKK>
__device__ __forceinline__ void scatter_rows(const int64_t ndofs, const int *__restrict__ offsets,
                                             const int *__restrict__ indices, const double *__restrict__ xE,
                                             double *__restrict__ y, const unsigned char *__restrict__ marker,
                                             const double *__restrict__ xin, const double *__restrict__ w, double &prod,
                                             const int64_t bid)
{
   for (int64_t base = bid*(256*SKK); base < ndofs; base += (int64_t)gridDim.x*(256*SKK))
   {
      int o0[SKK], n[SKK], id[SKK][NB];
      double t[SKK][NB], v[SKK];
      const int64_t last = ndofs - 1;
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         const int64_t ic = i < ndofs ? i : last;          // clamped: loads stay unconditional
         o0[j] = offsets[ic];
         n[j] = offsets[ic + 1] - o0[j];
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
#pragma unroll
         for (int k = 0; k < NB; k++) { id[j][k] = n[j] > 0 ? indices[o0[j] + (k < n[j] ? k : n[j] - 1)] : 0; }
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
#pragma unroll
         for (int k = 0; k < NB; k++) { t[j][k] = xE[id[j][k] >= 0 ? id[j][k] : -1 - id[j][k]]; }
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         double a = 0.0;
#pragma unroll
         for (int k = 0; k < NB; k++)
         {
            if (k < n[j]) { a += (id[j][k] >= 0 || MODE == 2) ? t[j][k] : -t[j][k]; }
         }
         for (int k = NB; k < n[j]; k++)
         {
            const int idx = indices[o0[j] + k];
            const double tt = xE[idx >= 0 ? idx : -1 - idx];
            a += (idx >= 0 || MODE == 2) ? tt : -tt;
         }
         v[j] = a;
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         if (i < ndofs)
         {
            double r = v[j];
            if (marker)
            {
               const unsigned char m = marker[i];        // 1: essential (owned) -> x; 2: essential ghost -> 0
               if (m) { r = (m == 1) ? xin[i] : 0.0; }
            }
            if (MODE == 1) { r += y[i]; }
            y[i] = r;
            if (DOT) { prod += w[i]*r; }
         }
      }
   }
}

// Block-cooperative form of the same kernel.  The row-per-thread form
// above walks offsets -> indices -> values as three dependent loads per thread, and a thread's row lengths differ.
// Here a workgroup takes the SAME 1024 rows (thread t: rows t, t+256, ...: the same d.Ad partial sums, bit for
// bit), but reads the block's entries -- one contiguous run of `indices` -- one entry per thread and sweep, fully
// coalesced and balanced, gathers their values into LDS, and only then sums every row from LDS in the reference's
// order.  Blocks with more than CAP entries (vertex and edge rows of MFEM's numbering) go through in four
// 256-row parts; a part that still does not fit (valence > CAP/256) is summed straight from memory.
constexpr int SCAT_CAP = 3072;
template <int MODE, bool DOT, bool NT = false>
__device__ __forceinline__ void scatter_block_rows(const int64_t ndofs, const int *__restrict__ offsets,
                                                   const int *__restrict__ indices, const double *__restrict__ xE,
                                                   double *__restrict__ y, const unsigned char *__restrict__ marker,
                                                   const double *__restrict__ xin, const double *__restrict__ w,
                                                   double &prod, double *__restrict__ sval, const int64_t bid)
{
   constexpr int CAP = SCAT_CAP, K = CAP/256;
   const int t = threadIdx.x;
   // entries [e_lo, e_lo + cnt) of `indices` -> signed values in sval[0, cnt)   (cnt <= CAP, uniform)
   auto fill = [&](const int e_lo, const int cnt)
   {
      int id[K];
      double v[K];
#pragma unroll
      for (int k = 0; k < K; k++)
      {
         if (256*k < cnt)
         {
            const int e = t + 256*k;
            const int *ip = &indices[e_lo + (e < cnt ? e : cnt - 1)];
            id[k] = NT ? __builtin_nontemporal_load(ip) : *ip;
         }
      }
#pragma unroll
      for (int k = 0; k < K; k++)
      {
         if (256*k < cnt)
         {
#if defined(MFA_E2L_PROBE) && MFA_E2L_PROBE == 1          // probe builds only (tools/r3_job17.sh): no gather at all
            v[k] = (double)id[k];
#elif defined(MFA_E2L_PROBE) && MFA_E2L_PROBE == 2        // ... a gather that always hits (a 32 KB window of the E-vector)
            v[k] = xE[(id[k] >= 0 ? id[k] : -1 - id[k]) & 4095];
#else
            v[k] = xE[id[k] >= 0 ? id[k] : -1 - id[k]];       // re-used across the four sweeps of the numbering: cached
#endif
         }
      }
#pragma unroll
      for (int k = 0; k < K; k++)
      {
         const int e = t + 256*k;
         if (256*k < cnt && e < cnt) { sval[e] = (id[k] >= 0 || MODE == 2) ? v[k] : -v[k]; }
      }
   };
   for (int64_t base = bid*1024; base < ndofs; base += (int64_t)gridDim.x*1024)
   {
      const int64_t last = ndofs - 1;
      int o0[4], n[4];
      double val[4];
      // the epilogue's operands (essential marker, d of the d.Ad sum) are requested now, so that they arrive
      // while the entries are gathered instead of after the row sums
      unsigned char mk[4];
      double wv[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
         const int64_t i = base + t + j*256;
         const int64_t ic = i < ndofs ? i : last;
         o0[j] = offsets[ic];
         n[j] = i < ndofs ? offsets[ic + 1] - o0[j] : 0;
         mk[j] = marker ? marker[ic] : (unsigned char)0;
         wv[j] = DOT ? w[ic] : 0.0;
      }
      const int64_t rend = base + 1024 < ndofs ? base + 1024 : ndofs;
      const int o_lo = offsets[base], nent = offsets[rend] - o_lo;
      if (nent <= CAP)
      {
         if (nent > 0) { fill(o_lo, nent); }
         __syncthreads();
#pragma unroll
         for (int j = 0; j < 4; j++)
         {
            double a = 0.0;
            const double *sv = sval + (o0[j] - o_lo);
            for (int k = 0; k < n[j]; k++) { a += sv[k]; }
            val[j] = a;
         }
         __syncthreads();
      }
      else
      {
#pragma unroll
         for (int j = 0; j < 4; j++)
         {
            const int64_t r0 = base + j*256;                               // uniform
            const int64_t r1 = r0 + 256 < ndofs ? r0 + 256 : ndofs;
            const int p_lo = r0 < ndofs ? offsets[r0] : 0, pcnt = r0 < ndofs ? offsets[r1] - p_lo : 0;
            double a = 0.0;
            if (pcnt <= CAP)
            {
               if (pcnt > 0) { fill(p_lo, pcnt); }
               __syncthreads();
               const double *sv = sval + (o0[j] - p_lo);
               for (int k = 0; k < n[j]; k++) { a += sv[k]; }
               __syncthreads();
            }
            else
            {
               for (int k = 0; k < n[j]; k++)
               {
                  const int idx = indices[o0[j] + k];
                  const double tt = xE[idx >= 0 ? idx : -1 - idx];
                  a += (idx >= 0 || MODE == 2) ? tt : -tt;
               }
            }
            val[j] = a;
         }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
         const int64_t i = base + t + j*256;
         if (i < ndofs)
         {
            double r = val[j];
            if (mk[j]) { r = (mk[j] == 1) ? xin[i] : 0.0; }
            if (MODE == 1) { r += y[i]; }
            y[i] = r;      // cached: the caller's next kernel reads it (a non-temporal store measured no different)
            if (DOT) { prod += wv[j]*r; }
         }
      }
   }
}

template <int MODE, bool DOT>
__global__ void __launch_bounds__(256)
k_scatter(const int64_t ndofs, const int *__restrict__ offsets,
          const int *__restrict__ indices, const double *__restrict__ xE,
          double *__restrict__ y, const unsigned char *__restrict__ marker,
          const double *__restrict__ xin, const double *__restrict__ w,
          double *__restrict__ partials, const int *__restrict__ skip, const int flags, const int64_t ndirect)
{
   __shared__ double sval[SCAT_CAP];
   __shared__ double s4[4];
   if (skip && *skip) { return; }
   double prod = 0.0;
   // rows ndofs .. ndofs + ndirect - 1 were written by the element kernel (element-interior dofs): nothing to sum,
   // only their part of (w, y), streamed in a fixed thread order -- and the constraint: an element-interior dof can
   // be in the essential list too (a pinned dof, dofs of an element attribute), and its row then takes x (marker 1) or
   // 0 (marker 2) like any other row (ConstrainedOperator, linalg/operator.cpp:624-629)
   if ((DOT || marker) && ndirect > 0)
   {
      for (int64_t i = ndofs + (int64_t)blockIdx.x*256 + threadIdx.x; i < ndofs + ndirect; i += (int64_t)gridDim.x*256)
      {
         const unsigned char m = marker ? marker[i] : (unsigned char)0;
         double r = 0.0;
         if (m) { r = (m == 1) ? xin[i] : 0.0; y[i] = r; }
         else if (DOT) { r = y[i]; }
         if (DOT) { prod += w[i]*r; }
      }
   }
   // Wave-uniform choice by the table's own average row length nnz/ndofs (profiles/r1/e2l_rows_probe.txt,
   // profiles/r2/e2l_blk_probe.txt): the block-cooperative form below 6 entries per row (p >= 2); long rows
   // (p = 1: 8 entries) overflow its LDS tile and are faster row per thread, 8 entries of two rows in flight.
   const int64_t nnz = offsets[ndofs];
   // Logical block id XCD-contiguous (the partial sums stay indexed by the logical id: same bits): a tuning aid only
   // (MFEM_AMD_SCATTER_XCD=1).  On the tensor-grid numbering of the synthetic boxes it gained 2.7 % at p = 2 and was
   // adopted for 3..6 entries per row; on the reference's own numbering (C++ drivers) the same rule cost 27 % (BP3) and
   // 58 % (BP1) at p = 2 -- profiles/r2/xcd_probe.txt -- so nothing selects it by default any more.
   const bool xcd = (flags & 1) != 0;
   const int64_t bid = xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
   if (!(flags & 2) && nnz < 6*ndofs)
   {
      // streamed-once operands non-temporal when the tables cannot stay in the Infinity Cache between applications
      // anyway (profiles/r2/e2l_nt_probe.txt); small problems keep everything cached
      if ((flags & 4) && ndofs + ndirect > 3000000) { scatter_block_rows<MODE, DOT, true>(ndofs, offsets, indices, xE, y, marker, xin, w, prod, sval, bid); }
      else { scatter_block_rows<MODE, DOT>(ndofs, offsets, indices, xE, y, marker, xin, w, prod, sval, bid); }
   }
   else if (nnz >= 3*ndofs) { scatter_rows<MODE, DOT, 8, 2>(ndofs, offsets, indices, xE, y, marker, xin, w, prod, bid); }
   else if (20*nnz >= 31*ndofs) { scatter_rows<MODE, DOT, 4, 4>(ndofs, offsets, indices, xE, y, marker, xin, w, prod, bid); }
   else { scatter_rows<MODE, DOT, 2, 4>(ndofs, offsets, indices, xE, y, marker, xin, w, prod, bid); }
   if (DOT)
   {
      const double r = block_sum_256(prod, s4);
      if (threadIdx.x == 0) { partials[bid] = r; }
   }
}

// Chunk-pipelined E->L (fem/restriction.cpp:152-180 over element chunks): the element kernel runs over
// contiguous element chunks that write one slot of a small, re-used E-vector ring, and the rows are summed
// chunk by chunk while the chunk's E-vector part is still in the caches.  A sub-row is the part of an
// L-row whose entries lie in ONE chunk; sub-rows of a row are visited in chunk (= element) order, the first
// starts from 0, every later one from the partial sum the previous left in y -- the same additions in the
// same order as the reference's sequential left-to-right row sum, bit for bit.
//   rows[i] = L-row | SUB_CONT (continue from y[row]) | SUB_OPEN (more sub-rows follow: not the final value)
constexpr int SUB_CONT = 1 << 30;
constexpr unsigned SUB_OPEN = 1u << 31;
constexpr int SUB_ROW = SUB_CONT - 1;

template <bool DOT, int NB, int SKK>
__device__ __forceinline__ void scatter_subrows(const int64_t nsub, const int *__restrict__ rows,
                                                const int *__restrict__ offsets, const int *__restrict__ indices,
                                                const double *__restrict__ xE, double *__restrict__ y,
                                                const unsigned char *__restrict__ marker, const double *__restrict__ xin,
                                                const double *__restrict__ w, double &prod)
{
   for (int64_t base = (int64_t)blockIdx.x*(256*SKK); base < nsub; base += (int64_t)gridDim.x*(256*SKK))
   {
      int o0[SKK], n[SKK], id[SKK][NB], code[SKK];
      double t[SKK][NB], v[SKK];
      const int64_t last = nsub - 1;
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         const int64_t ic = i < nsub ? i : last;
         code[j] = rows[ic];
         o0[j] = offsets[ic];
         n[j] = offsets[ic + 1] - o0[j];
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
#pragma unroll
         for (int k = 0; k < NB; k++) { id[j][k] = n[j] > 0 ? indices[o0[j] + (k < n[j] ? k : n[j] - 1)] : 0; }
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         v[j] = 0.0;
         if (code[j] & SUB_CONT) { v[j] = y[code[j] & SUB_ROW]; }
#pragma unroll
         for (int k = 0; k < NB; k++) { t[j][k] = xE[id[j][k] >= 0 ? id[j][k] : -1 - id[j][k]]; }
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         double a = v[j];
#pragma unroll
         for (int k = 0; k < NB; k++)
         {
            if (k < n[j]) { a += id[j][k] >= 0 ? t[j][k] : -t[j][k]; }
         }
         for (int k = NB; k < n[j]; k++)
         {
            const int idx = indices[o0[j] + k];
            const double tt = xE[idx >= 0 ? idx : -1 - idx];
            a += idx >= 0 ? tt : -tt;
         }
         v[j] = a;
      }
#pragma unroll
      for (int j = 0; j < SKK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         if (i < nsub)
         {
            const int row = code[j] & SUB_ROW;
            const bool open = ((unsigned)code[j] & SUB_OPEN) != 0;
            double r = v[j];
            if (marker && !open)
            {
               const unsigned char m = marker[row];
               if (m) { r = (m == 1) ? xin[row] : 0.0; }
            }
            y[row] = r;
            if (DOT && !open) { prod += w[row]*r; }
         }
      }
   }
}

template <bool DOT>
__global__ void __launch_bounds__(256)
k_scatter_sub(const int64_t nsub, const int *__restrict__ rows, const int *__restrict__ offsets,
              const int *__restrict__ indices, const double *__restrict__ xE,
              double *__restrict__ y, const unsigned char *__restrict__ marker,
              const double *__restrict__ xin, const double *__restrict__ w,
              double *__restrict__ partials, const int *__restrict__ skip)
{
   __shared__ double s4[4];
   if (skip && *skip) { return; }
   double prod = 0.0;
   const int64_t nnz = offsets[nsub] - offsets[0];
   if (nnz >= 3*nsub) { scatter_subrows<DOT, 8, 2>(nsub, rows, offsets, indices, xE, y, marker, xin, w, prod); }
   else if (20*nnz >= 31*nsub) { scatter_subrows<DOT, 4, 4>(nsub, rows, offsets, indices, xE, y, marker, xin, w, prod); }
   else { scatter_subrows<DOT, 2, 4>(nsub, rows, offsets, indices, xE, y, marker, xin, w, prod); }
   if (DOT)
   {
      const double r = block_sum_256(prod, s4);
      if (threadIdx.x == 0) { partials[blockIdx.x] = r; }
   }
}

// Vector spaces (ElementRestriction with vdim > 1, fem/restriction.cpp:152-180): row i is a
// vdof -- byNODES i = c*ndofs + dof, byVDIM i = dof*vd + c -- and sums the entries
// xE[(idx % nd) + nd*(c + vd*(idx / nd))] of its scalar dof's CSR row, in the same order.
struct ScatVec { int vd, byvdim, nd; int64_t ndofs; };

// JB entries of the row [j0, j0+m) for CB components starting at c0: loads batched, sums in order
template <int MODE, int JB, int CB>
__device__ __forceinline__ void scatter_vec_chunk(const ScatVec &sv, const int j0, const int m, const int c0,
                                                  const int *__restrict__ indices, const double *__restrict__ xE,
                                                  double (&v)[CB])
{
   int id[JB];
   int64_t base[JB];
   double t[JB][CB];
#pragma unroll
   for (int q = 0; q < JB; q++) { id[q] = indices[j0 + (q < m ? q : m - 1)]; }
#pragma unroll
   for (int q = 0; q < JB; q++)
   {
      const int64_t a = id[q] >= 0 ? id[q] : -1 - id[q];
      base[q] = (a % sv.nd) + (int64_t)sv.nd*sv.vd*(a/sv.nd);
   }
#pragma unroll
   for (int q = 0; q < JB; q++)
   {
#pragma unroll
      for (int k = 0; k < CB; k++)
      {
         const int c = c0 + k < sv.vd ? c0 + k : sv.vd - 1;
         t[q][k] = xE[base[q] + (int64_t)sv.nd*c];
      }
   }
#pragma unroll
   for (int q = 0; q < JB; q++)
   {
      if (q < m)
      {
#pragma unroll
         for (int k = 0; k < CB; k++) { v[k] += (id[q] >= 0 || MODE == 2) ? t[q][k] : -t[q][k]; }
      }
   }
}

// NR scalar rows per thread, 256 apart; the first NB entries of every row are in flight together for CB = 3
// components (as in k_scatter); longer rows continue four entries at a time.
template <int MODE, bool DOT, int NB, int NR>
__device__ __forceinline__ void scatter_vec_rows(const ScatVec &sv, const int *__restrict__ offsets,
                                                 const int *__restrict__ indices, const double *__restrict__ xE,
                                                 double *__restrict__ y, const unsigned char *__restrict__ marker,
                                                 const double *__restrict__ xin, const double *__restrict__ w, double &prod)
{
   constexpr int CB = 3;
   const int vd = sv.vd;
   for (int64_t ib = (int64_t)blockIdx.x*(256*NR) + threadIdx.x; ib < sv.ndofs; ib += (int64_t)gridDim.x*(256*NR))
   {
      int o0[NR], n[NR];
#pragma unroll
      for (int r = 0; r < NR; r++)
      {
         const int64_t i = ib + r*256;
         const int64_t ic = i < sv.ndofs ? i : ib;
         o0[r] = offsets[ic];
         n[r] = i < sv.ndofs ? offsets[ic + 1] - o0[r] : 0;
      }
      for (int c0 = 0; c0 < vd; c0 += CB)
      {
         int id[NR][NB];
         int64_t base[NR][NB];
         double t[NR][NB][CB], v[NR][CB];
#pragma unroll
         for (int r = 0; r < NR; r++)
         {
#pragma unroll
            for (int q = 0; q < NB; q++) { id[r][q] = n[r] > 0 ? indices[o0[r] + (q < n[r] ? q : n[r] - 1)] : 0; }
         }
#pragma unroll
         for (int r = 0; r < NR; r++)
         {
#pragma unroll
            for (int q = 0; q < NB; q++)
            {
               const int64_t a = id[r][q] >= 0 ? id[r][q] : -1 - id[r][q];
               base[r][q] = (a % sv.nd) + (int64_t)sv.nd*sv.vd*(a/sv.nd);
            }
         }
#pragma unroll
         for (int r = 0; r < NR; r++)
         {
#pragma unroll
            for (int q = 0; q < NB; q++)
            {
#pragma unroll
               for (int k = 0; k < CB; k++)
               {
                  const int c = c0 + k < vd ? c0 + k : vd - 1;
                  t[r][q][k] = xE[base[r][q] + (int64_t)sv.nd*c];
               }
            }
         }
#pragma unroll
         for (int r = 0; r < NR; r++)
         {
#pragma unroll
            for (int k = 0; k < CB; k++) { v[r][k] = 0.0; }
#pragma unroll
            for (int q = 0; q < NB; q++)
            {
               if (q < n[r])
               {
#pragma unroll
                  for (int k = 0; k < CB; k++) { v[r][k] += (id[r][q] >= 0 || MODE == 2) ? t[r][q][k] : -t[r][q][k]; }
               }
            }
            for (int j0 = o0[r] + NB; j0 < o0[r] + n[r]; j0 += 4)
            {
               const int m = o0[r] + n[r] - j0;
               scatter_vec_chunk<MODE, 4, CB>(sv, j0, m < 4 ? m : 4, c0, indices, xE, v[r]);
            }
         }
#pragma unroll
         for (int r = 0; r < NR; r++)
         {
            const int64_t i = ib + r*256;
            if (i >= sv.ndofs) { continue; }
#pragma unroll
            for (int k = 0; k < CB; k++)
            {
               const int c = c0 + k;
               if (c < vd)
               {
                  const int64_t row = sv.byvdim ? i*vd + c : i + c*sv.ndofs;
                  double val = v[r][k];
                  if (marker)
                  {
                     const unsigned char mk = marker[row];
                     if (mk) { val = (mk == 1) ? xin[row] : 0.0; }
                  }
                  if (MODE == 1) { val += y[row]; }
                  y[row] = val;
                  if (DOT) { prod += w[row]*val; }
               }
            }
         }
      }
   }
}

template <int MODE, bool DOT>
__global__ void __launch_bounds__(256)
k_scatter_vec(const ScatVec sv, const int *__restrict__ offsets,
              const int *__restrict__ indices, const double *__restrict__ xE,
              double *__restrict__ y, const unsigned char *__restrict__ marker,
              const double *__restrict__ xin, const double *__restrict__ w,
              double *__restrict__ partials, const int *__restrict__ skip)
{
   // Threads own SCALAR dofs: a CSR row (offsets, indices, the / and % by nd) is read once for all
   // components, three components at a time (vdim = 3 is the case that matters).
   __shared__ double s4[4];
   if (skip && *skip) { return; }
   double prod = 0.0;
   // batch depth from the table's average row length, as in k_scatter (here the deep batch pays only at p = 1)
   if ((int64_t)offsets[sv.ndofs] >= 4*sv.ndofs) { scatter_vec_rows<MODE, DOT, 8, 1>(sv, offsets, indices, xE, y, marker, xin, w, prod); }
   else { scatter_vec_rows<MODE, DOT, 4, 2>(sv, offsets, indices, xE, y, marker, xin, w, prod); }
   if (DOT)
   {
      const double r = block_sum_256(prod, s4);
      if (threadIdx.x == 0) { partials[blockIdx.x] = r; }
   }
}

__global__ void __launch_bounds__(256)
k_set_marker(const int64_t n, const int *__restrict__ list, unsigned char *__restrict__ marker)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i < n) { marker[list[i]] = 1; }
}

__global__ void __launch_bounds__(256)
k_mask_gather(const int64_t n, const int *__restrict__ gmap,
              const unsigned char *__restrict__ marker, int *__restrict__ out)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int g = gmap[i];
   out[i] = marker[g] ? -1 : g;
}

__global__ void __launch_bounds__(256)
k_set_subvector(const int64_t n, const int *__restrict__ list, const double *__restrict__ x,
                const double value, double *__restrict__ y)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int k = list[i];
   y[k] = x ? x[k] : value;
}

// Shared-dof exchange kernels (DeviceConformingProlongationOperator, fem/pfespace.cpp:5347-5494)
__global__ void __launch_bounds__(256)
k_pack(const int64_t n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ buf)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i < n) { buf[i] = x[idx[i]]; }
}

__global__ void __launch_bounds__(256)
k_unpack(const int64_t n, const int *__restrict__ idx, const double *__restrict__ buf, double *__restrict__ y)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i < n) { y[idx[i]] = buf[i]; }
}

__global__ void __launch_bounds__(256)
k_accumulate(const int64_t n, const int *__restrict__ dst, const int *__restrict__ off,
             const int *__restrict__ src, const double *__restrict__ buf, double *__restrict__ y)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int k = dst[i];
   double sum = y[k];
   const int end = off[i + 1];
   for (int j = off[i]; j != end; ++j) { sum += buf[src[j]]; }
   y[k] = sum;
}

// Element-attribute markers of an integrator (PABilinearFormExtension::AddMultWithMarkers /
// AssembleDiagonal, fem/bilinearform_ext.cpp:753-774, 374-398): element e takes part when
// markers[attributes[e] - 1] != 0.  MODE 0: y(:,e) += x(:,e) on marked elements; MODE 1: d(:,e) = 0 on
// the others.  One thread per E-vector entry.
template <int MODE>
__global__ void __launch_bounds__(256)
k_evec_markers(const int64_t n, const int nd, const int nmarkers, const int *__restrict__ markers,
               const int *__restrict__ attributes, const double *__restrict__ x, double *__restrict__ y)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int attr = attributes[i/nd];
   const bool on = attr > 0 && attr <= nmarkers && markers[attr - 1] != 0;
   if (MODE == 0) { if (on) { y[i] += x[i]; } }
   else { if (!on) { y[i] = 0.0; } }
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255)/256); }

// tuning aids: MFEM_AMD_SCATTER_XCD=1 XCD-contiguous row blocks with the row-per-thread kernel (measured 2-5 %
// slower, profiles/r2/xcd_probe.txt); MFEM_AMD_SCATTER_BLK=0 row-per-thread kernel at every order
static int scatter_flags()
{
   static const int v = ((getenv("MFEM_AMD_SCATTER_XCD") && atoi(getenv("MFEM_AMD_SCATTER_XCD"))) ? 1 : 0) |
                        ((getenv("MFEM_AMD_SCATTER_BLK") && !atoi(getenv("MFEM_AMD_SCATTER_BLK"))) ? 2 : 0) |
                        ((getenv("MFEM_AMD_SCATTER_NT") && !atoi(getenv("MFEM_AMD_SCATTER_NT"))) ? 0 : 4);
   return v;
}

// E->L with the fused dot; nblocks partials are written (fixed by the caller from n only).
int scatter_dot(int64_t ndofs, const int *offsets, const int *indices, const double *xE,
                double *y, const unsigned char *marker, const double *xin, const double *w,
                double *partials, int nblocks, const int *skip, hipStream_t s, int64_t ndirect)
{
   if (w)
   {
      k_scatter<0, true><<<dim3(nblocks), dim3(256), 0, s>>>(ndofs, offsets, indices, xE, y,
                                                           marker, xin, w, partials, skip, scatter_flags(), ndirect);
   }
   else
   {
      k_scatter<0, false><<<dim3(nblocks), dim3(256), 0, s>>>(ndofs, offsets, indices, xE, y,
                                                            marker, xin, nullptr, nullptr, skip, scatter_flags(), marker ? ndirect : 0);
   }
   MFA_CHECK_LAUNCH("k_scatter");
   return 0;
}

// The rows of a list summed ahead of the full pass (partitioned runs: the ghost rows, whose sums travel to their owners
// while k_scatter works on all rows -- which writes the same bits into them again).  Same sequential left-to-right
// sums from 0, same essential-row patch.  Rows >= nrows (element-interior rows, written by the element kernel; never
// ghosts) are left alone.  The lists are short (the surface of a rank's part): one row per thread, four entries in flight.
__global__ void __launch_bounds__(256)
k_scatter_list(const int64_t nlist, const int *__restrict__ rows, const int64_t nrows, const int *__restrict__ offsets,
               const int *__restrict__ indices, const double *__restrict__ xE, double *__restrict__ y,
               const unsigned char *__restrict__ marker, const double *__restrict__ xin, const int *__restrict__ skip)
{
   if (skip && *skip) { return; }
   for (int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x; i < nlist; i += (int64_t)gridDim.x*256)
   {
      const int64_t row = rows[i];
      if (row < 0 || row >= nrows) { continue; }
      const int o0 = offsets[row], n = offsets[row + 1] - o0;
      int id[4];
      double t[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { id[k] = n > 0 ? indices[o0 + (k < n ? k : n - 1)] : 0; }
#pragma unroll
      for (int k = 0; k < 4; k++) { t[k] = n > 0 ? xE[id[k] >= 0 ? id[k] : -1 - id[k]] : 0.0; }
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; k++) { if (k < n) { a += id[k] >= 0 ? t[k] : -t[k]; } }
      for (int k = 4; k < n; k++)
      {
         const int idx = indices[o0 + k];
         const double tt = xE[idx >= 0 ? idx : -1 - idx];
         a += idx >= 0 ? tt : -tt;
      }
      if (marker)
      {
         const unsigned char m = marker[row];
         if (m) { a = (m == 1) ? xin[row] : 0.0; }
      }
      y[row] = a;
   }
}

int scatter_list(int64_t nlist, const int *rows, int64_t nrows, const int *offsets, const int *indices, const double *xE,
                 double *y, const unsigned char *marker, const double *xin, const int *skip, hipStream_t s)
{
   if (nlist <= 0) { return 0; }
   const int64_t nb = (nlist + 255)/256;
   k_scatter_list<<<dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, s>>>(nlist, rows, nrows, offsets, indices, xE, y,
                                                                                 marker, xin, skip);
   MFA_CHECK_LAUNCH("k_scatter_list");
   return 0;
}

// vector-space variant of scatter_dot (rows = vdofs)
int scatter_dot_vec(int64_t ndofs, int nd, int vd, int byvdim, const int *offsets, const int *indices,
                    const double *xE, double *y, const unsigned char *marker, const double *xin,
                    const double *w, double *partials, int nblocks, const int *skip, hipStream_t s)
{
   const ScatVec sv = {vd, byvdim, nd, ndofs};
   if (w)
   {
      k_scatter_vec<0, true><<<dim3(nblocks), dim3(256), 0, s>>>(sv, offsets, indices, xE, y, marker, xin, w, partials, skip);
   }
   else
   {
      k_scatter_vec<0, false><<<dim3(nblocks), dim3(256), 0, s>>>(sv, offsets, indices, xE, y, marker, xin, nullptr, nullptr, skip);
   }
   MFA_CHECK_LAUNCH("k_scatter_vec");
   return 0;
}

} // namespace mfa

using namespace mfa;

extern "C" int mfem_amd_restrict_scatter_vdim(int64_t ndofs, int nd, int vdim, int byvdim,
                                              const int *offsets, const int *indices,
                                              const double *xE, double *y, int add, int absv,
                                              void *stream)
{
   if (ndofs == 0) { return 0; }
   MFA_REQUIRE(ndofs > 0 && nd > 0 && vdim >= 1 && offsets && indices && xE && y, "restrict_scatter_vdim: bad arguments");
   MFA_REQUIRE(!(add && absv), "restrict_scatter_vdim: add and abs cannot be combined");
   MFA_REQUIRE(ndofs*vdim < ((int64_t)1 << 39), "restrict_scatter_vdim: too many vdofs");
   const ScatVec sv = {vdim, byvdim ? 1 : 0, nd, ndofs};
   const unsigned nb = (unsigned)((ndofs + 511)/512);         // two scalar dofs per thread
   hipStream_t s = as_stream(stream);
   if (absv) { k_scatter_vec<2, false><<<dim3(nb), dim3(256), 0, s>>>(sv, offsets, indices, xE, y, nullptr, nullptr, nullptr, nullptr, nullptr); }
   else if (add) { k_scatter_vec<1, false><<<dim3(nb), dim3(256), 0, s>>>(sv, offsets, indices, xE, y, nullptr, nullptr, nullptr, nullptr, nullptr); }
   else { k_scatter_vec<0, false><<<dim3(nb), dim3(256), 0, s>>>(sv, offsets, indices, xE, y, nullptr, nullptr, nullptr, nullptr, nullptr); }
   MFA_CHECK_LAUNCH("k_scatter_vec");
   return 0;
}

extern "C" int mfem_amd_restrict_gather(int64_t nedofs, const int *gather_map,
                                        const double *x, double *yE, void *stream)
{
   if (nedofs == 0) { return 0; }
   MFA_REQUIRE(nedofs > 0 && gather_map && x && yE, "restrict_gather: bad arguments");
   MFA_REQUIRE(nedofs < ((int64_t)1 << 39), "restrict_gather: nedofs too large");
   k_gather<<<dim3(grid_for(nedofs)), dim3(256), 0, as_stream(stream)>>>(nedofs, gather_map, x, yE);
   MFA_CHECK_LAUNCH("k_gather");
   return 0;
}

extern "C" int mfem_amd_restrict_gather_vdim(int64_t nedofs, int nd, int vd, int byvdim,
                                             int64_t ndofs, const int *gather_map,
                                             const double *x, double *yE, void *stream)
{
   if (nedofs == 0) { return 0; }
   MFA_REQUIRE(nedofs > 0 && nd > 0 && vd > 0 && nedofs % nd == 0 && gather_map && x && yE,
               "restrict_gather_vdim: bad arguments");
   k_gather_vdim<<<dim3(grid_for(nedofs)), dim3(256), 0, as_stream(stream)>>>(
      nedofs, nd, vd, byvdim, ndofs, gather_map, x, yE);
   MFA_CHECK_LAUNCH("k_gather_vdim");
   return 0;
}

extern "C" int mfem_amd_restrict_scatter(int64_t ndofs, const int *offsets, const int *indices,
                                         const double *xE, double *y, int add, int absv,
                                         void *stream)
{
   if (ndofs == 0) { return 0; }
   MFA_REQUIRE(ndofs > 0 && offsets && indices && xE && y, "restrict_scatter: bad arguments");
   MFA_REQUIRE(!(add && absv), "restrict_scatter: add and abs cannot be combined");
   const unsigned nb = (unsigned)((ndofs + 1023)/1024);      // four rows per thread (the two-row body loops twice)
   hipStream_t s = as_stream(stream);
   if (absv)
   {
      k_scatter<2, false><<<dim3(nb), dim3(256), 0, s>>>(ndofs, offsets, indices, xE, y, nullptr,
                                                       nullptr, nullptr, nullptr, nullptr, scatter_flags(), 0);
   }
   else if (add)
   {
      k_scatter<1, false><<<dim3(nb), dim3(256), 0, s>>>(ndofs, offsets, indices, xE, y, nullptr,
                                                       nullptr, nullptr, nullptr, nullptr, scatter_flags(), 0);
   }
   else
   {
      k_scatter<0, false><<<dim3(nb), dim3(256), 0, s>>>(ndofs, offsets, indices, xE, y, nullptr,
                                                       nullptr, nullptr, nullptr, nullptr, scatter_flags(), 0);
   }
   MFA_CHECK_LAUNCH("k_scatter");
   return 0;
}

// Are the last ne*(D-2)^3 L-rows the element-interior dofs, element by element in lexicographic order, each
// owned by its element alone?  (True of every H1 space numbered as the reference numbers it: vertex, edge, face,
// then interior dofs, fem/fespace.cpp; the gather map is then gmap[e*D^3 + i(iz+1,iy+1,ix+1)] = I0 + e*NI + li.)
__global__ void __launch_bounds__(256)
k_interior_rows_check(const int D, const int64_t ne, const int64_t ndofs, const int *__restrict__ gmap,
                      const int *__restrict__ offsets, int *__restrict__ bad)
{
   const int DI = D - 2, NI = DI*DI*DI, D3 = D*D*D;
   const int64_t total = ne*NI, i0 = ndofs - total;
   for (int64_t idx = (int64_t)blockIdx.x*256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x*256)
   {
      const int64_t e = idx/NI;
      const int li = (int)(idx - e*NI);
      const int ix = li % DI, iy = (li/DI) % DI, iz = li/(DI*DI);
      const int64_t row = gmap[e*D3 + ((iz + 1)*D + (iy + 1))*D + (ix + 1)];
      bool ok = row == i0 + idx;
      if (ok) { ok = offsets[row + 1] - offsets[row] == 1; }
      if (!ok) { atomicOr(bad, 1); }
   }
}

extern "C" int mfem_amd_interior_rows(int D1D, int64_t ne, int64_t ndofs, const int *gather_map,
                                      const int *offsets, int *work, int64_t *direct_rows, void *stream)
{
   MFA_REQUIRE(direct_rows, "interior_rows: direct_rows is NULL");
   *direct_rows = 0;
   if (D1D < 3 || ne <= 0) { return 0; }
   MFA_REQUIRE(D1D <= MAX_D1D && ndofs > 0 && gather_map && offsets && work, "interior_rows: bad arguments");
   const int64_t ni = (int64_t)(D1D - 2)*(D1D - 2)*(D1D - 2);
   if (ne*ni > ndofs) { return 0; }
   hipStream_t s = as_stream(stream);
   if (int rc = check_hip(hipMemsetAsync(work, 0, sizeof(int), s), "hipMemsetAsync")) { return rc; }
   k_interior_rows_check<<<dim3(grid_for(ne*ni)), dim3(256), 0, s>>>(D1D, ne, ndofs, gather_map, offsets, work);
   MFA_CHECK_LAUNCH("k_interior_rows_check");
   int bad = 1;
   if (int rc = check_hip(hipMemcpyAsync(&bad, work, sizeof(int), hipMemcpyDeviceToHost, s), "hipMemcpyAsync")) { return rc; }
   if (int rc = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize")) { return rc; }
   if (!bad) { *direct_rows = ne*ni; }
   return 0;
}

// `indices` of the transpose-gather for the packed E-vector (common.h: packed_pos): entry e*D^3 + i -> e*NS + packed_pos(i),
// signs kept (-1 - idx).  Element-interior entries (rows the element kernel writes itself) have no place in the packed
// form; they map to the element's first entry and are never read.
__global__ void __launch_bounds__(256)
k_pack_indices(const int D, const int64_t nnz, const int *__restrict__ indices, int *__restrict__ packed)
{
   const int D3 = D*D*D, NS = packed_size(D);
   for (int64_t k = (int64_t)blockIdx.x*256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x*256)
   {
      const int idx = indices[k];
      const int j = idx >= 0 ? idx : -1 - idx;
      const int e = j/D3, i = j - e*D3;
      const int pos = packed_pos(D, i/(D*D), (i/D) % D, i % D);
      const int out = e*NS + (pos < 0 ? 0 : pos);
      packed[k] = idx >= 0 ? out : -1 - out;
   }
}

extern "C" int mfem_amd_pack_indices(int D1D, int64_t nnz, const int *indices, int *packed_indices, void *stream)
{
   if (nnz == 0) { return 0; }
   MFA_REQUIRE(D1D >= 3 && D1D <= MAX_D1D && nnz > 0 && indices && packed_indices, "pack_indices: bad arguments (D1D >= 3)");
   const int64_t nb = (nnz + 255)/256;
   k_pack_indices<<<dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, as_stream(stream)>>>(D1D, nnz, indices, packed_indices);
   MFA_CHECK_LAUNCH("k_pack_indices");
   return 0;
}

extern "C" int mfem_amd_restrict_scatter_subrows(int64_t nsub, const int *rows, const int *offsets,
                                                 const int *indices, const double *xE, double *y,
                                                 void *stream)
{
   if (nsub == 0) { return 0; }
   MFA_REQUIRE(nsub > 0 && rows && offsets && indices && xE && y, "restrict_scatter_subrows: bad arguments");
   const unsigned nb = (unsigned)((nsub + 1023)/1024);
   k_scatter_sub<false><<<dim3(nb), dim3(256), 0, as_stream(stream)>>>(nsub, rows, offsets, indices, xE, y, nullptr,
                                                                       nullptr, nullptr, nullptr, nullptr);
   MFA_CHECK_LAUNCH("k_scatter_sub");
   return 0;
}

extern "C" int mfem_amd_set_marker(int64_t nlist, const int *list, unsigned char *marker,
                                   void *stream)
{
   if (nlist == 0) { return 0; }
   MFA_REQUIRE(nlist > 0 && list && marker, "set_marker: bad arguments");
   k_set_marker<<<dim3(grid_for(nlist)), dim3(256), 0, as_stream(stream)>>>(nlist, list, marker);
   MFA_CHECK_LAUNCH("k_set_marker");
   return 0;
}

extern "C" int mfem_amd_mask_gather_map(int64_t nedofs, const int *gather_map,
                                        const unsigned char *marker, int *masked, void *stream)
{
   if (nedofs == 0) { return 0; }
   MFA_REQUIRE(nedofs > 0 && gather_map && marker && masked, "mask_gather_map: bad arguments");
   k_mask_gather<<<dim3(grid_for(nedofs)), dim3(256), 0, as_stream(stream)>>>(nedofs, gather_map, marker, masked);
   MFA_CHECK_LAUNCH("k_mask_gather");
   return 0;
}

extern "C" int mfem_amd_set_subvector(int64_t nlist, const int *list, const double *x,
                                      double value, double *y, void *stream)
{
   if (nlist == 0) { return 0; }
   MFA_REQUIRE(nlist > 0 && list && y, "set_subvector: bad arguments");
   k_set_subvector<<<dim3(grid_for(nlist)), dim3(256), 0, as_stream(stream)>>>(nlist, list, x, value, y);
   MFA_CHECK_LAUNCH("k_set_subvector");
   return 0;
}

extern "C" int mfem_amd_shared_pack(int64_t n, const int *idx, const double *x, double *buf, void *stream)
{
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && idx && x && buf, "shared_pack: bad arguments");
   k_pack<<<dim3(grid_for(n)), dim3(256), 0, as_stream(stream)>>>(n, idx, x, buf);
   MFA_CHECK_LAUNCH("k_pack");
   return 0;
}

extern "C" int mfem_amd_shared_unpack(int64_t n, const int *idx, const double *buf, double *y, void *stream)
{
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && idx && buf && y, "shared_unpack: bad arguments");
   k_unpack<<<dim3(grid_for(n)), dim3(256), 0, as_stream(stream)>>>(n, idx, buf, y);
   MFA_CHECK_LAUNCH("k_unpack");
   return 0;
}

extern "C" int mfem_amd_shared_accumulate(int64_t nuniq, const int *unq_ldof, const int *unq_shr_i,
                                          const int *unq_shr_j, const double *shr_buf, double *y,
                                          void *stream)
{
   if (nuniq == 0) { return 0; }
   MFA_REQUIRE(nuniq > 0 && unq_ldof && unq_shr_i && unq_shr_j && shr_buf && y, "shared_accumulate: bad arguments");
   k_accumulate<<<dim3(grid_for(nuniq)), dim3(256), 0, as_stream(stream)>>>(nuniq, unq_ldof, unq_shr_i, unq_shr_j, shr_buf, y);
   MFA_CHECK_LAUNCH("k_accumulate");
   return 0;
}

extern "C" int mfem_amd_evec_add_marked(int64_t ne, int nd, int nmarkers, const int *markers,
                                        const int *attributes, const double *xE, double *yE, void *stream)
{
   const int64_t n = ne*nd;
   if (n == 0) { return 0; }
   MFA_REQUIRE(ne > 0 && nd > 0 && nmarkers >= 0 && markers && attributes && xE && yE && xE != yE, "evec_add_marked: bad arguments");
   k_evec_markers<0><<<dim3(grid_for(n)), dim3(256), 0, as_stream(stream)>>>(n, nd, nmarkers, markers, attributes, xE, yE);
   MFA_CHECK_LAUNCH("k_evec_markers<add>");
   return 0;
}

extern "C" int mfem_amd_evec_zero_unmarked(int64_t ne, int nd, int nmarkers, const int *markers,
                                           const int *attributes, double *dE, void *stream)
{
   const int64_t n = ne*nd;
   if (n == 0) { return 0; }
   MFA_REQUIRE(ne > 0 && nd > 0 && nmarkers >= 0 && markers && attributes && dE, "evec_zero_unmarked: bad arguments");
   k_evec_markers<1><<<dim3(grid_for(n)), dim3(256), 0, as_stream(stream)>>>(n, nd, nmarkers, markers, attributes, nullptr, dE);
   MFA_CHECK_LAUNCH("k_evec_markers<zero>");
   return 0;
}

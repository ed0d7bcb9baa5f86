// Tuning aid (not part of the product): what an E-vector stored in L-dof order would cost and buy.
//   k_segsum   : E->L as a contiguous segmented sum y[i] = sum yS[off[i] .. off[i+1])   (no index array)
//   k_permute  : yS[pos[i]] = yE[i]  -- the scattered 8-byte stores the element kernel would issue instead
//                of its coalesced E-vector store (pos = inverse of the CSR `indices`)
//   k_copy     : yS[i] = yE[i]       -- the coalesced baseline of the same volume
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC tools/e2l_sorted_probe.hip -o tools/libe2l_sorted.so
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int SK = 4;

__global__ void __launch_bounds__(256)
k_segsum(const int64_t ndofs, const int *__restrict__ offsets, const double *__restrict__ yS, double *__restrict__ y)
{
   for (int64_t base = (int64_t)blockIdx.x*(256*SK); base < ndofs; base += (int64_t)gridDim.x*(256*SK))
   {
      int o0[SK], n[SK];
      double t0[SK], t1[SK], v[SK];
      const int64_t last = ndofs - 1;
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         const int64_t ic = i < ndofs ? i : last;
         o0[j] = offsets[ic];
         n[j] = offsets[ic + 1] - o0[j];
      }
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         t0[j] = yS[o0[j]];
         t1[j] = yS[o0[j] + (n[j] > 1 ? 1 : 0)];
      }
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         double a = 0.0;
         if (n[j] > 0) { a += t0[j]; }
         if (n[j] > 1) { a += t1[j]; }
         for (int k = 2; k < n[j]; k++) { a += yS[o0[j] + k]; }
         v[j] = a;
      }
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         if (i < ndofs) { y[i] = v[j]; }
      }
   }
}

// row-driven E->L with NB index/value pairs of every row in flight (the shipped kernel keeps 2), rows longer
// than NB finish sequentially from entry NB on
template <int NB, int SK>
__device__ __forceinline__ void rows_body(const int64_t ndofs, const int *__restrict__ offsets, const int *__restrict__ indices,
                                          const double *__restrict__ xE, double *__restrict__ y)
{
   for (int64_t base = (int64_t)blockIdx.x*(256*SK); base < ndofs; base += (int64_t)gridDim.x*(256*SK))
   {
      int o0[SK], n[SK], id[SK][NB];
      double t[SK][NB], v[SK];
      const int64_t last = ndofs - 1;
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         const int64_t ic = i < ndofs ? i : last;
         o0[j] = offsets[ic];
         n[j] = offsets[ic + 1] - o0[j];
      }
#pragma unroll
      for (int j = 0; j < SK; j++)
#pragma unroll
         for (int k = 0; k < NB; k++) { id[j][k] = indices[o0[j] + (k < n[j] ? k : (n[j] > 0 ? n[j] - 1 : 0))]; }
#pragma unroll
      for (int j = 0; j < SK; j++)
#pragma unroll
         for (int k = 0; k < NB; k++) { t[j][k] = xE[id[j][k]]; }
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         double a = 0.0;
#pragma unroll
         for (int k = 0; k < NB; k++) { if (k < n[j]) { a += t[j][k]; } }
         for (int k = NB; k < n[j]; k++) { a += xE[indices[o0[j] + k]]; }
         v[j] = a;
      }
#pragma unroll
      for (int j = 0; j < SK; j++)
      {
         const int64_t i = base + threadIdx.x + j*256;
         if (i < ndofs) { y[i] = v[j]; }
      }
   }
}

template <int NB, int SKK = 4>
__global__ void __launch_bounds__(256)
k_rows(const int64_t ndofs, const int *__restrict__ offsets, const int *__restrict__ indices,
       const double *__restrict__ xE, double *__restrict__ y)
{
   rows_body<NB, SKK>(ndofs, offsets, indices, xE, y);
}

// one kernel, the batch depth chosen from the table's own average row length (a wave-uniform branch)
__global__ void __launch_bounds__(256)
k_rows_auto(const int64_t ndofs, const int *__restrict__ offsets, const int *__restrict__ indices,
            const double *__restrict__ xE, double *__restrict__ y)
{
   const int64_t nnz = offsets[ndofs];
   if (nnz >= 3*ndofs) { rows_body<8, 2>(ndofs, offsets, indices, xE, y); }
   else { rows_body<4, 4>(ndofs, offsets, indices, xE, y); }
}

extern "C" int probe_rows(int nb_pairs, int64_t ndofs, const int *offsets, const int *indices, const double *xE, double *y)
{
   if (ndofs <= 0 || !offsets || !indices || !xE || !y) { return -1; }
   const unsigned nb = (unsigned)((ndofs + 256*SK - 1)/(256*SK));
   if (nb_pairs == 4) { k_rows<4><<<dim3(nb), dim3(256), 0, 0>>>(ndofs, offsets, indices, xE, y); }
   else if (nb_pairs == 8) { k_rows<8><<<dim3(nb), dim3(256), 0, 0>>>(ndofs, offsets, indices, xE, y); }
   else if (nb_pairs >= 100)      // 100*NB + SKK
   {
      const int NBv = nb_pairs/100, SKv = nb_pairs % 100;
      const unsigned g = (unsigned)((ndofs + 256*SKv - 1)/(256*SKv));
#define V(A, B) if (NBv == A && SKv == B) { k_rows<A, B><<<dim3(g), dim3(256), 0, 0>>>(ndofs, offsets, indices, xE, y); return (int)hipGetLastError(); }
      V(4, 6) V(4, 8) V(3, 4) V(3, 6) V(3, 8) V(2, 8) V(2, 4) V(6, 2) V(6, 3) V(4, 2) V(4, 3) V(5, 3)
#undef V
      return -3;
   }
   else if (nb_pairs == 0) { k_rows_auto<<<dim3(nb), dim3(256), 0, 0>>>(ndofs, offsets, indices, xE, y); }
   else { return -2; }
   return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256)
k_permute(const int64_t n, const int *__restrict__ pos, const double *__restrict__ yE, double *__restrict__ yS)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i < n) { yS[pos[i]] = yE[i]; }
}

__global__ void __launch_bounds__(256)
k_copy(const int64_t n, const double *__restrict__ yE, double *__restrict__ yS)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i < n) { yS[i] = yE[i]; }
}

extern "C" int probe_segsum(int64_t ndofs, const int *offsets, const double *yS, double *y)
{
   if (ndofs <= 0 || !offsets || !yS || !y) { return -1; }
   const unsigned nb = (unsigned)((ndofs + 256*SK - 1)/(256*SK));
   k_segsum<<<dim3(nb), dim3(256), 0, 0>>>(ndofs, offsets, yS, y);
   return (int)hipGetLastError();
}
extern "C" int probe_permute(int64_t n, const int *pos, const double *yE, double *yS)
{
   if (n <= 0 || !pos || !yE || !yS) { return -1; }
   k_permute<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, 0>>>(n, pos, yE, yS);
   return (int)hipGetLastError();
}
extern "C" int probe_copy(int64_t n, const double *yE, double *yS)
{
   if (n <= 0 || !yE || !yS) { return -1; }
   k_copy<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, 0>>>(n, yE, yS);
   return (int)hipGetLastError();
}

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

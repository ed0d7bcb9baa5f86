// gridbar_probe -- what a grid-wide barrier costs on MI355X, against the ~6 us that a dependent kernel launch costs
// (DESIGN 8-3): the number that decides whether a persistent, cooperative PCG kernel (five barriers per iteration instead
// of six kernel boundaries) can pay at 1e5..1e6 dofs.  One resident workgroup set (grid = k workgroups per CU), B barriers
// in a row, each an atomic arrival counter + a spin on a generation word (agent scope), with a token amount of work between
// them; the same work as B dependent launches of a trivial kernel beside it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gridbar_probe tools/gridbar_probe.hip ;  tools/gridbar_probe [wgs_per_cu] [barriers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned *count, volatile unsigned *gen, const unsigned nwg)
{
   __syncthreads();
   if (threadIdx.x == 0)
   {
      const unsigned g = __hip_atomic_load((unsigned*)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1)
      {
         __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store((unsigned*)gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      else
      {
         while (__hip_atomic_load((unsigned*)gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) { __builtin_amdgcn_s_sleep(1); }
      }
   }
   __syncthreads();
}

__global__ void __launch_bounds__(256) k_barriers(const int B, unsigned *count, unsigned *gen, double *x, const int n)
{
   for (int b = 0; b < B; b++)
   {
      for (int i = blockIdx.x*256 + threadIdx.x; i < n; i += gridDim.x*256) { x[i] = x[i]*1.0000001 + 1e-9; }
      grid_barrier(count, gen, gridDim.x);
   }
}
__global__ void __launch_bounds__(256) k_work(double *x, const int n)
{
   for (int i = blockIdx.x*256 + threadIdx.x; i < n; i += gridDim.x*256) { x[i] = x[i]*1.0000001 + 1e-9; }
}

int main(int argc, char **argv)
{
   const int per_cu = argc > 1 ? atoi(argv[1]) : 1, B = argc > 2 ? atoi(argv[2]) : 1000;
   hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
   const int cus = prop.multiProcessorCount, nwg = per_cu*cus, n = 1 << 17;
   unsigned *sync; double *x;
   hipMalloc(&sync, 256); hipMemset(sync, 0, 256); hipMalloc(&x, n*8); hipMemset(x, 0, n*8);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   float tb = 0, tl = 0;
   for (int w = 0; w < 2; w++)
   {
      hipEventRecord(e0);
      void *args[] = {(void*)&B, (void*)&sync, (void*)&sync, (void*)&x, (void*)&n};
      unsigned *gen = sync + 32;
      args[2] = (void*)&gen;
      if (hipLaunchCooperativeKernel((const void*)k_barriers, dim3(nwg), dim3(256), args, 0, 0) != hipSuccess) { printf("cooperative launch failed\n"); return 1; }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&tb, e0, e1);
      hipEventRecord(e0);
      for (int b = 0; b < B; b++) { k_work<<<nwg, 256>>>(x, n); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&tl, e0, e1);
   }
   printf("%d CUs, %d workgroups (%d per CU), %d steps of %d doubles: grid barrier %.2f us per step, dependent kernel launch %.2f us per step\n",
          cus, nwg, per_cu, B, n, tb/B*1e3, tl/B*1e3);
   return 0;
}

// Streaming rate by access width (tuning probe, gfx950): each workgroup of 256 threads sums a contiguous slice of a buffer read
// 4, 8 or 16 bytes a lane, cached or non-temporal, with U loads in flight per thread; and stores 8 / 16 bytes a lane.
// hipcc --offload-arch=gfx950 -O3 -o tools/ldwidth_probe tools/ldwidth_probe.hip ; tools/ldwidth_probe [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <class T, bool NT> __device__ __forceinline__ T ld(const T *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ int red(int v) { return v; }
__device__ __forceinline__ int red(v2i v) { return v.x ^ v.y; }
__device__ __forceinline__ int red(v4i v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <class T, bool NT, int U>
__global__ void __launch_bounds__(256) k_read(const T *__restrict__ src, const size_t n, int *out)
{
   int acc = 0;
   const size_t per = (size_t)256*U;
   for (size_t base = (size_t)blockIdx.x*per; base + per <= n; base += (size_t)gridDim.x*per)
   {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) { v[u] = ld<T, NT>(&src[base + threadIdx.x + 256*u]); }
#pragma unroll
      for (int u = 0; u < U; u++) { acc ^= red(v[u]); }
   }
   if (acc == 0x12345678) { out[0] = acc; }
}
template <class T>
__global__ void __launch_bounds__(256) k_write(T *__restrict__ dst, const size_t n, const T val)
{
   for (size_t i = (size_t)blockIdx.x*256 + threadIdx.x; i < n; i += (size_t)gridDim.x*256) { dst[i] = val; }
}
template <class F> float timeit(F f, int reps)
{
   hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
   for (int i = 0; i < 5; i++) { f(); }
   hipEventRecord(a);
   for (int i = 0; i < reps; i++) { f(); }
   hipEventRecord(b); hipEventSynchronize(b);
   float ms = 0; hipEventElapsedTime(&ms, a, b);
   return ms/reps;
}
int main(int argc, char **argv)
{
   const size_t mb = argc > 1 ? atoi(argv[1]) : 1024;
   const size_t bytes = mb << 20;
   char *buf; int *out;
   CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&out, 64));
   CK(hipMemset(buf, 1, bytes));
   const int grid = 256*8;
   // spin the clocks up
   for (int i = 0; i < 200; i++) { k_read<v4i, false, 4><<<grid, 256>>>((const v4i*)buf, bytes/16, out); }
   CK(hipDeviceSynchronize());
#define RD(T, NT, U) { float ms = timeit([&]() { k_read<T, NT, U><<<grid, 256>>>((const T*)buf, bytes/sizeof(T), out); }, 20); \
   printf("read  %2zu B/lane %s U=%2d : %7.3f ms  %6.2f TB/s\n", sizeof(T), NT ? "nt    " : "cached", U, ms, bytes/ms*1e-9); }
   RD(int, false, 4) RD(int, false, 16) RD(int, true, 4) RD(int, true, 16)
   RD(v2i, false, 4) RD(v2i, false, 8) RD(v2i, true, 4) RD(v2i, true, 8)
   RD(v4i, false, 2) RD(v4i, false, 4) RD(v4i, true, 2) RD(v4i, true, 4)
   { float ms = timeit([&]() { k_write<double><<<grid, 256>>>((double*)buf, bytes/8, 1.0); }, 20); printf("write  8 B/lane : %7.3f ms %6.2f TB/s\n", ms, bytes/ms*1e-9); }
   { double2 v; v.x = 1.0; v.y = 2.0; float ms = timeit([&]() { k_write<double2><<<grid, 256>>>((double2*)buf, bytes/16, v); }, 20); printf("write 16 B/lane : %7.3f ms %6.2f TB/s\n", ms, bytes/ms*1e-9); }
   CK(hipDeviceSynchronize());
   printf("done\n");
   return 0;
}

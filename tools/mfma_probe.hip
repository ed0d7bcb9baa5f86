// mfma_probe -- north_star: "fp64 MFMA is used only if rocprof shows it beats the vector path at high p".
// The x contraction of the (9,10) diffusion kernel (BX, GX = B x, G x along dx for the 81 (dy,dz) rows of an element,
// fem/integ/bilininteg_diffusion_kernels.hpp:1064-1082), standalone, in two forms:
//   k_vec : the library's form -- one thread per (dy,dz) row, the row's D values in registers, every output by fp64 FMA with
//           the basis factors as scalar operands (csrc/pa_apply.hip, RB contractions);
//   k_mfma: v_mfma_f64_16x16x4_f64 -- M = Q padded to 16 (rows qx), K = D padded to 12 (three k-steps), N = the 81 rows padded
//           to 96 (six column tiles): 36 MFMAs per element for B and G.
// Both read X (D^3 per element) from memory and write BX, GX (2 Q D^2 per element).  With R = 1 that traffic bounds both
// forms (77 MB per launch at 4096 elements); with R > 1 the contraction is repeated R times in registers on slightly
// shifted inputs, accumulating into the same outputs, so that the timing is the arithmetic's.  Run under rocprofv3
// --kernel-trace --stats and --pmc.   usage: mfma_probe [elements] [launches] [R]
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

constexpr int D = 9, Q = 10, D2 = D*D, D3 = D*D*D;
struct Basis { double b[Q*D], g[Q*D]; };     // [q*D + d], by value: scalar operands

__global__ void __launch_bounds__(256) k_vec(const int ne, const int R, const Basis bg, const double *__restrict__ X, double *__restrict__ BX, double *__restrict__ GX)
{
   // three elements per workgroup, 81 of each 85-thread group active... simpler: thread t -> row t of a flat (element,row) space
   const int64_t r = (int64_t)blockIdx.x*256 + threadIdx.x;         // row = e*81 + (dz*9+dy)
   if (r >= (int64_t)ne*D2) { return; }
   double x[D], u[Q], v[Q];
#pragma unroll
   for (int d = 0; d < D; d++) { x[d] = X[r*D + d]; }
#pragma unroll
   for (int q = 0; q < Q; q++) { u[q] = 0.0; v[q] = 0.0; }
   for (int rep = 0; rep < R; rep++)
   {
      // as in csrc/pa_apply.hip: an opaque 0 keeps the 2*Q*D basis factors from being hoisted out of the loop, where they
      // do not fit the SGPR file (102) and are spilled to VGPR lanes -- 12 v_readlane per FMA in the first version of this
      // probe; re-read per pass they are scalar loads from the kernel-argument segment (scalar cache)
      int zo;
      asm volatile("s_mov_b32 %0, 0" : "=s"(zo));
#pragma unroll
      for (int q = 0; q < Q; q++)
      {
#pragma unroll
         for (int d = 0; d < D; d++) { u[q] += bg.b[q*D + d + zo]*x[d]; v[q] += bg.g[q*D + d + zo]*x[d]; }
      }
#pragma unroll
      for (int d = 0; d < D; d++) { x[d] += 1e-3; }
   }
#pragma unroll
   for (int q = 0; q < Q; q++) { BX[r*Q + q] = u[q]; GX[r*Q + q] = v[q]; }
}

typedef double d4 __attribute__((ext_vector_type(4)));

// one wave per element: lane l = (k = l/16, n = l%16)
__global__ void __launch_bounds__(256) k_mfma(const int ne, const int R, const int layout, const double *__restrict__ Bpad, const double *__restrict__ Gpad,
                                              const double *__restrict__ X, double *__restrict__ BX, double *__restrict__ GX)
{
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int e = blockIdx.x*4 + wave;
   if (e >= ne) { return; }
   const int kk = lane >> 4, n = lane & 15;
   // A operands: Bpad[16][12] (rows qx, zero padded), lane holds A[row = n][k = 4*ks + kk]
   double ab[3], ag[3];
#pragma unroll
   for (int ks = 0; ks < 3; ks++) { ab[ks] = Bpad[n*12 + 4*ks + kk]; ag[ks] = Gpad[n*12 + 4*ks + kk]; }
   const double *Xe = X + (int64_t)e*D3;
#pragma unroll
   for (int t = 0; t < 6; t++)
   {
      const int col = t*16 + n;                  // (dz,dy) row of the element, 81 valid
      d4 cb = {0, 0, 0, 0}, cg = {0, 0, 0, 0};
      double xb[3];
#pragma unroll
      for (int ks = 0; ks < 3; ks++)
      {
         const int k = 4*ks + kk;
         xb[ks] = (col < D2 && k < D) ? Xe[col*D + k] : 0.0;     // B operand: X[k][col]
      }
      for (int rep = 0; rep < R; rep++)
      {
#pragma unroll
         for (int ks = 0; ks < 3; ks++)
         {
            cb = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks], xb[ks], cb, 0, 0, 0);
            cg = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[ks], xb[ks], cg, 0, 0, 0);
         }
#pragma unroll
         for (int ks = 0; ks < 3; ks++) { xb[ks] += (col < D2 && 4*ks + kk < D) ? 1e-3 : 0.0; }
      }
      // D: lane holds rows 4*kk + i, column n
      if (col < D2)
      {
#pragma unroll
         for (int i = 0; i < 4; i++)
         {
            const int q = layout ? 4*i + kk : 4*kk + i;       // (accumulator layout found at run time: main tries both)
            if (q < Q) { BX[((int64_t)e*D2 + col)*Q + q] = cb[i]; GX[((int64_t)e*D2 + col)*Q + q] = cg[i]; }
         }
      }
   }
}

int main(int argc, char **argv)
{
   const int ne = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 200, R = argc > 3 ? atoi(argv[3]) : 1;
   std::vector<double> hx((size_t)ne*D3), hb(16*12, 0.0), hg(16*12, 0.0);
   Basis bg;
   for (int q = 0; q < Q; q++) for (int d = 0; d < D; d++)
   {
      bg.b[q*D + d] = std::sin(0.3 + q*1.7 + d*0.9); bg.g[q*D + d] = std::cos(0.1 + q*0.7 + d*1.3);
      hb[q*12 + d] = bg.b[q*D + d]; hg[q*12 + d] = bg.g[q*D + d];
   }
   for (size_t i = 0; i < hx.size(); i++) { hx[i] = std::sin(0.001*i) + 0.5; }
   double *X, *BX, *GX, *BX2, *GX2, *Bp, *Gp;
   const size_t nout = (size_t)ne*D2*Q;
   hipMalloc(&X, hx.size()*8); hipMalloc(&BX, nout*8); hipMalloc(&GX, nout*8); hipMalloc(&BX2, nout*8); hipMalloc(&GX2, nout*8);
   hipMalloc(&Bp, 16*12*8); hipMalloc(&Gp, 16*12*8);
   hipMemcpy(X, hx.data(), hx.size()*8, hipMemcpyHostToDevice);
   hipMemcpy(Bp, hb.data(), 16*12*8, hipMemcpyHostToDevice); hipMemcpy(Gp, hg.data(), 16*12*8, hipMemcpyHostToDevice);
   const int gv = (int)(((int64_t)ne*D2 + 255)/256), gm = (ne + 3)/4;
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   float tv = 0, tm = 0;
   // which accumulator layout does v_mfma_f64_16x16x4_f64 have?  compare one launch of each with the vector result
   int layout = 0;
   {
      std::vector<double> a(nout), b(nout);
      k_vec<<<gv, 256>>>(ne, R, bg, X, BX, GX);
      hipMemcpy(a.data(), BX, nout*8, hipMemcpyDeviceToHost);
      double best = 1e300;
      for (int l = 0; l < 2; l++)
      {
         k_mfma<<<gm, 256>>>(ne, R, l, Bp, Gp, X, BX2, GX2);
         hipMemcpy(b.data(), BX2, nout*8, hipMemcpyDeviceToHost);
         double err = 0;
         for (size_t i = 0; i < nout; i++) { err = std::fmax(err, std::fabs(a[i] - b[i])); }
         if (err < best) { best = err; layout = l; }
      }
      printf("accumulator layout %d (row = %s)\n", layout, layout ? "4*i + lane/16" : "4*(lane/16) + i");
   }
   for (int w = 0; w < 2; w++)
   {
      hipEventRecord(e0);
      for (int i = 0; i < reps; i++) { k_vec<<<gv, 256>>>(ne, R, bg, X, BX, GX); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&tv, e0, e1);
      hipEventRecord(e0);
      for (int i = 0; i < reps; i++) { k_mfma<<<gm, 256>>>(ne, R, layout, Bp, Gp, X, BX2, GX2); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&tm, e0, e1);
   }
   std::vector<double> a(nout), b(nout);
   hipMemcpy(a.data(), BX, nout*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), BX2, nout*8, hipMemcpyDeviceToHost);
   double err = 0, mx = 0;
   for (size_t i = 0; i < nout; i++) { err = std::fmax(err, std::fabs(a[i] - b[i])); mx = std::fmax(mx, std::fabs(a[i])); }
   hipMemcpy(a.data(), GX, nout*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), GX2, nout*8, hipMemcpyDeviceToHost);
   for (size_t i = 0; i < nout; i++) { err = std::fmax(err, std::fabs(a[i] - b[i])); }
   const double flop = 2.0*2*Q*D*D2*ne*R;        // useful flops (B and G)
   printf("R = %d: ", R);
   printf("(D,Q) = (%d,%d), %d elements, %d launches: vector FMA %.2f us/launch (%.2f TFLOP/s useful), fp64 MFMA 16x16x4 %.2f us/launch "
          "(%.2f TFLOP/s useful, %.2f issued); max |difference| %.2e of %.2e\n", D, Q, ne, reps, tv/reps*1e3, flop/(tv/reps*1e-3)*1e-12,
          tm/reps*1e3, flop/(tm/reps*1e-3)*1e-12, 36.0*2048*ne*R/(tm/reps*1e-3)*1e-12, err, mx);
   return err <= 1e-11*mx ? 0 : 1;
}

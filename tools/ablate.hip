// Tuning aid (not part of the product): ablation variants of the BP3 element kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC tools/ablate.hip mfem_amd/csrc/runtime.hip -o tools/libablate.so
#define MFA_ABLATION 1   // compile the ablation branches of k_pa_apply in (the library is built without them)
#include "../mfem_amd/csrc/pa_apply.hip"

template <int D, int Q, int ABL, bool BLK>
static int run(int64_t ne, const double *B, const double *G, const double *Dq, const double *x,
               const int *gmap, double *yE, int grid_per_cu)
{
   constexpr int NBZ = nbz_for(Q);
   BasisT<D*Q> bg;
   fill_basis(bg, D, Q, B, G);
   const int64_t nb = (ne + NBZ - 1)/NBZ;
   int64_t grid = 256*(int64_t)grid_per_cu;
   if (grid > nb) { grid = nb; }
   k_pa_apply<1, D, Q, NBZ, true, false, BLK, ABL><<<dim3((unsigned)grid), dim3(Q, Q, NBZ), 0, 0>>>(ne, nb, (ABL & 256) ? 0 : 1, bg, Dq, x, gmap, yE, mfa::VecArgs{});
   return (int)hipGetLastError();
}

extern "C" int ablate_run(int D, int abl, int blk, int64_t ne, const double *B, const double *G, const double *Dq,
                          const double *x, const int *gmap, double *yE, int grid_per_cu)
{
#define CASE(DD, QQ) if (D == DD) { \
      if (abl == 0) return blk ? run<DD, QQ, 0, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu) : run<DD, QQ, 0, false>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 1) return blk ? run<DD, QQ, 1, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu) : run<DD, QQ, 1, false>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 49) return run<DD, QQ, 49, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 81) return run<DD, QQ, 81, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 145) return run<DD, QQ, 145, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 273) return run<DD, QQ, 273, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 256) return run<DD, QQ, 256, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 17) return run<DD, QQ, 17, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 5) return run<DD, QQ, 5, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 9) return run<DD, QQ, 9, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      if (abl == 13) return run<DD, QQ, 13, true>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); \
      return run<DD, QQ, 2, false>(ne, B, G, Dq, x, gmap, yE, grid_per_cu); }
   CASE(3, 4) CASE(5, 6) CASE(7, 8)
   return -1;
}

// Tuning aid (not part of the product): compute-phase ablations of the element kernel in its product
// configuration (one batch per workgroup, register-blocked contractions), pa_data not read (ABL & 2), so the
// time is the contraction phases' alone.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC tools/ablate2.hip mfem_amd/csrc/runtime.hip -o tools/libablate2.so
#define MFA_ABLATION 1   // compile the ablation branches of k_pa_apply in (the library is built without them)
#include "../mfem_amd/csrc/pa_apply.hip"

template <int KIND, int D, int Q, int ABL, int RB>
static int run(int64_t ne, const double *B, const double *G, const double *Dq, const double *x,
               const int *gmap, double *yE)
{
   constexpr int NBZ = nbz_for(Q);
   BasisT<D*Q> bg;
   fill_basis(bg, D, Q, B, G);
   const int64_t nb = (ne + NBZ - 1)/NBZ;
   k_pa_apply<KIND, D, Q, NBZ, true, false, true, ABL, false, 0, RB><<<dim3((unsigned)nb), dim3(Q, Q, NBZ), 0, 0>>>(ne, nb, 1, bg, Dq, x, gmap, yE, mfa::VecArgs{});
   return (int)hipGetLastError();
}

extern "C" int ablate2_run(int D, int abl, int rb, int64_t ne, const double *B, const double *G, const double *Dq,
                           const double *x, const int *gmap, double *yE)
{
#define VAR(DD, QQ, A, R) if (D == DD && abl == A && rb == R) { return run<1, DD, QQ, A, R>(ne, B, G, Dq, x, gmap, yE); }
#define CASE(DD, QQ) \
   VAR(DD, QQ, 0, 1) VAR(DD, QQ, 2, 1) VAR(DD, QQ, 2 + 512, 1) VAR(DD, QQ, 2 + 1024, 1) VAR(DD, QQ, 2 + 2048, 1) VAR(DD, QQ, 2 + 4096, 1) \
   VAR(DD, QQ, 2 + 1024 + 2048 + 4096, 1) VAR(DD, QQ, 2, 2) VAR(DD, QQ, 2 + 512, 2) VAR(DD, QQ, 2, 0)
   CASE(5, 6) CASE(6, 7) CASE(7, 8)
   return -1;
}

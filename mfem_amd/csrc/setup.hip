// PA setup on gfx950: geometric factors, quadrature-point data, element diagonals.
//
// Replaces GeometricFactors::Compute + Derivatives3D/Det3D (mesh/mesh.cpp:15178-15231,
// fem/qinterp/grad.hpp:232-374, fem/qinterp/det.hpp:209-282), PADiffusionSetup3D
// (fem/integ/bilininteg_diffusion_kernels.cpp:243-367), the mass setup lambda
// (fem/integ/bilininteg_mass_pa.cpp:72-77), SmemPADiffusionDiagonal3D
// (fem/integ/bilininteg_diffusion_kernels.hpp:368-484) and SmemPAMassAssembleDiagonal3D
// (fem/integ/bilininteg_mass_kernels.hpp:324-408).
//
// All of these run once per operator (setup), are HBM-bound streams, and take runtime
// (Dn, Q1D) so that any nodal order of a curved mesh works.  The fused nodes->D kernel
// never materialises J (9*Q^3*NE doubles = 1.5x the size of pa_data): it reads 3*Dn^3
// doubles of nodes per element and writes the 6*Q^3 (or Q^3) doubles of D once.
#include "common.h"

namespace mfa
{

// J at the quadrature points of one element from its E-nodes, by sum factorisation
// through LDS.  On return (after the trailing barrier) thread-strided q-points can read
//   J(r,c) at point q  =  sJ[(r*3 + c)*Q3 + q].
// LDS layout (doubles): sB[Q*Dn] sG[Q*Dn] sN[3*Dn^3] s1[3*2*Dn^2*Q] s2[3*3*Dn*Q^2] sJ[9*Q^3]
__device__ __forceinline__ void element_jacobians(const int Dn, const int Q,
                                                  const BasisT<MAX_DQ> &bg,
                                                  const double *__restrict__ nodesE_e,
                                                  double *dyn, double *&sJ_out)
{
   const int D3 = Dn*Dn*Dn, DDQ = Dn*Dn*Q, DQQ = Dn*Q*Q, Q3 = Q*Q*Q, QQ = Q*Q;
   double *sB = dyn, *sG = sB + Q*Dn, *sN = sG + Q*Dn;
   double *s1 = sN + 3*D3, *s2 = s1 + 6*DDQ, *sJ = s2 + 9*DQQ;
   const int tid = threadIdx.x, nt = blockDim.x;
   for (int i = tid; i < Q*Dn; i += nt) { sB[i] = bg.b[i]; sG[i] = bg.g[i]; }
   for (int i = tid; i < 3*D3; i += nt) { sN[i] = nodesE_e[i]; }
   __syncthreads();
   for (int i = tid; i < 3*DDQ; i += nt)        // [r][dz][dy][qx] : B and G along x
   {
      const int r = i/DDQ, j = i % DDQ;
      const int qx = j % Q, dyz = j/Q;
      double u = 0.0, v = 0.0;
      for (int dx = 0; dx < Dn; dx++)
      {
         const double s = sN[r*D3 + dyz*Dn + dx];
         u += sB[qx*Dn + dx]*s;
         v += sG[qx*Dn + dx]*s;
      }
      s1[(2*r)*DDQ + j] = u;
      s1[(2*r + 1)*DDQ + j] = v;
   }
   __syncthreads();
   for (int i = tid; i < 3*DQQ; i += nt)        // [r][c][dz][qy][qx]
   {
      const int r = i/DQQ, j = i % DQQ;
      const int qx = j % Q, qy = (j/Q) % Q, dz = j/QQ;
      double u = 0.0, v = 0.0, w = 0.0;
      for (int dy = 0; dy < Dn; dy++)
      {
         const double bx = s1[(2*r)*DDQ + (dz*Dn + dy)*Q + qx];
         const double gx = s1[(2*r + 1)*DDQ + (dz*Dn + dy)*Q + qx];
         u += gx*sB[qy*Dn + dy];
         v += bx*sG[qy*Dn + dy];
         w += bx*sB[qy*Dn + dy];
      }
      s2[(3*r + 0)*DQQ + j] = u;
      s2[(3*r + 1)*DQQ + j] = v;
      s2[(3*r + 2)*DQQ + j] = w;
   }
   __syncthreads();
   for (int i = tid; i < 3*Q3; i += nt)         // J(r, c) at q
   {
      const int r = i/Q3, q = i % Q3;
      const int qq = q % QQ, qz = q/QQ;
      double u = 0.0, v = 0.0, w = 0.0;
      for (int dz = 0; dz < Dn; dz++)
      {
         const double bz = sB[qz*Dn + dz], gz = sG[qz*Dn + dz];
         u += s2[(3*r + 0)*DQQ + dz*QQ + qq]*bz;
         v += s2[(3*r + 1)*DQQ + dz*QQ + qq]*bz;
         w += s2[(3*r + 2)*DQQ + dz*QQ + qq]*gz;
      }
      sJ[(r*3 + 0)*Q3 + q] = u;
      sJ[(r*3 + 1)*Q3 + q] = v;
      sJ[(r*3 + 2)*Q3 + q] = w;
   }
   __syncthreads();
   sJ_out = sJ;
}

static size_t jac_lds_bytes(int Dn, int Q)
{
   return sizeof(double)*(size_t)(2*Q*Dn + 3*Dn*Dn*Dn + 6*Dn*Dn*Q + 9*Dn*Q*Q + 9*Q*Q*Q);
}

__global__ void __launch_bounds__(256)
k_jacobians(const int Dn, const int Q, const BasisT<MAX_DQ> bg,
            const double *__restrict__ nodesE, double *__restrict__ J)
{
   extern __shared__ __align__(16) double dyn[];
   const int64_t e = blockIdx.x;
   const int Q3 = Q*Q*Q;
   double *sJ;
   element_jacobians(Dn, Q, bg, nodesE + e*(int64_t)(3*Dn*Dn*Dn), dyn, sJ);
   // reference layout J(q, r, c, e): q + Q3*(r + 3*(c + 3*e))
   for (int i = threadIdx.x; i < 9*Q3; i += blockDim.x)
   {
      const int rc = i/Q3, q = i % Q3, r = rc/3, c = rc % 3;
      J[q + (int64_t)Q3*(r + 3*(c + 3*e))] = sJ[i];
   }
}

__device__ __forceinline__ void diffusion_point(const double J11, const double J21, const double J31,
                                                const double J12, const double J22, const double J32,
                                                const double J13, const double J23, const double J33,
                                                const double W, const double C1, const double C2,
                                                const double C3, double *out, double &detJ)
{
   detJ = J11*(J22*J33 - J32*J23) - J21*(J12*J33 - J32*J13) + J31*(J12*J23 - J22*J13);
   const double w_detJ = W/detJ;
   const double A11 = (J22*J33) - (J23*J32);
   const double A12 = (J32*J13) - (J12*J33);
   const double A13 = (J12*J23) - (J22*J13);
   const double A21 = (J31*J23) - (J21*J33);
   const double A22 = (J11*J33) - (J13*J31);
   const double A23 = (J21*J13) - (J11*J23);
   const double A31 = (J21*J32) - (J31*J22);
   const double A32 = (J31*J12) - (J11*J32);
   const double A33 = (J11*J22) - (J12*J21);
   out[0] = w_detJ*(C1*A11*A11 + C2*A12*A12 + C3*A13*A13);
   out[1] = w_detJ*(C1*A11*A21 + C2*A12*A22 + C3*A13*A23);
   out[2] = w_detJ*(C1*A11*A31 + C2*A12*A32 + C3*A13*A33);
   out[3] = w_detJ*(C1*A21*A21 + C2*A22*A22 + C3*A23*A23);
   out[4] = w_detJ*(C1*A21*A31 + C2*A22*A32 + C3*A23*A33);
   out[5] = w_detJ*(C1*A31*A31 + C2*A32*A32 + C3*A33*A33);
}

// Matrix coefficient (PADiffusionSetup3D, bilininteg_diffusion_kernels.cpp:297-348): c holds the
// 6 entries (1,1),(1,2),(1,3),(2,2),(2,3),(3,3) of a symmetric M, or the 9 of a full M as the
// reference's ProjectTranspose stores them; D = (W/detJ) adj(J) M adj(J)^T with the reference's
// grouping (R = M adj(J)^T first).  out: 6 planes (symmetric) or 9 row-major planes.
__device__ __forceinline__ void diffusion_point_matrix(const double J11, const double J21, const double J31,
                                                       const double J12, const double J22, const double J32,
                                                       const double J13, const double J23, const double J33,
                                                       const double W, const double *c, const bool symmetric,
                                                       double *out, double &detJ)
{
   detJ = J11*(J22*J33 - J32*J23) - J21*(J12*J33 - J32*J13) + J31*(J12*J23 - J22*J13);
   const double w_detJ = W/detJ;
   const double A11 = (J22*J33) - (J23*J32);
   const double A12 = (J32*J13) - (J12*J33);
   const double A13 = (J12*J23) - (J22*J13);
   const double A21 = (J31*J23) - (J21*J33);
   const double A22 = (J11*J33) - (J13*J31);
   const double A23 = (J21*J13) - (J11*J23);
   const double A31 = (J21*J32) - (J31*J22);
   const double A32 = (J31*J12) - (J11*J32);
   const double A33 = (J11*J22) - (J12*J21);
   const double M11 = c[0], M12 = c[1], M13 = c[2];
   const double M21 = symmetric ? M12 : c[3];
   const double M22 = symmetric ? c[3] : c[4];
   const double M23 = symmetric ? c[4] : c[5];
   const double M31 = symmetric ? M13 : c[6];
   const double M32 = symmetric ? M23 : c[7];
   const double M33 = symmetric ? c[5] : c[8];
   const double R11 = M11*A11 + M12*A12 + M13*A13;
   const double R12 = M11*A21 + M12*A22 + M13*A23;
   const double R13 = M11*A31 + M12*A32 + M13*A33;
   const double R21 = M21*A11 + M22*A12 + M23*A13;
   const double R22 = M21*A21 + M22*A22 + M23*A23;
   const double R23 = M21*A31 + M22*A32 + M23*A33;
   const double R31 = M31*A11 + M32*A12 + M33*A13;
   const double R32 = M31*A21 + M32*A22 + M33*A23;
   const double R33 = M31*A31 + M32*A32 + M33*A33;
   const double D11 = w_detJ*(A11*R11 + A12*R21 + A13*R31);
   const double D12 = w_detJ*(A11*R12 + A12*R22 + A13*R32);
   const double D13 = w_detJ*(A11*R13 + A12*R23 + A13*R33);
   const double D22 = w_detJ*(A21*R12 + A22*R22 + A23*R32);
   const double D23 = w_detJ*(A21*R13 + A22*R23 + A23*R33);
   const double D33 = w_detJ*(A31*R13 + A32*R23 + A33*R33);
   out[0] = D11; out[1] = D12; out[2] = D13;
   if (symmetric) { out[3] = D22; out[4] = D23; out[5] = D33; }
   else
   {
      out[3] = w_detJ*(A21*R11 + A22*R21 + A23*R31);
      out[4] = D22; out[5] = D23;
      out[6] = w_detJ*(A31*R11 + A32*R21 + A33*R31);
      out[7] = w_detJ*(A31*R12 + A32*R22 + A33*R32);
      out[8] = D33;
   }
}

// kind 0: mass, 1: diffusion.  One workgroup per element.
__global__ void __launch_bounds__(256)
k_setup_from_nodes(const int kind, const int Dn, const int Q, const int coeffDim,
                   const BasisT<MAX_DQ> bg, const double *__restrict__ W,
                   const double *__restrict__ nodesE, const double *__restrict__ C,
                   const int const_c, double *__restrict__ D, double *__restrict__ detJ_out,
                   const int layout)
{
   extern __shared__ __align__(16) double dyn[];
   const int64_t e = blockIdx.x;
   const int Q3 = Q*Q*Q;
   double *sJ;
   element_jacobians(Dn, Q, bg, nodesE + e*(int64_t)(3*Dn*Dn*Dn), dyn, sJ);
   for (int q = threadIdx.x; q < Q3; q += blockDim.x)
   {
      const double J11 = sJ[0*Q3 + q], J12 = sJ[1*Q3 + q], J13 = sJ[2*Q3 + q];
      const double J21 = sJ[3*Q3 + q], J22 = sJ[4*Q3 + q], J23 = sJ[5*Q3 + q];
      const double J31 = sJ[6*Q3 + q], J32 = sJ[7*Q3 + q], J33 = sJ[8*Q3 + q];
      const double *c = const_c ? C : C + coeffDim*(q + (int64_t)Q3*e);
      double out[9], detJ;
      if (kind)
      {
         const int nc = coeffDim == 9 ? 9 : 6;
         if (coeffDim >= 6)
         {
            diffusion_point_matrix(J11, J21, J31, J12, J22, J32, J13, J23, J33, W[q], c, coeffDim == 6, out, detJ);
         }
         else
         {
            const double C1 = c[0], C2 = c[coeffDim == 3 ? 1 : 0], C3 = c[coeffDim == 3 ? 2 : 0];
            diffusion_point(J11, J21, J31, J12, J22, J32, J13, J23, J33, W[q], C1, C2, C3, out, detJ);
         }
         for (int k = 0; k < nc; k++) { D[layout ? blocked_index(1, Q, e, q, k) : q + (int64_t)Q3*(k + nc*e)] = out[k]; }
      }
      else
      {
         detJ = J11*(J22*J33 - J32*J23) - J21*(J12*J33 - J32*J13) + J31*(J12*J23 - J22*J13);
         D[layout ? blocked_index(0, Q, e, q, 0) : q + (int64_t)Q3*e] = W[q]*c[0]*detJ;
      }
      if (detJ_out) { detJ_out[q + (int64_t)Q3*e] = detJ; }
   }
}

// MFEM layout (q, c, e) -> blocked layout (common.h); padding entries are left untouched.
__global__ void __launch_bounds__(256)
k_to_blocked(const int kind, const int Q, const int64_t n, const double *__restrict__ Din,
             double *__restrict__ Dout)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int Q3 = Q*Q*Q, nc = kind ? 6 : 1;
   const int q = (int)(i % Q3), c = (int)((i/Q3) % nc);
   const int64_t e = i/((int64_t)Q3*nc);
   Dout[blocked_index(kind, Q, e, q, c)] = Din[i];
}

// blocked layout -> MFEM layout (the inverse permutation)
__global__ void __launch_bounds__(256)
k_from_blocked(const int kind, const int Q, const int64_t n, const double *__restrict__ Dblk,
               double *__restrict__ Dout)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int Q3 = Q*Q*Q, nc = kind ? 6 : 1;
   const int q = (int)(i % Q3), c = (int)((i/Q3) % nc);
   const int64_t e = i/((int64_t)Q3*nc);
   Dout[i] = Dblk[blocked_index(kind, Q, e, q, c)];
}

// Domain linear form, one workgroup per element (runtime D, Q):
// yE(dx,dy,dz,e) += sum_q B(qx,dx) B(qy,dy) B(qz,dz) F(q,e), F = W*f*detJ at the points,
// contracted x, then y, then z through LDS.
__global__ void __launch_bounds__(256)
k_domain_lf(const int D, const int Q, const BasisT<MAX_DQ> bg, const double *__restrict__ F,
            double *__restrict__ Y)
{
   extern __shared__ __align__(16) double dyn[];
   const int Q3 = Q*Q*Q, D3 = D*D*D;
   double *sB = dyn, *s0 = sB + Q*D, *s1 = s0 + Q3, *s2 = s1 + D*Q*Q;
   const int tid = threadIdx.x, nt = blockDim.x;
   const int64_t e = blockIdx.x;
   for (int i = tid; i < Q*D; i += nt) { sB[i] = bg.b[i]; }
   for (int i = tid; i < Q3; i += nt) { s0[i] = F[e*Q3 + i]; }
   __syncthreads();
   for (int i = tid; i < D*Q*Q; i += nt)         // [dx][qy][qz]
   {
      const int dx = i % D, qy = (i/D) % Q, qz = i/(D*Q);
      double s = 0.0;
      for (int qx = 0; qx < Q; qx++) { s += sB[qx*D + dx]*s0[qx + Q*(qy + Q*qz)]; }
      s1[i] = s;
   }
   __syncthreads();
   for (int i = tid; i < D*D*Q; i += nt)         // [dx][dy][qz]
   {
      const int dx = i % D, dy = (i/D) % D, qz = i/(D*D);
      double s = 0.0;
      for (int qy = 0; qy < Q; qy++) { s += sB[qy*D + dy]*s1[dx + D*(qy + Q*qz)]; }
      s2[i] = s;
   }
   __syncthreads();
   for (int i = tid; i < D3; i += nt)
   {
      const int dxy = i % (D*D), dz = i/(D*D);
      double s = 0.0;
      for (int qz = 0; qz < Q; qz++) { s += sB[qz*D + dz]*s2[dxy + D*D*qz]; }
      Y[e*D3 + i] += s;
   }
}

__global__ void __launch_bounds__(256)
k_det(const int64_t n, const int Q3, const double *__restrict__ J, double *__restrict__ detJ)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int64_t e = i/Q3, q = i % Q3;
   const double *Je = J + 9*(int64_t)Q3*e + q;
   const double J11 = Je[0], J21 = Je[Q3], J31 = Je[2*(int64_t)Q3];
   const double J12 = Je[3*(int64_t)Q3], J22 = Je[4*(int64_t)Q3], J32 = Je[5*(int64_t)Q3];
   const double J13 = Je[6*(int64_t)Q3], J23 = Je[7*(int64_t)Q3], J33 = Je[8*(int64_t)Q3];
   detJ[i] = J11*(J22*J33 - J32*J23) - J21*(J12*J33 - J32*J13) + J31*(J12*J23 - J22*J13);
}

__global__ void __launch_bounds__(256)
k_diffusion_setup(const int64_t n, const int Q3, const int coeffDim,
                  const double *__restrict__ W, const double *__restrict__ J,
                  const double *__restrict__ C, const int const_c, double *__restrict__ D)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const int64_t e = i/Q3, q = i % Q3;
   const double *Je = J + 9*(int64_t)Q3*e + q;
   const double J11 = Je[0], J21 = Je[Q3], J31 = Je[2*(int64_t)Q3];
   const double J12 = Je[3*(int64_t)Q3], J22 = Je[4*(int64_t)Q3], J32 = Je[5*(int64_t)Q3];
   const double J13 = Je[6*(int64_t)Q3], J23 = Je[7*(int64_t)Q3], J33 = Je[8*(int64_t)Q3];
   const double *c = const_c ? C : C + coeffDim*i;
   double out[9], detJ;
   const int nc = coeffDim == 9 ? 9 : 6;
   if (coeffDim >= 6)
   {
      diffusion_point_matrix(J11, J21, J31, J12, J22, J32, J13, J23, J33, W[q], c, coeffDim == 6, out, detJ);
   }
   else
   {
      const double C1 = c[0], C2 = c[coeffDim == 3 ? 1 : 0], C3 = c[coeffDim == 3 ? 2 : 0];
      diffusion_point(J11, J21, J31, J12, J22, J32, J13, J23, J33, W[q], C1, C2, C3, out, detJ);
   }
   for (int k = 0; k < nc; k++) { D[q + (int64_t)Q3*(k + nc*e)] = out[k]; }
}

__global__ void __launch_bounds__(256)
k_mass_setup(const int64_t n, const int Q3, const double *__restrict__ W,
             const double *__restrict__ detJ, const double *__restrict__ C, const int const_c,
             double *__restrict__ D)
{
   const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
   if (i >= n) { return; }
   const double coeff = const_c ? C[0] : C[i];
   D[i] = W[i % Q3]*coeff*detJ[i];
}

// Element diagonals, one workgroup per element (runtime D, Q).  Same contraction order as
// the reference kernels (z, then y, then x; 9 (i,j) passes for diffusion).
template <int KIND>
__global__ void __launch_bounds__(256)
k_diag(const int D, const int Q, const BasisT<MAX_DQ> bg, const double *__restrict__ Dq,
       double *__restrict__ Y, const int layout, const int nonsym)
{
   extern __shared__ __align__(16) double dyn[];
   const int Q3 = Q*Q*Q, QQ = Q*Q, D3 = D*D*D;
   double *sB = dyn, *sG = sB + Q*D, *QQD = sG + Q*D, *QDD = QQD + Q*Q*D;
   const int tid = threadIdx.x, nt = blockDim.x;
   const int64_t e = blockIdx.x;
   for (int i = tid; i < Q*D; i += nt) { sB[i] = bg.b[i]; sG[i] = bg.g[i]; }
   __syncthreads();
   const int npass = KIND ? 9 : 1;
   for (int pass = 0; pass < npass; pass++)
   {
      const int ii = pass/3, jj = pass % 3;
      // symmetric: the reference's ksym map; full D: k = 3 i + j (bilininteg_diffusion_kernels.hpp:428-431)
      const int ksym = KIND ? (jj >= ii ? 3 - (3 - ii)*(2 - ii)/2 + jj : 3 - (3 - jj)*(2 - jj)/2 + ii) : 0;
      const int k = nonsym ? 3*ii + jj : ksym;
      const double *De = Dq + (int64_t)Q3*(k + (KIND ? (nonsym ? 9 : 6) : 1)*e);   // layout 0
      for (int i = tid; i < QQ*D; i += nt)        // [qx][qy][dz]
      {
         const int dz = i % D, qy = (i/D) % Q, qx = i/(D*Q);
         double s = 0.0;
         for (int qz = 0; qz < Q; qz++)
         {
            const int q = qx + Q*(qy + Q*qz);
            const double O = layout ? Dq[blocked_index(KIND, Q, e, q, k)] : De[q];
            const double Bz = sB[qz*D + dz], Gz = sG[qz*D + dz];
            const double L = (KIND && ii == 2) ? Gz : Bz, R = (KIND && jj == 2) ? Gz : Bz;
            s += L*O*R;
         }
         QQD[i] = s;
      }
      __syncthreads();
      for (int i = tid; i < Q*D*D; i += nt)       // [qx][dy][dz]
      {
         const int dz = i % D, dy = (i/D) % D, qx = i/(D*D);
         double s = 0.0;
         for (int qy = 0; qy < Q; qy++)
         {
            const double By = sB[qy*D + dy], Gy = sG[qy*D + dy];
            const double L = (KIND && ii == 1) ? Gy : By, R = (KIND && jj == 1) ? Gy : By;
            s += L*QQD[(qx*Q + qy)*D + dz]*R;
         }
         QDD[i] = s;
      }
      __syncthreads();
      for (int i = tid; i < D3; i += nt)
      {
         const int dx = i % D, dy = (i/D) % D, dz = i/(D*D);
         double s = 0.0;
         for (int qx = 0; qx < Q; qx++)
         {
            const double Bx = sB[qx*D + dx], Gx = sG[qx*D + dx];
            const double L = (KIND && ii == 0) ? Gx : Bx, R = (KIND && jj == 0) ? Gx : Bx;
            s += L*QDD[(qx*D + dy)*D + dz]*R;
         }
         Y[e*D3 + i] += s;
      }
      __syncthreads();
   }
}

static int fill_basis_dyn(BasisT<MAX_DQ> &bg, int D, int Q, const double *B, const double *G)
{
   MFA_REQUIRE(D >= 1 && D <= MAX_D1D && Q >= 1 && Q <= MAX_Q1D && D*Q <= MAX_DQ,
               "unsupported (D1D,Q1D) = (%d,%d); limits D1D<=%d, Q1D<=%d, D1D*Q1D<=%d", D, Q,
               MAX_D1D, MAX_Q1D, MAX_DQ);
   MFA_REQUIRE(B, "null basis table");
   for (int q = 0; q < Q; q++)
      for (int d = 0; d < D; d++)
      {
         bg.b[q*D + d] = B[q + Q*d];
         bg.g[q*D + d] = G ? G[q + Q*d] : 0.0;
      }
   return 0;
}

template <typename K>
static int set_lds(K kern, size_t lds)
{
   MFA_REQUIRE(lds <= 160*1024, "setup kernel needs %zu B of LDS (> 160 KiB)", lds);
   if (lds > 64*1024)
   {
      return check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "hipFuncSetAttribute");
   }
   return 0;
}

} // namespace mfa

using namespace mfa;

extern "C" int mfem_amd_geom_jacobians_3d(int Dn, int Q1D, int64_t ne, const double *Bn,
                                          const double *Gn, const double *nodesE, double *J,
                                          void *stream)
{
   if (ne == 0) { return 0; }
   MFA_REQUIRE(ne > 0 && ne < 2147483647 && nodesE && J && Gn, "geom_jacobians_3d: bad arguments");
   BasisT<MAX_DQ> bg;
   if (int rc = fill_basis_dyn(bg, Dn, Q1D, Bn, Gn)) { return rc; }
   const size_t lds = jac_lds_bytes(Dn, Q1D);
   if (int rc = set_lds(k_jacobians, lds)) { return rc; }
   k_jacobians<<<dim3((unsigned)ne), dim3(256), lds, as_stream(stream)>>>(Dn, Q1D, bg, nodesE, J);
   MFA_CHECK_LAUNCH("k_jacobians");
   return 0;
}

extern "C" int mfem_amd_geom_det_3d(int Q1D, int64_t ne, const double *J, double *detJ,
                                    void *stream)
{
   const int64_t n = ne*Q1D*Q1D*Q1D;
   if (n == 0) { return 0; }
   MFA_REQUIRE(n > 0 && J && detJ, "geom_det_3d: bad arguments");
   k_det<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, as_stream(stream)>>>(n, Q1D*Q1D*Q1D, J, detJ);
   MFA_CHECK_LAUNCH("k_det");
   return 0;
}

extern "C" int mfem_amd_diffusion_setup_3d(int Q1D, int coeffDim, int64_t ne, const double *W,
                                           const double *J, const double *C, int const_c,
                                           double *D, void *stream)
{
   const int64_t n = ne*Q1D*Q1D*Q1D;
   if (n == 0) { return 0; }
   MFA_REQUIRE(Q1D >= 1 && Q1D <= MAX_Q1D, "diffusion_setup_3d: Q1D = %d out of range", Q1D);
   MFA_REQUIRE(coeffDim == 1 || coeffDim == 3 || coeffDim == 6 || coeffDim == 9,
               "diffusion_setup_3d: coeffDim = %d (1 scalar, 3 vector, 6 symmetric matrix, 9 matrix)", coeffDim);
   MFA_REQUIRE(n > 0 && W && J && C && D, "diffusion_setup_3d: bad arguments");
   const double *dW = W;
   k_diffusion_setup<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, as_stream(stream)>>>(
      n, Q1D*Q1D*Q1D, coeffDim, dW, J, C, const_c, D);
   MFA_CHECK_LAUNCH("k_diffusion_setup");
   return 0;
}

extern "C" int mfem_amd_mass_setup_3d(int Q1D, int64_t ne, const double *W, const double *detJ,
                                      const double *C, int const_c, double *D, void *stream)
{
   const int64_t n = ne*Q1D*Q1D*Q1D;
   if (n == 0) { return 0; }
   MFA_REQUIRE(Q1D >= 1 && Q1D <= MAX_Q1D, "mass_setup_3d: Q1D = %d out of range", Q1D);
   MFA_REQUIRE(n > 0 && W && detJ && C && D, "mass_setup_3d: bad arguments");
   const double *dW = W;
   k_mass_setup<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, as_stream(stream)>>>(
      n, Q1D*Q1D*Q1D, dW, detJ, C, const_c, D);
   MFA_CHECK_LAUNCH("k_mass_setup");
   return 0;
}

extern "C" int mfem_amd_pa_setup_from_nodes_3d(int kind, int Dn, int Q1D, int coeffDim,
                                               int64_t ne, const double *Bn, const double *Gn,
                                               const double *W, const double *nodesE,
                                               const double *C, int const_c, int pa_layout,
                                               double *D, double *detJ_out, void *stream)
{
   if (ne == 0) { return 0; }
   const TraceRange range("mfem_amd: PA setup (nodes -> J -> D)");
   MFA_REQUIRE(kind == 0 || kind == 1, "pa_setup_from_nodes_3d: kind must be 0 or 1");
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_setup_from_nodes_3d: pa_layout must be 0 or 1");
   MFA_REQUIRE(coeffDim == 1 || (kind == 1 && (coeffDim == 3 || coeffDim == 6 || coeffDim == 9)),
               "pa_setup_from_nodes_3d: coeffDim = %d unsupported", coeffDim);
   MFA_REQUIRE(!(coeffDim == 9 && pa_layout), "pa_setup_from_nodes_3d: the 9-plane D of a full matrix coefficient has the MFEM layout only");
   MFA_REQUIRE(ne > 0 && ne < 2147483647 && W && nodesE && C && D && Gn, "pa_setup_from_nodes_3d: bad arguments");
   BasisT<MAX_DQ> bg;
   if (int rc = fill_basis_dyn(bg, Dn, Q1D, Bn, Gn)) { return rc; }
   const double *dW = W;
   const size_t lds = jac_lds_bytes(Dn, Q1D);
   if (int rc = set_lds(k_setup_from_nodes, lds)) { return rc; }
   k_setup_from_nodes<<<dim3((unsigned)ne), dim3(256), lds, as_stream(stream)>>>(
      kind, Dn, Q1D, coeffDim, bg, dW, nodesE, C, const_c, D, detJ_out, pa_layout);
   MFA_CHECK_LAUNCH("k_setup_from_nodes");
   return 0;
}

extern "C" int mfem_amd_pa_data_to_blocked(int kind, int Q1D, int64_t ne, const double *D,
                                           double *Dblocked, void *stream)
{
   if (ne == 0) { return 0; }
   MFA_REQUIRE((kind == 0 || kind == 1) && Q1D >= 1 && Q1D <= MAX_Q1D && D && Dblocked && D != Dblocked,
               "pa_data_to_blocked: bad arguments");
   const int64_t n = ne*(int64_t)(kind ? 6 : 1)*Q1D*Q1D*Q1D;
   k_to_blocked<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, as_stream(stream)>>>(kind, Q1D, n, D, Dblocked);
   MFA_CHECK_LAUNCH("k_to_blocked");
   return 0;
}

extern "C" int mfem_amd_pa_data_from_blocked(int kind, int Q1D, int64_t ne, const double *Dblocked,
                                             double *D, void *stream)
{
   if (ne == 0) { return 0; }
   MFA_REQUIRE((kind == 0 || kind == 1) && Q1D >= 1 && Q1D <= MAX_Q1D && D && Dblocked && D != Dblocked,
               "pa_data_from_blocked: bad arguments");
   const int64_t n = ne*(int64_t)(kind ? 6 : 1)*Q1D*Q1D*Q1D;
   k_from_blocked<<<dim3((unsigned)((n + 255)/256)), dim3(256), 0, as_stream(stream)>>>(kind, Q1D, n, Dblocked, D);
   MFA_CHECK_LAUNCH("k_from_blocked");
   return 0;
}

extern "C" int mfem_amd_domain_lf_3d(int D1D, int Q1D, int64_t ne, const double *B,
                                     const double *F, double *yE, void *stream)
{
   if (ne == 0) { return 0; }
   MFA_REQUIRE(ne > 0 && ne < 2147483647 && F && yE, "domain_lf_3d: bad arguments");
   BasisT<MAX_DQ> bg;
   if (int rc = fill_basis_dyn(bg, D1D, Q1D, B, nullptr)) { return rc; }
   const size_t lds = sizeof(double)*(size_t)(Q1D*D1D + Q1D*Q1D*Q1D + D1D*Q1D*Q1D + D1D*D1D*Q1D);
   if (int rc = set_lds(k_domain_lf, lds)) { return rc; }
   k_domain_lf<<<dim3((unsigned)ne), dim3(256), lds, as_stream(stream)>>>(D1D, Q1D, bg, F, yE);
   MFA_CHECK_LAUNCH("k_domain_lf");
   return 0;
}

extern "C" int mfem_amd_pa_diffusion_diag_3d(int D1D, int Q1D, int64_t ne, int symmetric,
                                             const double *B, const double *G, const double *D,
                                             int pa_layout, double *diagE, void *stream)
{
   if (ne == 0) { return 0; }
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_diffusion_diag_3d: pa_layout must be 0 or 1");
   MFA_REQUIRE(symmetric || pa_layout == 0, "pa_diffusion_diag_3d: the non-symmetric (9-plane) D has the MFEM layout only");
   MFA_REQUIRE(ne > 0 && ne < 2147483647 && D && diagE && G, "pa_diffusion_diag_3d: bad arguments");
   BasisT<MAX_DQ> bg;
   if (int rc = fill_basis_dyn(bg, D1D, Q1D, B, G)) { return rc; }
   const size_t lds = sizeof(double)*(size_t)(2*Q1D*D1D + Q1D*Q1D*D1D + Q1D*D1D*D1D);
   if (int rc = set_lds(k_diag<1>, lds)) { return rc; }
   k_diag<1><<<dim3((unsigned)ne), dim3(256), lds, as_stream(stream)>>>(D1D, Q1D, bg, D, diagE, pa_layout, symmetric ? 0 : 1);
   MFA_CHECK_LAUNCH("k_diag<diffusion>");
   return 0;
}

extern "C" int mfem_amd_pa_mass_diag_3d(int D1D, int Q1D, int64_t ne, const double *B,
                                        const double *D, int pa_layout, double *diagE,
                                        void *stream)
{
   if (ne == 0) { return 0; }
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_mass_diag_3d: pa_layout must be 0 or 1");
   MFA_REQUIRE(ne > 0 && ne < 2147483647 && D && diagE, "pa_mass_diag_3d: bad arguments");
   BasisT<MAX_DQ> bg;
   if (int rc = fill_basis_dyn(bg, D1D, Q1D, B, nullptr)) { return rc; }
   const size_t lds = sizeof(double)*(size_t)(2*Q1D*D1D + Q1D*Q1D*D1D + Q1D*D1D*D1D);
   if (int rc = set_lds(k_diag<0>, lds)) { return rc; }
   k_diag<0><<<dim3((unsigned)ne), dim3(256), lds, as_stream(stream)>>>(D1D, Q1D, bg, D, diagE, pa_layout, 0);
   MFA_CHECK_LAUNCH("k_diag<mass>");
   return 0;
}

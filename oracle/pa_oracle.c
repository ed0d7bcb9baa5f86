/* TEST INFRASTRUCTURE -- CPU oracle for the MFEM partial-assembly + Jacobi-PCG path.
 *
 * Plain-C restatement of the reference's `-d cpu` algorithm (loop nests and summation
 * order of the reference's host execution).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (mfem_amd/) never does.
 *
 * Checked against: the tests/golden npz files (arrays dumped from the reference itself, see
 * oracle/ref_dump.cpp) and the known-answer table in BASELINE.md section 2.
 * PARITY UNPINNED by this build's rules: those arrays came from a serial libmfem.a that the survey stage had
 * produced with the reference's own build system (cmake + generated config/_config.hpp); the reference cannot
 * be built from its sources alone here, it holds no golden vectors of its own for this path, and the
 * generating binary is gone -- the fixtures are kept, not reproducible.  The independent dense assembly of
 * tests/full_assembly.py (the reference's own test method, PA == assembled matrix) is the second check.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).  All sizes that can exceed 2^31 are int64_t.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXQ 16

/* ---------------------------------------------------------------------------------
 * Quadrature (fem/intrules.cpp:424-510 GaussLegendre, :512-642 GaussLobatto), points and
 * weights on [0,1].
 * ------------------------------------------------------------------------------ */
void orc_gauss_legendre(int np, double *x, double *w)
{
   if (np == 1) { x[0] = 0.5; w[0] = 1.0; return; }
   if (np == 2)
   {
      x[0] = 0.21132486540518711775; x[1] = 0.78867513459481288225;
      w[0] = w[1] = 0.5; return;
   }
   if (np == 3)
   {
      x[0] = 0.11270166537925831148; x[1] = 0.5; x[2] = 0.88729833462074168852;
      w[0] = w[2] = 5./18.; w[1] = 4./9.; return;
   }
   const int n = np, m = (n + 1)/2;
   for (int i = 1; i <= m; i++)
   {
      double z = cos(M_PI*(i - 0.25)/(n + 0.5));
      double pp, p1, dz, xi = 0.;
      int done = 0;
      while (1)
      {
         double p2 = 1;
         p1 = z;
         for (int j = 2; j <= n; j++)
         {
            double p3 = p2;
            p2 = p1;
            p1 = ((2*j - 1)*z*p2 - (j - 1)*p3)/j;
         }
         pp = n*(z*p1 - p2)/(z*z - 1);
         if (done) { break; }
         dz = p1/pp;
         if (fabs(dz) < 1e-16)
         {
            done = 1;
            xi = ((1 - z) + dz)/2;
         }
         z -= dz;
      }
      x[i-1] = xi;
      x[n-i] = 1 - xi;
      w[i-1] = w[n-i] = 1./(4*xi*(1 - xi)*pp*pp);
   }
}

void orc_gauss_lobatto(int np, double *x, double *w)
{
   if (np == 1) { x[0] = 0.5; w[0] = 1.0; return; }
   x[0] = 0.0; x[np-1] = 1.0;
   w[0] = w[np-1] = 1.0/(np*(np - 1));
   for (int i = 1; i <= (np - 1)/2; ++i)
   {
      double x_i = sin(M_PI*((double)(i)/(np - 1) - 0.5));
      double z_i = 0., p_l;
      int done = 0;
      for (int iter = 0; 1; ++iter)
      {
         double p_lm1 = 1.0;
         p_l = x_i;
         for (int l = 1; l < (np - 1); ++l)
         {
            double p_lp1 = ((2*l + 1)*x_i*p_l - l*p_lm1)/(l + 1);
            p_lm1 = p_l;
            p_l = p_lp1;
         }
         if (done) { break; }
         double dx = (x_i*p_l - p_lm1)/(np*p_l);
         if (fabs(dx) < 1e-16)
         {
            done = 1;
            z_i = ((1.0 + x_i) - dx)/2;
         }
         if (iter >= 8) { break; }
         x_i -= dx;
      }
      x[i] = z_i;
      w[i] = 1.0/(np*(np - 1)*p_l*p_l);
      x[np-1-i] = 1.0 - z_i;
      w[np-1-i] = w[i];
   }
}

/* Tensor rule weights, x fastest (fem/intrules.cpp:56-84). */
void orc_tensor_weights_3d(int Q, const double *w1, double *W)
{
   for (int iz = 0; iz < Q; iz++)
      for (int iy = 0; iy < Q; iy++)
         for (int ix = 0; ix < Q; ix++)
         { W[(iz*Q + iy)*Q + ix] = w1[ix]*w1[iy]*w1[iz]; }
}

/* ---------------------------------------------------------------------------------
 * 1D Lagrange basis on `nodes` (p+1 increasing points), barycentric form with
 * derivative (fem/fe/fe_base.cpp:1754-1772 weights, :1861-1928 Eval), tabulated at nq
 * points into the DofToQuad TENSOR layout B[q + nq*d], G[q + nq*d]
 * (fem/fe/fe_base.cpp:2622-2673).
 * ------------------------------------------------------------------------------ */
void orc_basis_1d(int p, const double *nodes, int nq, const double *qpts,
                  double *B, double *G)
{
   double w[MAXQ + 1], u[MAXQ + 1], d[MAXQ + 1];
   const double *x = nodes;
   for (int i = 0; i <= p; i++) { w[i] = 1.0; }
   for (int i = 0; i <= p; i++)
   {
      for (int j = 0; j < i; j++)
      {
         double xij = x[i] - x[j];
         w[i] *=  xij;
         w[j] *= -xij;
      }
   }
   for (int i = 0; i <= p; i++) { w[i] = 1.0/w[i]; }

   for (int q = 0; q < nq; q++)
   {
      const double y = qpts[q];
      if (p == 0) { u[0] = 1.0; d[0] = 0.0; }
      else
      {
         int i, k;
         double l, lp, lk, sk, si;
         lk = 1.0;
         for (k = 0; k < p; k++)
         {
            if (y >= (x[k] + x[k+1])/2) { lk *= y - x[k]; }
            else
            {
               for (i = k + 1; i <= p; i++) { lk *= y - x[i]; }
               break;
            }
         }
         l = lk*(y - x[k]);
         sk = 0.0;
         for (i = 0; i < k; i++)
         {
            si = 1.0/(y - x[i]);
            sk += si;
            u[i] = l*si*w[i];
         }
         u[k] = lk*w[k];
         for (i++; i <= p; i++)
         {
            si = 1.0/(y - x[i]);
            sk += si;
            u[i] = l*si*w[i];
         }
         lp = l*sk + lk;
         for (i = 0; i < k; i++) { d[i] = (lp*w[i] - u[i])/(y - x[i]); }
         d[k] = sk*u[k];
         for (i++; i <= p; i++) { d[i] = (lp*w[i] - u[i])/(y - x[i]); }
      }
      for (int dd = 0; dd <= p; dd++)
      {
         B[q + nq*dd] = u[dd];
         G[q + nq*dd] = d[dd];
      }
   }
}

/* ---------------------------------------------------------------------------------
 * ElementRestriction constructor (fem/restriction.cpp:26-107): gather_map, offsets,
 * indices from the native element->dof table and the lexicographic dof_map.
 * Integer work: must match the reference bit for bit.
 * ------------------------------------------------------------------------------ */
void orc_restriction_build(int ne, int dof, int ndofs, const int *element_map,
                           const int *dof_map, int *gather_map, int *offsets,
                           int *indices)
{
   for (int i = 0; i <= ndofs; ++i) { offsets[i] = 0; }
   for (int e = 0; e < ne; ++e)
   {
      for (int d = 0; d < dof; ++d)
      {
         const int sgid = element_map[dof*e + d];
         const int gid = (sgid >= 0) ? sgid : -1 - sgid;
         ++offsets[gid + 1];
      }
   }
   for (int i = 1; i <= ndofs; ++i) { offsets[i] += offsets[i - 1]; }
   for (int e = 0; e < ne; ++e)
   {
      for (int d = 0; d < dof; ++d)
      {
         const int sdid = dof_map ? dof_map[d] : 0;
         const int did = (!dof_map) ? d : (sdid >= 0 ? sdid : -1 - sdid);
         const int sgid = element_map[dof*e + did];
         const int gid = (sgid >= 0) ? sgid : -1 - sgid;
         const int lid = dof*e + d;
         const int plus = (sgid >= 0 && sdid >= 0) || (sgid < 0 && sdid < 0);
         gather_map[lid] = plus ? gid : -1 - gid;
         indices[offsets[gid]++] = plus ? lid : -1 - lid;
      }
   }
   for (int i = ndofs; i > 0; --i) { offsets[i] = offsets[i - 1]; }
   offsets[0] = 0;
}

/* ElementRestriction::Mult (fem/restriction.cpp:109-129), vdim components stored
 * byNODES (t=0: x(j,c) = x[j + c*ndofs]) or byVDIM (t=1: x[c + vd*j]). */
void orc_restriction_mult(int64_t nedofs, int nd, int vd, int byvdim, int64_t ndofs,
                          const int *gather_map, const double *x, double *y)
{
   #pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < nedofs; i++)
   {
      const int gid = gather_map[i];
      const int plus = gid >= 0;
      const int64_t j = plus ? gid : -1 - gid;
      for (int c = 0; c < vd; ++c)
      {
         const double v = byvdim ? x[c + (int64_t)vd*j] : x[j + c*ndofs];
         y[(i % nd) + nd*(c + (int64_t)vd*(i/nd))] = plus ? v : -v;
      }
   }
}

/* ElementRestriction::TAddMultTranspose<ADD> / AbsMultTranspose
 * (fem/restriction.cpp:152-180, 196-221), scalar (vdim = 1).  Sequential left-to-right
 * sum per L-dof starting from 0. */
void orc_restriction_mult_transpose(int64_t ndofs, const int *offsets, const int *indices,
                                    const double *xE, double *y, int add, int absv)
{
   #pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < ndofs; i++)
   {
      const int offset = offsets[i], next_offset = offsets[i + 1];
      double dof_value = 0;
      for (int j = offset; j < next_offset; ++j)
      {
         const int idx_j = (indices[j] >= 0) ? indices[j] : -1 - indices[j];
         dof_value += (indices[j] >= 0 || absv) ? xE[idx_j] : -xE[idx_j];
      }
      if (add) { y[i] += dof_value; }
      else { y[i] = dof_value; }
   }
}

/* ---------------------------------------------------------------------------------
 * Geometric factors: E-nodes (Dn,Dn,Dn,3,NE) -> J(Q,Q,Q,3,3,NE), J(q,r,c,e) = dx_r/dxi_c
 * (mesh/mesh.cpp:15178-15231; fem/qinterp/grad.hpp:232-374, byNODES layout, no
 * physical-gradient transform).
 * ------------------------------------------------------------------------------ */
void orc_jacobians_3d(int Dn, int Q, int64_t ne, const double *B, const double *G,
                      const double *xE, double *J)
{
   const int64_t D3 = (int64_t)Dn*Dn*Dn, Q3 = (int64_t)Q*Q*Q;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      double *bx = (double*)malloc(sizeof(double)*2*Dn*Dn*Q);   /* (qx,dy,dz): B and G along x */
      double *by = (double*)malloc(sizeof(double)*3*Dn*Q*Q);    /* (qx,qy,dz): 3 combos */
      double *gx = bx + Dn*Dn*Q;
      for (int r = 0; r < 3; r++)
      {
         const double *X = xE + (e*3 + r)*D3;
         for (int dz = 0; dz < Dn; dz++)
            for (int dy = 0; dy < Dn; dy++)
               for (int qx = 0; qx < Q; qx++)
               {
                  double u = 0, v = 0;
                  for (int dx = 0; dx < Dn; dx++)
                  {
                     const double s = X[dx + Dn*(dy + Dn*dz)];
                     u += B[qx + Q*dx]*s;
                     v += G[qx + Q*dx]*s;
                  }
                  bx[qx + Q*(dy + Dn*dz)] = u;
                  gx[qx + Q*(dy + Dn*dz)] = v;
               }
         for (int dz = 0; dz < Dn; dz++)
            for (int qy = 0; qy < Q; qy++)
               for (int qx = 0; qx < Q; qx++)
               {
                  double u = 0, v = 0, w = 0;
                  for (int dy = 0; dy < Dn; dy++)
                  {
                     u += gx[qx + Q*(dy + Dn*dz)]*B[qy + Q*dy];
                     v += bx[qx + Q*(dy + Dn*dz)]*G[qy + Q*dy];
                     w += bx[qx + Q*(dy + Dn*dz)]*B[qy + Q*dy];
                  }
                  by[0*Dn*Q*Q + qx + Q*(qy + Q*dz)] = u;
                  by[1*Dn*Q*Q + qx + Q*(qy + Q*dz)] = v;
                  by[2*Dn*Q*Q + qx + Q*(qy + Q*dz)] = w;
               }
         for (int qz = 0; qz < Q; qz++)
            for (int qy = 0; qy < Q; qy++)
               for (int qx = 0; qx < Q; qx++)
               {
                  double u = 0, v = 0, w = 0;
                  for (int dz = 0; dz < Dn; dz++)
                  {
                     u += by[0*Dn*Q*Q + qx + Q*(qy + Q*dz)]*B[qz + Q*dz];
                     v += by[1*Dn*Q*Q + qx + Q*(qy + Q*dz)]*B[qz + Q*dz];
                     w += by[2*Dn*Q*Q + qx + Q*(qy + Q*dz)]*G[qz + Q*dz];
                  }
                  const int64_t q = qx + Q*(qy + (int64_t)Q*qz);
                  J[q + Q3*(r + 3*(0 + 3*e))] = u;
                  J[q + Q3*(r + 3*(1 + 3*e))] = v;
                  J[q + Q3*(r + 3*(2 + 3*e))] = w;
               }
      }
      free(bx); free(by);
   }
}

/* detJ(q,e) (fem/qinterp/det.hpp:209-282 -> kernels::Det<3>). */
void orc_det_3d(int Q, int64_t ne, const double *J, double *detJ)
{
   const int64_t Q3 = (int64_t)Q*Q*Q;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      for (int64_t q = 0; q < Q3; q++)
      {
         const double *Je = J + 9*Q3*e + q;
         const double J11 = Je[0*Q3], J21 = Je[1*Q3], J31 = Je[2*Q3];
         const double J12 = Je[3*Q3], J22 = Je[4*Q3], J32 = Je[5*Q3];
         const double J13 = Je[6*Q3], J23 = Je[7*Q3], J33 = Je[8*Q3];
         detJ[q + Q3*e] = J11*(J22*J33 - J32*J23) - J21*(J12*J33 - J32*J13) +
                          J31*(J12*J23 - J22*J13);
      }
   }
}

/* ---------------------------------------------------------------------------------
 * PADiffusionSetup3D (fem/integ/bilininteg_diffusion_kernels.cpp:243-367).
 * coeffDim in {1,3} (scalar / vector coefficient), symmetric 6-component output
 * D(q,k,e), k order (1,1),(2,1),(3,1),(2,2),(3,2),(3,3); coeffDim 6 (symmetric matrix
 * coefficient, 6 planes out) or 9 (full matrix, 9 row-major planes out), :297-348.
 * const_c: C has coeffDim entries; else C is (coeffDim,Q,Q,Q,NE).
 * ------------------------------------------------------------------------------ */
void orc_diffusion_setup_3d(int Q, int coeffDim, int64_t ne, const double *W,
                            const double *J, const double *C, int const_c, double *D)
{
   const int64_t Q3 = (int64_t)Q*Q*Q;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      for (int64_t q = 0; q < Q3; q++)
      {
         const double *Je = J + 9*Q3*e + q;
         const double J11 = Je[0*Q3], J21 = Je[1*Q3], J31 = Je[2*Q3];
         const double J12 = Je[3*Q3], J22 = Je[4*Q3], J32 = Je[5*Q3];
         const double J13 = Je[6*Q3], J23 = Je[7*Q3], J33 = Je[8*Q3];
         const double detJ = J11*(J22*J33 - J32*J23) - J21*(J12*J33 - J32*J13) +
                             J31*(J12*J23 - J22*J13);
         const double w_detJ = W[q]/detJ;
         const double A11 = (J22*J33) - (J23*J32);
         const double A12 = (J32*J13) - (J12*J33);
         const double A13 = (J12*J23) - (J22*J13);
         const double A21 = (J31*J23) - (J21*J33);
         const double A22 = (J11*J33) - (J13*J31);
         const double A23 = (J21*J13) - (J11*J23);
         const double A31 = (J21*J32) - (J31*J22);
         const double A32 = (J31*J12) - (J11*J32);
         const double A33 = (J11*J22) - (J12*J21);
         const double *c = const_c ? C : C + coeffDim*(q + Q3*e);
         if (coeffDim == 6 || coeffDim == 9)
         {
            const int symmetric = coeffDim != 9;
            const double M11 = c[0], M12 = c[1], M13 = c[2];
            const double M21 = !symmetric ? c[3] : M12;
            const double M22 = !symmetric ? c[4] : c[3];
            const double M23 = !symmetric ? c[5] : c[4];
            const double M31 = !symmetric ? c[6] : M13;
            const double M32 = !symmetric ? c[7] : M23;
            const double M33 = !symmetric ? c[8] : c[5];
            const double R11 = M11*A11 + M12*A12 + M13*A13;
            const double R12 = M11*A21 + M12*A22 + M13*A23;
            const double R13 = M11*A31 + M12*A32 + M13*A33;
            const double R21 = M21*A11 + M22*A12 + M23*A13;
            const double R22 = M21*A21 + M22*A22 + M23*A23;
            const double R23 = M21*A31 + M22*A32 + M23*A33;
            const double R31 = M31*A11 + M32*A12 + M33*A13;
            const double R32 = M31*A21 + M32*A22 + M33*A23;
            const double R33 = M31*A31 + M32*A32 + M33*A33;
            double *Dm = D + (symmetric ? 6 : 9)*Q3*e + q;
            const double D12 = w_detJ*(A11*R12 + A12*R22 + A13*R32);
            const double D22 = w_detJ*(A21*R12 + A22*R22 + A23*R32);
            const double D23 = w_detJ*(A21*R13 + A22*R23 + A23*R33);
            const double D33 = w_detJ*(A31*R13 + A32*R23 + A33*R33);
            Dm[0*Q3] = w_detJ*(A11*R11 + A12*R21 + A13*R31);
            Dm[1*Q3] = D12;
            Dm[2*Q3] = w_detJ*(A11*R13 + A12*R23 + A13*R33);
            Dm[4*Q3] = symmetric ? D23 : D22;
            Dm[5*Q3] = symmetric ? D33 : D23;
            if (symmetric) { Dm[3*Q3] = D22; }
            else
            {
               Dm[3*Q3] = w_detJ*(A21*R11 + A22*R21 + A23*R31);
               Dm[6*Q3] = w_detJ*(A31*R11 + A32*R21 + A33*R31);
               Dm[7*Q3] = w_detJ*(A31*R12 + A32*R22 + A33*R32);
               Dm[8*Q3] = D33;
            }
            continue;
         }
         const double C1 = c[0];
         const double C2 = c[coeffDim == 3 ? 1 : 0];
         const double C3 = c[coeffDim == 3 ? 2 : 0];
         double *De = D + 6*Q3*e + q;
         De[0*Q3] = w_detJ*(C1*A11*A11 + C2*A12*A12 + C3*A13*A13);
         De[1*Q3] = w_detJ*(C1*A11*A21 + C2*A12*A22 + C3*A13*A23);
         De[2*Q3] = w_detJ*(C1*A11*A31 + C2*A12*A32 + C3*A13*A33);
         De[3*Q3] = w_detJ*(C1*A21*A21 + C2*A22*A22 + C3*A23*A23);
         De[4*Q3] = w_detJ*(C1*A21*A31 + C2*A22*A32 + C3*A23*A33);
         De[5*Q3] = w_detJ*(C1*A31*A31 + C2*A32*A32 + C3*A33*A33);
      }
   }
}

/* Mass setup (fem/integ/bilininteg_mass_pa.cpp:72-77): v(q,e) = W(q)*c*detJ(q,e). */
void orc_mass_setup_3d(int Q, int64_t ne, const double *W, const double *detJ,
                       const double *C, int const_c, double *D)
{
   const int64_t Q3 = (int64_t)Q*Q*Q;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      for (int64_t q = 0; q < Q3; q++)
      {
         const double coeff = const_c ? C[0] : C[q + Q3*e];
         D[q + Q3*e] = W[q]*coeff*detJ[q + Q3*e];
      }
   }
}

/* ---------------------------------------------------------------------------------
 * PADiffusionApply3D (fem/integ/bilininteg_diffusion_kernels.hpp:795-985), the
 * reference's host loop nest; symmetric D.  Y += A_e X.
 * ------------------------------------------------------------------------------ */
/* one element; D1D, Q1D are literals in the specialised calls below (the compiler then unrolls what the reference's
 * templated kernels unroll), run-time values in the generic one: the same additions in the same order either way */
static inline __attribute__((always_inline))
void diffusion_apply_elem(const int D1D, const int Q1D, const int symmetric, const double *B, const double *G,
                          const double *D, const double *X, double *Y, double (*grad)[3])
{
   const int64_t Q3 = (int64_t)Q1D*Q1D*Q1D;
   double gradXY[MAXQ][MAXQ][3], gradX[MAXQ][3];
   for (int64_t q = 0; q < Q3; q++) { grad[q][0] = grad[q][1] = grad[q][2] = 0.0; }
   for (int dz = 0; dz < D1D; ++dz)
   {
      for (int qy = 0; qy < Q1D; ++qy)
         for (int qx = 0; qx < Q1D; ++qx)
         { gradXY[qy][qx][0] = gradXY[qy][qx][1] = gradXY[qy][qx][2] = 0.0; }
      for (int dy = 0; dy < D1D; ++dy)
      {
         for (int qx = 0; qx < Q1D; ++qx) { gradX[qx][0] = gradX[qx][1] = 0.0; }
         for (int dx = 0; dx < D1D; ++dx)
         {
            const double s = X[dx + D1D*(dy + D1D*dz)];
            for (int qx = 0; qx < Q1D; ++qx)
            {
               gradX[qx][0] += s*B[qx + Q1D*dx];
               gradX[qx][1] += s*G[qx + Q1D*dx];
            }
         }
         for (int qy = 0; qy < Q1D; ++qy)
         {
            const double wy = B[qy + Q1D*dy], wDy = G[qy + Q1D*dy];
            for (int qx = 0; qx < Q1D; ++qx)
            {
               const double wx = gradX[qx][0], wDx = gradX[qx][1];
               gradXY[qy][qx][0] += wDx*wy;
               gradXY[qy][qx][1] += wx*wDy;
               gradXY[qy][qx][2] += wx*wy;
            }
         }
      }
      for (int qz = 0; qz < Q1D; ++qz)
      {
         const double wz = B[qz + Q1D*dz], wDz = G[qz + Q1D*dz];
         for (int qy = 0; qy < Q1D; ++qy)
            for (int qx = 0; qx < Q1D; ++qx)
            {
               double *g = grad[qx + Q1D*(qy + Q1D*qz)];
               g[0] += gradXY[qy][qx][0]*wz;
               g[1] += gradXY[qy][qx][1]*wz;
               g[2] += gradXY[qy][qx][2]*wDz;
            }
      }
   }
   for (int64_t q = 0; q < Q3; q++)
   {
      /* plane order, symmetric and full (:903-911) */
      const double O11 = D[q + 0*Q3], O12 = D[q + 1*Q3], O13 = D[q + 2*Q3];
      const double O21 = symmetric ? O12 : D[q + 3*Q3];
      const double O22 = symmetric ? D[q + 3*Q3] : D[q + 4*Q3];
      const double O23 = symmetric ? D[q + 4*Q3] : D[q + 5*Q3];
      const double O31 = symmetric ? O13 : D[q + 6*Q3];
      const double O32 = symmetric ? O23 : D[q + 7*Q3];
      const double O33 = symmetric ? D[q + 5*Q3] : D[q + 8*Q3];
      const double gX = grad[q][0], gY = grad[q][1], gZ = grad[q][2];
      grad[q][0] = (O11*gX) + (O12*gY) + (O13*gZ);
      grad[q][1] = (O21*gX) + (O22*gY) + (O23*gZ);
      grad[q][2] = (O31*gX) + (O32*gY) + (O33*gZ);
   }
   for (int qz = 0; qz < Q1D; ++qz)
   {
      double gXY[MAXQ][MAXQ][3];
      for (int dy = 0; dy < D1D; ++dy)
         for (int dx = 0; dx < D1D; ++dx)
         { gXY[dy][dx][0] = gXY[dy][dx][1] = gXY[dy][dx][2] = 0; }
      for (int qy = 0; qy < Q1D; ++qy)
      {
         double gXd[MAXQ][3];
         for (int dx = 0; dx < D1D; ++dx) { gXd[dx][0] = gXd[dx][1] = gXd[dx][2] = 0; }
         for (int qx = 0; qx < Q1D; ++qx)
         {
            const double *g = grad[qx + Q1D*(qy + Q1D*qz)];
            for (int dx = 0; dx < D1D; ++dx)
            {
               const double wx = B[qx + Q1D*dx], wDx = G[qx + Q1D*dx];
               gXd[dx][0] += g[0]*wDx;
               gXd[dx][1] += g[1]*wx;
               gXd[dx][2] += g[2]*wx;
            }
         }
         for (int dy = 0; dy < D1D; ++dy)
         {
            const double wy = B[qy + Q1D*dy], wDy = G[qy + Q1D*dy];
            for (int dx = 0; dx < D1D; ++dx)
            {
               gXY[dy][dx][0] += gXd[dx][0]*wy;
               gXY[dy][dx][1] += gXd[dx][1]*wDy;
               gXY[dy][dx][2] += gXd[dx][2]*wy;
            }
         }
      }
      for (int dz = 0; dz < D1D; ++dz)
      {
         const double wz = B[qz + Q1D*dz], wDz = G[qz + Q1D*dz];
         for (int dy = 0; dy < D1D; ++dy)
            for (int dx = 0; dx < D1D; ++dx)
            {
               Y[dx + D1D*(dy + D1D*dz)] +=
                  ((gXY[dy][dx][0]*wz) + (gXY[dy][dx][1]*wz) + (gXY[dy][dx][2]*wDz));
            }
      }
   }
}

#define ORC_REGISTERED_SHAPES(F) F(2,2) F(2,3) F(3,3) F(3,4) F(4,4) F(4,5) F(5,5) F(5,6) F(6,6) F(6,7) F(7,7) F(7,8) F(8,8) F(8,9) \
                                 F(9,9) F(9,10) F(4,6) F(5,8)

static void diffusion_apply_3d(int D1D, int Q1D, int64_t ne, int symmetric, const double *B,
                               const double *G, const double *Dd, const double *Xd,
                               double *Yd)
{
   const int64_t D3 = (int64_t)D1D*D1D*D1D, Q3 = (int64_t)Q1D*Q1D*Q1D;
   #pragma omp parallel
   {
      double (*grad)[3] = (double(*)[3])malloc(sizeof(double)*3*Q3);      /* per thread, not per element */
#define ORC_CASE(DD, QQ) \
      if (D1D == DD && Q1D == QQ && symmetric) \
      { \
         _Pragma("omp for schedule(static)") \
         for (int64_t e = 0; e < ne; e++) { diffusion_apply_elem(DD, QQ, 1, B, G, Dd + 6*Q3*e, Xd + D3*e, Yd + D3*e, grad); } \
      } else
      ORC_REGISTERED_SHAPES(ORC_CASE)
#undef ORC_CASE
      {
         #pragma omp for schedule(static)
         for (int64_t e = 0; e < ne; e++)
         {
            diffusion_apply_elem(D1D, Q1D, symmetric, B, G, Dd + (symmetric ? 6 : 9)*Q3*e, Xd + D3*e, Yd + D3*e, grad);
         }
      }
      free(grad);
   }
}

void orc_diffusion_apply_3d(int D1D, int Q1D, int64_t ne, const double *B,
                            const double *G, const double *Dd, const double *Xd,
                            double *Yd)
{
   diffusion_apply_3d(D1D, Q1D, ne, 1, B, G, Dd, Xd, Yd);
}

/* the same with the `symmetric` flag of the reference's signature: 0 = D has 9 row-major planes */
void orc_diffusion_apply_sym_3d(int D1D, int Q1D, int64_t ne, int symmetric, const double *B,
                                const double *G, const double *Dd, const double *Xd,
                                double *Yd)
{
   diffusion_apply_3d(D1D, Q1D, ne, symmetric, B, G, Dd, Xd, Yd);
}

/* ---------------------------------------------------------------------------------
 * PAMassApply3D_Element (fem/integ/bilininteg_mass_kernels.hpp:656-805), host loop nest.
 * Y += B^T diag(D) B X.
 * ------------------------------------------------------------------------------ */
static inline __attribute__((always_inline))
void mass_apply_elem(const int D1D, const int Q1D, const double *B, const double *D, const double *X, double *Y, double *sol_xyz)
{
   const int64_t Q3 = (int64_t)Q1D*Q1D*Q1D;
   for (int64_t q = 0; q < Q3; q++) { sol_xyz[q] = 0.0; }
   double sol_xy[MAXQ][MAXQ], sol_x[MAXQ];
   for (int dz = 0; dz < D1D; ++dz)
   {
      for (int qy = 0; qy < Q1D; ++qy)
         for (int qx = 0; qx < Q1D; ++qx) { sol_xy[qy][qx] = 0; }
      for (int dy = 0; dy < D1D; ++dy)
      {
         for (int qx = 0; qx < Q1D; ++qx) { sol_x[qx] = 0; }
         for (int dx = 0; dx < D1D; ++dx)
         {
            const double s = X[dx + D1D*(dy + D1D*dz)];
            for (int qx = 0; qx < Q1D; ++qx) { sol_x[qx] += B[qx + Q1D*dx]*s; }
         }
         for (int qy = 0; qy < Q1D; ++qy)
         {
            const double wy = B[qy + Q1D*dy];
            for (int qx = 0; qx < Q1D; ++qx) { sol_xy[qy][qx] += wy*sol_x[qx]; }
         }
      }
      for (int qz = 0; qz < Q1D; ++qz)
      {
         const double wz = B[qz + Q1D*dz];
         for (int qy = 0; qy < Q1D; ++qy)
            for (int qx = 0; qx < Q1D; ++qx)
            { sol_xyz[qx + Q1D*(qy + Q1D*qz)] += wz*sol_xy[qy][qx]; }
      }
   }
   for (int64_t q = 0; q < Q3; q++) { sol_xyz[q] *= D[q]; }
   for (int qz = 0; qz < Q1D; ++qz)
   {
      double s_xy[MAXQ][MAXQ];
      for (int dy = 0; dy < D1D; ++dy)
         for (int dx = 0; dx < D1D; ++dx) { s_xy[dy][dx] = 0; }
      for (int qy = 0; qy < Q1D; ++qy)
      {
         double s_x[MAXQ];
         for (int dx = 0; dx < D1D; ++dx) { s_x[dx] = 0; }
         for (int qx = 0; qx < Q1D; ++qx)
         {
            const double s = sol_xyz[qx + Q1D*(qy + Q1D*qz)];
            for (int dx = 0; dx < D1D; ++dx) { s_x[dx] += B[qx + Q1D*dx]*s; }
         }
         for (int dy = 0; dy < D1D; ++dy)
         {
            const double wy = B[qy + Q1D*dy];
            for (int dx = 0; dx < D1D; ++dx) { s_xy[dy][dx] += wy*s_x[dx]; }
         }
      }
      for (int dz = 0; dz < D1D; ++dz)
      {
         const double wz = B[qz + Q1D*dz];
         for (int dy = 0; dy < D1D; ++dy)
            for (int dx = 0; dx < D1D; ++dx)
            { Y[dx + D1D*(dy + D1D*dz)] += wz*s_xy[dy][dx]; }
      }
   }
}

void orc_mass_apply_3d(int D1D, int Q1D, int64_t ne, const double *B, const double *Dd,
                       const double *Xd, double *Yd)
{
   const int64_t D3 = (int64_t)D1D*D1D*D1D, Q3 = (int64_t)Q1D*Q1D*Q1D;
   #pragma omp parallel
   {
      double *sol_xyz = (double*)malloc(sizeof(double)*Q3);               /* per thread, not per element */
#define ORC_CASE(DD, QQ) \
      if (D1D == DD && Q1D == QQ) \
      { \
         _Pragma("omp for schedule(static)") \
         for (int64_t e = 0; e < ne; e++) { mass_apply_elem(DD, QQ, B, Dd + Q3*e, Xd + D3*e, Yd + D3*e, sol_xyz); } \
      } else
      ORC_REGISTERED_SHAPES(ORC_CASE) ORC_CASE(2, 4) ORC_CASE(3, 6)
#undef ORC_CASE
      {
         #pragma omp for schedule(static)
         for (int64_t e = 0; e < ne; e++) { mass_apply_elem(D1D, Q1D, B, Dd + Q3*e, Xd + D3*e, Yd + D3*e, sol_xyz); }
      }
      free(sol_xyz);
   }
}

/* ---------------------------------------------------------------------------------
 * SmemPADiffusionDiagonal3D (fem/integ/bilininteg_diffusion_kernels.hpp:368-484);
 * symmetric index map ksym (:428-431).  Y += diag(A_e).
 * ------------------------------------------------------------------------------ */
static void diffusion_diag_3d(int D1D, int Q1D, int64_t ne, int symmetric, const double *B,
                              const double *G, const double *Dd, double *Yd)
{
   const int64_t D3 = (int64_t)D1D*D1D*D1D, Q3 = (int64_t)Q1D*Q1D*Q1D;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      const double *D = Dd + (symmetric ? 6 : 9)*Q3*e;
      double *Y = Yd + D3*e;
      double *QQD = (double*)malloc(sizeof(double)*Q1D*Q1D*D1D);
      double *QDD = (double*)malloc(sizeof(double)*Q1D*D1D*D1D);
      for (int i = 0; i < 3; ++i)
      {
         for (int j = 0; j < 3; ++j)
         {
            const int ksym = j >= i ? 3 - (3-i)*(2-i)/2 + j : 3 - (3-j)*(2-j)/2 + i;
            const int k = symmetric ? ksym : 3*i + j;
            for (int qx = 0; qx < Q1D; ++qx)
               for (int qy = 0; qy < Q1D; ++qy)
                  for (int dz = 0; dz < D1D; ++dz)
                  {
                     double s = 0.0;
                     for (int qz = 0; qz < Q1D; ++qz)
                     {
                        const int64_t q = qx + (qy + (int64_t)qz*Q1D)*Q1D;
                        const double O = D[q + k*Q3];
                        const double Bz = B[qz + Q1D*dz], Gz = G[qz + Q1D*dz];
                        const double L = i == 2 ? Gz : Bz, R = j == 2 ? Gz : Bz;
                        s += L*O*R;
                     }
                     QQD[(qx*Q1D + qy)*D1D + dz] = s;
                  }
            for (int qx = 0; qx < Q1D; ++qx)
               for (int dz = 0; dz < D1D; ++dz)
                  for (int dy = 0; dy < D1D; ++dy)
                  {
                     double s = 0.0;
                     for (int qy = 0; qy < Q1D; ++qy)
                     {
                        const double By = B[qy + Q1D*dy], Gy = G[qy + Q1D*dy];
                        const double L = i == 1 ? Gy : By, R = j == 1 ? Gy : By;
                        s += L*QQD[(qx*Q1D + qy)*D1D + dz]*R;
                     }
                     QDD[(qx*D1D + dy)*D1D + dz] = s;
                  }
            for (int dz = 0; dz < D1D; ++dz)
               for (int dy = 0; dy < D1D; ++dy)
                  for (int dx = 0; dx < D1D; ++dx)
                  {
                     for (int qx = 0; qx < Q1D; ++qx)
                     {
                        const double Bx = B[qx + Q1D*dx], Gx = G[qx + Q1D*dx];
                        const double L = i == 0 ? Gx : Bx, R = j == 0 ? Gx : Bx;
                        Y[dx + D1D*(dy + D1D*dz)] += L*QDD[(qx*D1D + dy)*D1D + dz]*R;
                     }
                  }
         }
      }
      free(QQD); free(QDD);
   }
}

void orc_diffusion_diag_3d(int D1D, int Q1D, int64_t ne, const double *B,
                           const double *G, const double *Dd, double *Yd)
{
   diffusion_diag_3d(D1D, Q1D, ne, 1, B, G, Dd, Yd);
}

void orc_diffusion_diag_sym_3d(int D1D, int Q1D, int64_t ne, int symmetric, const double *B,
                               const double *G, const double *Dd, double *Yd)
{
   diffusion_diag_3d(D1D, Q1D, ne, symmetric, B, G, Dd, Yd);
}

/* SmemPAMassAssembleDiagonal3D (fem/integ/bilininteg_mass_kernels.hpp:324-408). */
void orc_mass_diag_3d(int D1D, int Q1D, int64_t ne, const double *B, const double *Dd,
                      double *Yd)
{
   const int64_t D3 = (int64_t)D1D*D1D*D1D, Q3 = (int64_t)Q1D*Q1D*Q1D;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      const double *D = Dd + Q3*e;
      double *Y = Yd + D3*e;
      double *QQD = (double*)malloc(sizeof(double)*Q1D*Q1D*D1D);
      double *QDD = (double*)malloc(sizeof(double)*Q1D*D1D*D1D);
      for (int qx = 0; qx < Q1D; ++qx)
         for (int qy = 0; qy < Q1D; ++qy)
            for (int dz = 0; dz < D1D; ++dz)
            {
               double s = 0.0;
               for (int qz = 0; qz < Q1D; ++qz)
               { s += B[qz + Q1D*dz]*B[qz + Q1D*dz]*D[qx + Q1D*(qy + (int64_t)Q1D*qz)]; }
               QQD[(qx*Q1D + qy)*D1D + dz] = s;
            }
      for (int qx = 0; qx < Q1D; ++qx)
         for (int dz = 0; dz < D1D; ++dz)
            for (int dy = 0; dy < D1D; ++dy)
            {
               double s = 0.0;
               for (int qy = 0; qy < Q1D; ++qy)
               { s += B[qy + Q1D*dy]*B[qy + Q1D*dy]*QQD[(qx*Q1D + qy)*D1D + dz]; }
               QDD[(qx*D1D + dy)*D1D + dz] = s;
            }
      for (int dz = 0; dz < D1D; ++dz)
         for (int dy = 0; dy < D1D; ++dy)
            for (int dx = 0; dx < D1D; ++dx)
            {
               double t = 0.0;
               for (int qx = 0; qx < Q1D; ++qx)
               { t += B[qx + Q1D*dx]*B[qx + Q1D*dx]*QDD[(qx*D1D + dy)*D1D + dz]; }
               Y[dx + D1D*(dy + D1D*dz)] += t;
            }
      free(QQD); free(QDD);
   }
}

/* ---------------------------------------------------------------------------------
 * DomainLFIntegrator device path for f = const (fem/integ/lininteg_domain.cpp:22-59,
 * lininteg_domain_kernels.hpp): b_E = B^T (W * detJ * f), then E->L sum.
 * ------------------------------------------------------------------------------ */
void orc_domain_lf_3d(int D1D, int Q1D, int64_t ne, const double *B, const double *W,
                      const double *detJ, double f, double *bE)
{
   const int64_t D3 = (int64_t)D1D*D1D*D1D, Q3 = (int64_t)Q1D*Q1D*Q1D;
   #pragma omp parallel for schedule(static)
   for (int64_t e = 0; e < ne; e++)
   {
      double *Y = bE + D3*e;
      double *t1 = (double*)malloc(sizeof(double)*D1D*Q1D*Q1D);
      double *t2 = (double*)malloc(sizeof(double)*D1D*D1D*Q1D);
      for (int qz = 0; qz < Q1D; qz++)
         for (int qy = 0; qy < Q1D; qy++)
            for (int dx = 0; dx < D1D; dx++)
            {
               double s = 0;
               for (int qx = 0; qx < Q1D; qx++)
               {
                  const int64_t q = qx + Q1D*(qy + (int64_t)Q1D*qz);
                  s += B[qx + Q1D*dx]*(W[q]*detJ[q + Q3*e]*f);
               }
               t1[dx + D1D*(qy + Q1D*qz)] = s;
            }
      for (int qz = 0; qz < Q1D; qz++)
         for (int dy = 0; dy < D1D; dy++)
            for (int dx = 0; dx < D1D; dx++)
            {
               double s = 0;
               for (int qy = 0; qy < Q1D; qy++) { s += B[qy + Q1D*dy]*t1[dx + D1D*(qy + Q1D*qz)]; }
               t2[dx + D1D*(dy + D1D*qz)] = s;
            }
      for (int dz = 0; dz < D1D; dz++)
         for (int dy = 0; dy < D1D; dy++)
            for (int dx = 0; dx < D1D; dx++)
            {
               double s = 0;
               for (int qz = 0; qz < Q1D; qz++) { s += B[qz + Q1D*dz]*t2[dx + D1D*(dy + D1D*qz)]; }
               Y[dx + D1D*(dy + D1D*dz)] = s;
            }
      free(t1); free(t2);
   }
}

/* ---------------------------------------------------------------------------------
 * Vector H1 spaces (vdim components of one scalar space; CEED BP2/BP4/BP6).
 * ------------------------------------------------------------------------------ */
/* ElementRestriction::TAddMultTranspose<ADD> / AbsMultTranspose for vdim > 1
 * (fem/restriction.cpp:152-180, 196-221): the scalar rows, per component, reading
 * xE(idx % nd, c, idx / nd). */
void orc_restriction_mult_transpose_vdim(int64_t ndofs, int nd, int vd, int byvdim,
                                         const int *offsets, const int *indices,
                                         const double *xE, double *y, int add, int absv)
{
   #pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < ndofs; i++)
   {
      const int offset = offsets[i], next_offset = offsets[i + 1];
      for (int c = 0; c < vd; ++c)
      {
         double dof_value = 0;
         for (int j = offset; j < next_offset; ++j)
         {
            const int64_t idx_j = (indices[j] >= 0) ? indices[j] : -1 - indices[j];
            const double v = xE[(idx_j % nd) + nd*(c + (int64_t)vd*(idx_j/nd))];
            dof_value += (indices[j] >= 0 || absv) ? v : -v;
         }
         const int64_t k = byvdim ? c + (int64_t)vd*i : i + c*ndofs;
         if (add) { y[k] += dof_value; }
         else { y[k] = dof_value; }
      }
   }
}

/* VectorDiffusionIntegrator::AddMultPA, 3D, scalar or vector coefficient
 * (fem/integ/bilininteg_vecdiffusion_pa.hpp:108-170: for every component i the scalar
 * diffusion kernel body with block k = i of pa_data (Q,Q,Q,9,vdim,NE) -- dim*dim planes
 * allocated, the symmetric six filled, bilininteg_vecdiffusion_pa.cpp:133-134,228-290);
 * E-vectors (D,D,D,vdim,NE). */
void orc_vector_diffusion_apply_3d(int D1D, int Q1D, int64_t ne, int vdim, const double *B,
                                   const double *G, const double *Dd, const double *xE, double *yE)
{
   const int64_t nd = (int64_t)D1D*D1D*D1D, nq6 = 9*(int64_t)Q1D*Q1D*Q1D;
   for (int64_t e = 0; e < ne; e++)
      for (int c = 0; c < vdim; c++)
      {
         const int64_t blk = e*vdim + c;
         orc_diffusion_apply_3d(D1D, Q1D, 1, B, G, Dd + blk*nq6, xE + blk*nd, yE + blk*nd);
      }
}

/* VectorMassIntegrator::AddMultPA, 3D (fem/integ/bilininteg_vecmass_pa.hpp; pa_data
 * (Q,Q,Q,coeff_vdim,NE), bilininteg_vecmass_pa.cpp:118-160): coeff_vdim = 1 (one scalar
 * field for all components) or vdim. */
void orc_vector_mass_apply_3d(int D1D, int Q1D, int64_t ne, int vdim, int coeff_vdim,
                              const double *B, const double *Dd, const double *xE, double *yE)
{
   const int64_t nd = (int64_t)D1D*D1D*D1D, nq = (int64_t)Q1D*Q1D*Q1D;
   for (int64_t e = 0; e < ne; e++)
      for (int c = 0; c < vdim; c++)
      {
         const int64_t blk = e*vdim + c;
         const double *De = Dd + (coeff_vdim == 1 ? e : blk)*nq;
         orc_mass_apply_3d(D1D, Q1D, 1, B, De, xE + blk*nd, yE + blk*nd);
      }
}

/* ---------------------------------------------------------------------------------
 * The L->L operator and the Krylov path.
 * ------------------------------------------------------------------------------ */
typedef struct
{
   int kind;            /* 0 mass, 1 diffusion */
   int D1D, Q1D;
   int64_t ne, ndofs;
   const double *B, *G, *pa_data;
   const int *gather_map, *offsets, *indices;
   int ness; const int *ess;      /* essential true dofs (constraint_list) */
   double *xE, *yE, *w, *z;       /* work */
   /* vector spaces: vdim = 0 scalar; else vdim components, ordering byNODES (0) / byVDIM (1),
    * pa_data as the reference's vector integrators store it, coeff_vdim = 1 or vdim */
   int vdim, byvdim, coeff_vdim;
} orc_op;

static int64_t orc_size(const orc_op *op) { return op->ndofs*(op->vdim > 0 ? op->vdim : 1); }

/* PABilinearFormExtension::MultInternal, H1 branch (fem/bilinearform_ext.cpp:530-559). */
void orc_pa_mult(const orc_op *op, const double *x, double *y)
{
   const int nd = op->D1D*op->D1D*op->D1D;
   const int64_t nE = op->ne*nd;
   if (op->vdim > 0)
   {
      const int vd = op->vdim;
      orc_restriction_mult(nE, nd, vd, op->byvdim, op->ndofs, op->gather_map, x, op->xE);
      memset(op->yE, 0, sizeof(double)*nE*vd);
      if (op->kind == 0) { orc_vector_mass_apply_3d(op->D1D, op->Q1D, op->ne, vd, op->coeff_vdim, op->B, op->pa_data, op->xE, op->yE); }
      else { orc_vector_diffusion_apply_3d(op->D1D, op->Q1D, op->ne, vd, op->B, op->G, op->pa_data, op->xE, op->yE); }
      orc_restriction_mult_transpose_vdim(op->ndofs, nd, vd, op->byvdim, op->offsets, op->indices, op->yE, y, 0, 0);
      return;
   }
   orc_restriction_mult(nE, nd, 1, 0, op->ndofs, op->gather_map, x, op->xE);
   memset(op->yE, 0, sizeof(double)*nE);
   if (op->kind == 0) { orc_mass_apply_3d(op->D1D, op->Q1D, op->ne, op->B, op->pa_data, op->xE, op->yE); }
   else { diffusion_apply_3d(op->D1D, op->Q1D, op->ne, op->kind != 2, op->B, op->G, op->pa_data, op->xE, op->yE); }   /* kind 2: full D */
   orc_restriction_mult_transpose(op->ndofs, op->offsets, op->indices, op->yE, y, 0, 0);
}

/* PABilinearFormExtension::AssembleDiagonal (fem/bilinearform_ext.cpp:370-454):
 * E-diagonal then AbsMultTranspose (:413-418). */
void orc_pa_diag(const orc_op *op, double *diag)
{
   const int nd = op->D1D*op->D1D*op->D1D;
   memset(op->yE, 0, sizeof(double)*op->ne*nd);
   if (op->kind == 0) { orc_mass_diag_3d(op->D1D, op->Q1D, op->ne, op->B, op->pa_data, op->yE); }
   else { diffusion_diag_3d(op->D1D, op->Q1D, op->ne, op->kind != 2, op->B, op->G, op->pa_data, op->yE); }
   orc_restriction_mult_transpose(op->ndofs, op->offsets, op->indices, op->yE, diag, 0, 1);
}

/* ConstrainedOperator::ConstrainedMult, DIAG_ONE (linalg/operator.cpp:586-646). */
void orc_constrained_mult(const orc_op *op, const double *x, double *y)
{
   const int64_t n = orc_size(op);
   if (op->ness == 0) { orc_pa_mult(op, x, y); return; }
   memcpy(op->z, x, sizeof(double)*n);
   for (int i = 0; i < op->ness; i++) { op->z[op->ess[i]] = 0.0; }
   orc_pa_mult(op, op->z, y);
   for (int i = 0; i < op->ness; i++) { y[op->ess[i]] = x[op->ess[i]]; }
}

/* ConstrainedOperator::EliminateRHS (linalg/operator.cpp:559-584):
 * w = 0; w[ess] = x[ess]; b -= A w; b[ess] = x[ess]. */
void orc_eliminate_rhs(const orc_op *op, const double *x, double *b)
{
   const int64_t n = orc_size(op);
   memset(op->w, 0, sizeof(double)*n);
   for (int i = 0; i < op->ness; i++) { op->w[op->ess[i]] = x[op->ess[i]]; }
   orc_pa_mult(op, op->w, op->z);
   for (int64_t i = 0; i < n; i++) { b[i] -= op->z[i]; }
   for (int i = 0; i < op->ness; i++) { b[op->ess[i]] = x[op->ess[i]]; }
}

/* OperatorJacobiSmoother::Setup (linalg/solvers.cpp:401-425), damping = 1. */
void orc_jacobi_setup(int64_t n, const double *diag, int ness, const int *ess, double *dinv)
{
   for (int64_t i = 0; i < n; i++) { dinv[i] = 1.0/diag[i]; }
   for (int i = 0; i < ness; i++) { dinv[ess[i]] = 1.0; }
}

/* TIMING LEG ONLY (bench.py cpu_baseline): the vector updates and dot products of orc_pcg threaded as MFEM's `-d cpu` path
 * with one MPI rank per core would run them -- every core updates its own contiguous block and sums its own part of a
 * dot product sequentially (the rank-local Vector::operator*), the partial sums are added in block (= rank) order (the
 * all-reduce).  The sum order differs from the single-rank sequential reference: never switched on by a parity test. */
static int orc_threaded_vectors = 0;
void orc_set_threaded_vectors(int on) { orc_threaded_vectors = on; }

/* Vector::operator* on host: strictly sequential (general/reducers.hpp:589-592). */
static double orc_dot(int64_t n, const double *a, const double *b)
{
   if (orc_threaded_vectors)
   {
      enum { MAXT = 1024 };
      static double part[MAXT];
      int T = omp_get_max_threads();
      if (T > MAXT) { T = MAXT; }
      #pragma omp parallel num_threads(T)
      {
         const int t = omp_get_thread_num();
         const int64_t i0 = n*t/T, i1 = n*(t + 1)/T;
         double s = 0.0;
         for (int64_t i = i0; i < i1; i++) { s += a[i]*b[i]; }
         part[t] = s;
      }
      double s = 0.0;
      for (int t = 0; t < T; t++) { s += part[t]; }
      return s;
   }
   double s = 0.0;
   for (int64_t i = 0; i < n; i++) { s += a[i]*b[i]; }
   return s;
}

/* CGSolver::Mult with Jacobi preconditioner, iterative_mode = false
 * (linalg/solvers.cpp:869-1050; Jacobi Mult :427-453).  dinv == NULL -> unpreconditioned.
 * hist[k] = (B r, r) at iteration k (hist must hold max_iter+1 entries).
 * Returns the final iteration count; *converged is set as the reference sets it. */
#define ORC_VLOOP _Pragma("omp parallel for schedule(static) if(orc_threaded_vectors)")
int orc_pcg(const orc_op *op, const double *dinv, const double *b, double *x,
            double rel_tol, double abs_tol, int max_iter, double *hist, int *converged)
{
   const int64_t n = orc_size(op);
   double *r = (double*)malloc(sizeof(double)*n), *d = (double*)malloc(sizeof(double)*n),
          *z = (double*)malloc(sizeof(double)*n);
   double r0, den, nom, betanom = 0, alpha, beta;
   int i, final_iter = 0;
   memcpy(r, b, sizeof(double)*n);
   memset(x, 0, sizeof(double)*n);
   if (dinv) { ORC_VLOOP for (int64_t k = 0; k < n; k++) { z[k] = 0.0 + dinv[k]*r[k]; d[k] = z[k]; } }
   else { memcpy(d, r, sizeof(double)*n); }
   nom = orc_dot(n, d, r);
   if (hist) { hist[0] = nom; }
   *converged = 0;
   if (nom < 0.0) { goto done; }
   r0 = fmax(nom*rel_tol*rel_tol, abs_tol*abs_tol);
   if (nom <= r0) { *converged = 1; goto done; }
   orc_constrained_mult(op, d, z);
   den = orc_dot(n, z, d);
   if (den == 0.0) { goto done; }
   final_iter = max_iter;
   for (i = 1; 1; )
   {
      alpha = nom/den;
      ORC_VLOOP for (int64_t k = 0; k < n; k++) { x[k] = x[k] + alpha*d[k]; }
      ORC_VLOOP for (int64_t k = 0; k < n; k++) { r[k] = r[k] + (-alpha)*z[k]; }
      if (dinv)
      {
         ORC_VLOOP for (int64_t k = 0; k < n; k++) { z[k] = 0.0 + dinv[k]*r[k]; }
         betanom = orc_dot(n, r, z);
      }
      else { betanom = orc_dot(n, r, r); }
      if (hist) { hist[i] = betanom; }
      if (betanom < 0.0) { final_iter = i; break; }
      if (betanom <= r0) { *converged = 1; final_iter = i; break; }
      if (++i > max_iter) { break; }
      beta = betanom/nom;
      if (dinv) { ORC_VLOOP for (int64_t k = 0; k < n; k++) { d[k] = z[k] + beta*d[k]; } }
      else { ORC_VLOOP for (int64_t k = 0; k < n; k++) { d[k] = r[k] + beta*d[k]; } }
      orc_constrained_mult(op, d, z);
      den = orc_dot(n, d, z);
      if (den == 0.0) { final_iter = i; break; }
      nom = betanom;
   }
done:
   free(r); free(d); free(z);
   return final_iter;
}

#undef ORC_VLOOP

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
   if (n > 0) { omp_set_num_threads(n); }
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
   return omp_get_max_threads();
#else
   return 1;
#endif
}

// vcoef_check -- Vector{Mass,Diffusion}Integrator(VectorCoefficient&) of the C++ layer against what it must equal:
// component c of the vector operator is the SCALAR operator with coefficient c_k (the reference applies the scalar
// kernel per component with coeff_vdim = vdim, fem/integ/bilininteg_vecdiffusion_pa.hpp:108-170,
// bilininteg_vecmass_pa.cpp:118-160).  For both orderings, a constant and a function coefficient: operator action,
// diagonal, and the Jacobi-PCG solve of the constrained system (component c = the scalar solve with c_k).
// Prints one line per check with the largest difference (0 = the same bits) and "OK" / "FAILED" last.
#include "mfem.hpp"
#include <cstdio>

using namespace mfem;

static double coef_k(int k, const Vector &x) { return (k + 1.0)*0.5 + x(0)*x(0) + (k == 1 ? x(1)*x(2) : 0.25*x(k)); }

int main(int argc, char *argv[])
{
   int p = 2, N = 3; bool fn = false;
   OptionsParser args(argc, argv);
   args.AddOption(&p, "-p", "--order", "order");
   args.AddOption(&N, "-n", "--elements", "elements per direction");
   args.AddOption(&fn, "-fn", "--function", "-cc", "--constant", "function instead of constant component coefficients");
   args.Parse();
   if (!args.Good()) { args.PrintUsage(std::cout); return 1; }
   Device device("hip");
   Mesh mesh = Mesh::MakeCartesian3D(N, N, N, Element::HEXAHEDRON, 1.0, 1.2, 0.9, true);
   mesh.SetCurvature(2);
   mesh.Transform([](const Vector &x, Vector &y) { y = x; y(0) += 0.05*x(1)*x(2); y(2) += 0.04*x(0)*x(0); });
   H1_FECollection fec(p, 3);
   FiniteElementSpace sfes(&mesh, &fec);
   const int ns = sfes.GetNDofs();
   double worst = 0.0;
   for (int mass = 0; mass < 2; mass++)
   {
      // the three scalar operators
      Vector cvals(3); cvals(0) = 1.5; cvals(1) = 0.5; cvals(2) = 2.0;
      VectorConstantCoefficient vconst(cvals);
      VectorFunctionCoefficient vfunc(3, [](const Vector &x, Vector &v) { v.SetSize(3); for (int k = 0; k < 3; k++) { v(k) = coef_k(k, x); } });
      VectorCoefficient &VQ = fn ? (VectorCoefficient&)vfunc : (VectorCoefficient&)vconst;
      Array<int> ess_bdr(mesh.bdr_attributes.Max()), sess;
      ess_bdr = 1;
      sfes.GetEssentialTrueDofs(ess_bdr, sess);
      std::vector<Vector> ys(3), ds(3), Xs(3);
      std::vector<int> its(3);
      Vector xs[3];
      for (int k = 0; k < 3; k++)
      {
         ConstantCoefficient ck(cvals(k));
         FunctionCoefficient fk([k](const Vector &x) { return coef_k(k, x); });
         Coefficient &Q = fn ? (Coefficient&)fk : (Coefficient&)ck;
         BilinearForm a(&sfes);
         a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
         if (mass) { a.AddDomainIntegrator(new MassIntegrator(Q)); } else { a.AddDomainIntegrator(new DiffusionIntegrator(Q)); }
         a.Assemble();
         xs[k].SetSize(ns); xs[k].Randomize(7 + k);
         ys[k].SetSize(ns); a.Mult(xs[k], ys[k]);
         ds[k].SetSize(ns); a.AssembleDiagonal(ds[k]);
         LinearForm b(&sfes); ConstantCoefficient one(1.0);
         b.AddDomainIntegrator(new DomainLFIntegrator(one)); b.Assemble();
         GridFunction u(&sfes); u = 0.0;
         OperatorPtr A; Vector B, X;
         a.FormLinearSystem(sess, u, b, A, X, B);
         OperatorJacobiSmoother M(a, sess);
         CGSolver cg; cg.SetRelTol(1e-8); cg.SetMaxIter(500); cg.SetPrintLevel(-1);
         cg.SetOperator(*A); cg.SetPreconditioner(M); cg.Mult(B, X);
         its[k] = cg.GetNumIterations(); Xs[k] = X;
      }
      for (int byvdim = 0; byvdim < 2; byvdim++)
      {
         FiniteElementSpace vfes(&mesh, &fec, 3, byvdim ? Ordering::byVDIM : Ordering::byNODES);
         auto at = [&](int i, int k) { return byvdim ? 3*i + k : i + k*ns; };
         BilinearForm a(&vfes);
         a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
         if (mass) { a.AddDomainIntegrator(new VectorMassIntegrator(VQ)); } else { a.AddDomainIntegrator(new VectorDiffusionIntegrator(VQ)); }
         a.Assemble();
         Vector x(3*ns), y(3*ns), d(3*ns);
         for (int k = 0; k < 3; k++) { for (int i = 0; i < ns; i++) { x(at(i, k)) = xs[k](i); } }
         a.Mult(x, y);
         a.AssembleDiagonal(d);
         double ey = 0.0, ed = 0.0;
         for (int k = 0; k < 3; k++) { for (int i = 0; i < ns; i++) { ey = std::max(ey, std::fabs(y(at(i, k)) - ys[k](i))); ed = std::max(ed, std::fabs(d(at(i, k)) - ds[k](i))); } }
         // the solve: essential dofs of all components on the boundary, f = 1 per component
         Array<int> vess;
         vfes.GetEssentialTrueDofs(ess_bdr, vess);
         LinearForm b(&vfes);
         Vector ones(3); ones = 1.0;
         VectorConstantCoefficient f(ones);
         b.AddDomainIntegrator(new VectorDomainLFIntegrator(f)); b.Assemble();
         GridFunction u(&vfes); u = 0.0;
         OperatorPtr A; Vector B, X;
         a.FormLinearSystem(vess, u, b, A, X, B);
         OperatorJacobiSmoother M(a, vess);
         CGSolver cg; cg.SetRelTol(1e-8); cg.SetMaxIter(500); cg.SetPrintLevel(-1);
         cg.SetOperator(*A); cg.SetPreconditioner(M); cg.Mult(B, X);
         double ex = 0.0, xn = 0.0;
         for (int k = 0; k < 3; k++) { for (int i = 0; i < ns; i++) { ex = std::max(ex, std::fabs(X(at(i, k)) - Xs[k](i))); xn = std::max(xn, std::fabs(Xs[k](i))); } }
         printf("%s %s %s: |y - y_scalar| = %.3e  |diag - diag_scalar| = %.3e  |X - X_scalar|/|X| = %.3e  (iterations %d; scalar solves %d %d %d)\n",
                mass ? "mass" : "diffusion", byvdim ? "byVDIM" : "byNODES", fn ? "function" : "constant", ey, ed, ex/xn,
                cg.GetNumIterations(), its[0], its[1], its[2]);
         worst = std::max(worst, std::max(ey, ed));
         if (ex/xn > 1e-6) { worst = 1.0; }       // the coupled solve stops on the sum of the three residuals: close, not equal
      }
   }
   printf(worst == 0.0 ? "OK\n" : "FAILED\n");
   return worst == 0.0 ? 0 : 1;
}

// TEST INFRASTRUCTURE (known-answer generator) -- not part of the product.
//
// BASELINE config 4 on the *reference* (serial MFEM, CPU): BP3 / BP5 on a curved high-order hex mesh,
// uniformly refined, with the variable coefficient 1 + x^2 + y z at the quadrature points (the
// `const_c = false` branch of PADiffusionSetup3D) -- the bake-off Kernel / Problem protocol of
// tests/benchmarks/bench_ceed.cpp on a mesh file: |A x| for x.Randomize(1), Jacobi-PCG iteration
// count and |X| at rel tol 1e-8.  Prints JSON lines = tests/golden/baseline_anchors_curved.json.
// data/escher-p3.mesh is tetrahedral (outside the hexahedral path); the curved hex meshes of the
// reference's data directory stand in for it.  Linked like oracle/ref_dump.cpp:
//   g++ -O2 -std=c++17 -I/tmp/mfem-build -I/root/reference oracle/ref_probe_curved.cpp -o /tmp/ref_probe_curved \
//       /tmp/mfem-build/libmfem.a -lrt
#include "mfem.hpp"
#include <cstdio>
#include <string>
using namespace mfem;

static double coef_fn(const Vector &x) { return 1.0 + x(0)*x(0) + x(1)*x(2); }

static void run(const char *name, int levels, int bp, int p)
{
   const bool GLL = bp >= 5;
   Mesh mesh((std::string("/root/reference/data/") + name).c_str(), 1, 1);
   for (int l = 0; l < levels; l++) { mesh.UniformRefinement(); }
   H1_FECollection fec(p, 3, BasisType::GaussLobatto);
   FiniteElementSpace fes(&mesh, &fec);
   IntegrationRules irs(0, GLL ? Quadrature1D::GaussLobatto : Quadrature1D::GaussLegendre);
   const IntegrationRule *ir = &irs.Get(Geometry::CUBE, 2*p + (GLL ? -1 : 3));
   FunctionCoefficient c(coef_fn);
   ConstantCoefficient one(1.0);
   GridFunction x(&fes), y(&fes);
   BilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   a.AddDomainIntegrator(new DiffusionIntegrator(c, ir));
   a.Assemble();
   x.Randomize(1);
   a.Mult(x, y);
   const double nAx = y.Norml2();
   Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess;
   ess_bdr = 1; fes.GetEssentialTrueDofs(ess_bdr, ess);
   LinearForm b(&fes); b.AddDomainIntegrator(new DomainLFIntegrator(one)); b.Assemble();
   x = 0.0;
   OperatorPtr A; Vector B, X;
   a.FormLinearSystem(ess, x, b, A, X, B);
   OperatorJacobiSmoother M(a, ess);
   CGSolver cg;
   cg.SetRelTol(1e-8); cg.SetAbsTol(0.0); cg.SetMaxIter(4000); cg.SetPrintLevel(-1);
   cg.SetOperator(*A); cg.SetPreconditioner(M); cg.iterative_mode = false;
   cg.Mult(B, X);
   printf("{\"mesh\": \"%s\", \"levels\": %d, \"problem\": \"bp%d\", \"p\": %d, \"NE\": %d, \"dofs\": %d, \"pcg_its\": %d, "
          "\"converged\": %d, \"norm_Ax\": %.16e, \"norm_X\": %.16e, \"norm_B\": %.16e}\n", name, levels, bp, p, mesh.GetNE(),
          fes.GetTrueVSize(), cg.GetNumIterations(), (int)cg.GetConverged(), nAx, X.Norml2(), B.Norml2());
   fflush(stdout);
}

int main()
{
   for (int bp = 3; bp <= 5; bp += 2)
   {
      for (int p = 2; p <= 4; p++) { run("toroid-hex.mesh", 2, bp, p); }
      run("fichera-q3.mesh", 1, bp, 3);
      run("toroid-hex.mesh", 3, bp, 6);
   }
   return 0;
}

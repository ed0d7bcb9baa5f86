// TEST INFRASTRUCTURE (known-answer generator) -- not part of the product.
//
// The bake-off Kernel / Problem set-up of tests/benchmarks/bench_ceed.cpp:24-200 for the vector
// problems BP2/BP4/BP6 on the *reference* (serial MFEM, CPU): |A x| for x.Randomize(1), and the
// Jacobi-PCG iteration count and |X| at rel tol 1e-12 (the protocol of BASELINE.md 2.2), for
// p = 1..8 on MakeCartesian3D(N,N,N), N = 48/p/... (see main).  Prints JSON lines; the result is
// tests/golden/baseline_anchors_vec.json.  Linked like oracle/ref_dump.cpp against the serial
// libmfem.a the survey stage left in /tmp/mfem-build:
//   g++ -O2 -std=c++17 -I/tmp/mfem-build -I/root/reference oracle/ref_probe_vec.cpp -o /tmp/ref_probe_vec \
//       /tmp/mfem-build/libmfem.a -lrt
#include "mfem.hpp"
#include <cstdio>
using namespace mfem;

static void run(int bp, int p, int N)
{
   const bool GLL = bp >= 5, mass = bp <= 2;
   const int VDIM = 3;
   Mesh mesh = Mesh::MakeCartesian3D(N, N, N, Element::HEXAHEDRON);
   H1_FECollection fec(p, 3, BasisType::GaussLobatto);
   FiniteElementSpace fes(&mesh, &fec, VDIM, Ordering::byVDIM);
   IntegrationRules irs(0, GLL ? Quadrature1D::GaussLobatto : Quadrature1D::GaussLegendre);
   const IntegrationRule *ir = &irs.Get(Geometry::CUBE, 2*p + (GLL ? -1 : 3));
   ConstantCoefficient one(1.0);
   GridFunction x(&fes), y(&fes);
   BilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   if (mass) { a.AddDomainIntegrator(new VectorMassIntegrator(one, ir)); }
   else { a.AddDomainIntegrator(new VectorDiffusionIntegrator(one, ir)); }
   a.Assemble();
   x.Randomize(1);
   a.Mult(x, y);
   const double nAx = y.Norml2();
   Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess;
   ess_bdr = 1; fes.GetEssentialTrueDofs(ess_bdr, ess);
   Vector uvec(3); uvec = 1.0; uvec /= uvec.Norml2();
   VectorConstantCoefficient unit_vec(uvec);
   LinearForm b(&fes); b.AddDomainIntegrator(new VectorDomainLFIntegrator(unit_vec)); b.Assemble();
   x = 0.0;
   OperatorPtr A; Vector B, X;
   a.FormLinearSystem(ess, x, b, A, X, B);
   OperatorJacobiSmoother M(a, ess);
   CGSolver cg;
   cg.SetRelTol(1e-12); cg.SetAbsTol(0.0); cg.SetMaxIter(2000); cg.SetPrintLevel(-1);
   cg.SetOperator(*A); cg.SetPreconditioner(M); cg.iterative_mode = false;
   cg.Mult(B, X);
   printf("{\"problem\": \"bp%d\", \"p\": %d, \"N\": %d, \"D1D\": %d, \"Q1D\": %d, \"NE\": %d, \"dofs\": %d, \"pcg_its\": %d, "
          "\"norm_Ax\": %.16e, \"norm_X\": %.16e, \"norm_B\": %.16e}\n", bp, p, N, p + 1, (int)std::lround(std::cbrt(ir->GetNPoints())), mesh.GetNE(),
          fes.GetTrueVSize(), cg.GetNumIterations(), nAx, X.Norml2(), B.Norml2());
   fflush(stdout);
}

int main()
{
   for (int bp = 2; bp <= 6; bp += 2) for (int p = 1; p <= 8; p++) { run(bp, p, 24/p > 0 ? 24/p : 1); }
   return 0;
}

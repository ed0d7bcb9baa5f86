// TEST INFRASTRUCTURE (fixture generator) -- not part of the product.
//
// Vector H1 spaces (CEED BP2/BP4/BP6: VectorMassIntegrator / VectorDiffusionIntegrator with
// AssemblyLevel::PARTIAL on a vdim = 3 space): runs the *reference* (serial MFEM, CPU) on small
// cases and dumps the arrays of the path in the GLD1 container of oracle/ref_dump.cpp;
// oracle/pack_fixtures.py converts them to tests/golden/vec_*.npz.  Linked, like ref_dump.cpp,
// against the serial libmfem.a the survey stage left in /tmp/mfem-build ("Oracle"):
//
//   g++ -O2 -std=c++17 -fno-access-control -I/tmp/mfem-build -I/root/reference \
//       oracle/ref_dump_vec.cpp -o /tmp/ref_dump_vec /tmp/mfem-build/libmfem.a -lrt
#include "mfem.hpp"
#include <cstdio>
#include <cstdint>
#include <sstream>
#include <string>
#include <vector>

using namespace mfem;
using namespace std;

struct Dump
{
   FILE *f;
   Dump(const string &path) { f = fopen(path.c_str(), "wb"); fwrite("GLD1", 1, 4, f); }
   ~Dump() { fclose(f); }
   void rec(const string &name, char ty, int64_t n, const void *p, int esz)
   {
      int32_t nl = (int32_t)name.size();
      fwrite(&nl, 4, 1, f); fwrite(name.data(), 1, nl, f);
      fwrite(&ty, 1, 1, f); fwrite(&n, 8, 1, f);
      if (n) { fwrite(p, esz, n, f); }
   }
   void d(const string &n, const Vector &v) { v.HostRead(); rec(n, 'd', v.Size(), v.GetData(), 8); }
   void d(const string &n, const Array<real_t> &v) { v.HostRead(); rec(n, 'd', v.Size(), v.GetData(), 8); }
   void d(const string &n, const vector<double> &v) { rec(n, 'd', v.size(), v.data(), 8); }
   void i(const string &n, const Array<int> &v) { v.HostRead(); rec(n, 'i', v.Size(), v.GetData(), 4); }
   void s(const string &n, int v) { rec(n, 'i', 1, &v, 4); }
};

static double smix(uint64_t seed, uint64_t i)
{
   uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (i + 1) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   z = z ^ (z >> 31);
   return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// kind: 0 = vector mass, 1 = vector diffusion.  rule: 0 = Gauss-Legendre 2p+3, 1 = Gauss-Lobatto
// 2p-1 (tests/benchmarks/bench_ceed.cpp:45,50), 2 = integrator default.
static void run_case(const string &out, Mesh &mesh, int p, int kind, int rule, int ordering)
{
   const int dim = 3, vdim = 3;
   H1_FECollection fec(p, dim, BasisType::GaussLobatto);
   FiniteElementSpace fes(&mesh, &fec, vdim, ordering);
   const FiniteElement &el = *fes.GetTypicalFE();
   IntegrationRules irs(0, rule == 1 ? Quadrature1D::GaussLobatto : Quadrature1D::GaussLegendre);
   const IntegrationRule *ir = nullptr;
   if (rule == 0) { ir = &irs.Get(Geometry::CUBE, 2*p + 3); }
   if (rule == 1) { ir = &irs.Get(Geometry::CUBE, 2*p - 1); }
   ConstantCoefficient one(1.0);
   BilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   BilinearFormIntegrator *bfi = nullptr;
   if (kind == 0) { bfi = ir ? new VectorMassIntegrator(one, ir) : new VectorMassIntegrator(one); }
   else { bfi = ir ? new VectorDiffusionIntegrator(one, ir) : new VectorDiffusionIntegrator(one); }
   a.AddDomainIntegrator(bfi);
   a.Assemble();
   if (!ir)
   {
      ir = (kind == 0) ? &MassIntegrator::GetRule(el, el, *mesh.GetTypicalElementTransformation())
           : &DiffusionIntegrator::GetRule(el, el);
   }
   Dump D(out);
   const DofToQuad &maps = el.GetDofToQuad(*ir, DofToQuad::TENSOR);
   const int D1D = maps.ndof, Q1D = maps.nqpt, NE = fes.GetNE(), ND = el.GetDof();
   D.s("p", p); D.s("kind", kind); D.s("rule", rule); D.s("vdim", vdim); D.s("byvdim", ordering == Ordering::byVDIM);
   D.s("D1D", D1D); D.s("Q1D", Q1D); D.s("NE", NE); D.s("ndofs", fes.GetNDofs());
   D.d("B", maps.B); D.d("G", maps.G);
   const auto *R = dynamic_cast<const ElementRestriction*>(fes.GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC));
   MFEM_VERIFY(R, "");
   D.i("gather_map", R->GatherMap()); D.i("offsets", R->Offsets()); D.i("indices", R->Indices());
   if (kind == 0) { D.d("pa_data", static_cast<VectorMassIntegrator*>(bfi)->pa_data); D.s("coeff_vdim", static_cast<VectorMassIntegrator*>(bfi)->coeff_vdim); }
   else { D.d("pa_data", static_cast<VectorDiffusionIntegrator*>(bfi)->pa_data); D.s("coeff_vdim", static_cast<VectorDiffusionIntegrator*>(bfi)->coeff_vdim); }

   const int N = fes.GetVSize();
   Vector x(N), y(N), xE(NE*ND*vdim), yE(NE*ND*vdim), diag(N);
   for (int i = 0; i < N; i++) { x(i) = smix(1, i); }
   a.Mult(x, y);
   R->Mult(x, xE); yE = 0.0; bfi->AddMultPA(xE, yE);
   a.AssembleDiagonal(diag);
   D.d("x", x); D.d("y", y); D.d("xE", xE); D.d("yE", yE); D.d("diag", diag);

   // bench_ceed Problem: all-boundary Dirichlet on every component, f = unit vector / |.|
   Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess;
   ess_bdr = 1; fes.GetEssentialTrueDofs(ess_bdr, ess);
   D.i("ess_tdof", ess);
   Vector uvec(dim); uvec = 1.0; uvec /= uvec.Norml2();
   VectorConstantCoefficient unit_vec(uvec);
   LinearForm b(&fes); b.AddDomainIntegrator(new VectorDomainLFIntegrator(unit_vec)); b.Assemble();
   D.d("b", b);
   GridFunction u(&fes); u = 0.0;
   OperatorPtr A; Vector Bv, X;
   a.FormLinearSystem(ess, u, b, A, X, Bv);
   D.d("B_rhs", Bv);
   { Vector cy(N); A->Mult(x, cy); D.d("y_constrained", cy); }
   OperatorJacobiSmoother M(a, ess);
   CGSolver cg;
   cg.SetRelTol(kind == 0 ? 1e-10 : 1e-6); cg.SetAbsTol(0.0); cg.SetMaxIter(2000); cg.SetPrintLevel(1);
   cg.SetOperator(*A); cg.SetPreconditioner(M); cg.iterative_mode = false;
   ostringstream oss; oss.precision(17);
   mfem::out.SetStream(oss);
   cg.Mult(Bv, X);
   mfem::out.SetStream(cout);
   vector<double> hist;
   {
      istringstream is(oss.str()); string line;
      while (getline(is, line))
      {
         size_t k = line.find("(B r, r) = ");
         if (k != string::npos) { hist.push_back(strtod(line.c_str() + k + 11, nullptr)); }
      }
   }
   D.d("X", X); D.d("cg_hist", hist); D.s("cg_iters", cg.GetNumIterations()); D.s("cg_converged", cg.GetConverged());
   // unpreconditioned CG, 32 iterations at most: bench_ceed.cpp Problem (rtol 1e-12, max_it 32)
   CGSolver cg2;
   cg2.SetRelTol(1e-12); cg2.SetMaxIter(32); cg2.SetPrintLevel(-1); cg2.SetOperator(*A); cg2.iterative_mode = false;
   Vector X2(N); cg2.Mult(Bv, X2);
   D.d("X_cg32", X2); D.s("cg32_iters", cg2.GetNumIterations());
   printf("%s: p=%d kind=%d rule=%d ord=%d D1D=%d Q1D=%d NE=%d vsize=%d its=%d |y|=%.15e |X|=%.15e\n", out.c_str(), p, kind, rule,
          ordering, D1D, Q1D, NE, N, cg.GetNumIterations(), y.Norml2(), X.Norml2());
}

int main(int argc, char **argv)
{
   const string dir = argc > 1 ? argv[1] : "/tmp/golden_vec";
   const string data = "/root/reference/data/";
   char name[256];
   for (int bp = 2; bp <= 6; bp += 2)          // BP2, BP4, BP6
   {
      const int kind = bp == 2 ? 0 : 1, rule = bp == 6 ? 1 : 0;
      for (int p = 1; p <= 4; p++)
      {
         Mesh mesh = Mesh::MakeCartesian3D(2, 2, 2, Element::HEXAHEDRON);
         snprintf(name, sizeof name, "%s/vec_cart2_bp%d_p%d.gld", dir.c_str(), bp, p);
         run_case(name, mesh, p, kind, rule, Ordering::byVDIM);      // bench_ceed's ordering for VDIM = 3
      }
      Mesh mesh = Mesh::MakeCartesian3D(3, 3, 3, Element::HEXAHEDRON);
      snprintf(name, sizeof name, "%s/vec_cart3_bp%d_p2_bynodes.gld", dir.c_str(), bp);
      run_case(name, mesh, 2, kind, rule, Ordering::byNODES);
   }
   {  // curved mesh, default rules (non-registered (D1D,Q1D) for the mass rule)
      Mesh mesh((data + "toroid-hex.mesh").c_str(), 1, 1);
      snprintf(name, sizeof name, "%s/vec_toroid-hex_diff_p2.gld", dir.c_str());
      run_case(name, mesh, 2, 1, 2, Ordering::byVDIM);
      snprintf(name, sizeof name, "%s/vec_toroid-hex_mass_p2.gld", dir.c_str());
      run_case(name, mesh, 2, 0, 2, Ordering::byNODES);
   }
   return 0;
}

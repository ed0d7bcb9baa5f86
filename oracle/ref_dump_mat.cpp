// TEST INFRASTRUCTURE (fixture generator) -- not part of the product.
//
// DiffusionIntegrator with vector and matrix coefficients under AssemblyLevel::PARTIAL
// (PADiffusionSetup3D's coeffDim 3 / 6 / 9 branches, the non-symmetric 9-plane D in the apply
// and diagonal kernels): runs the *reference* (serial MFEM, CPU) on small cases and dumps the
// arrays of the path in the GLD1 container of oracle/ref_dump.cpp; oracle/pack_fixtures.py
// converts them to tests/golden/mat/*.npz.  Linked, like ref_dump.cpp, against the serial
// libmfem.a the survey stage left in /tmp/mfem-build ("Oracle"):
//
//   g++ -O2 -std=c++17 -fno-access-control -I/tmp/mfem-build -I/root/reference \
//       oracle/ref_dump_mat.cpp -o /tmp/ref_dump_mat /tmp/mfem-build/libmfem.a -lrt
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

// the coefficient functions of the fixtures (restated in tests/test_gpu_matcoef.py and the C++
// host test): SPD symmetric part, a skew part for the full matrix
static void vec_fn(const Vector &x, Vector &v)
{
   v.SetSize(3);
   v(0) = 1.0 + x(0)*x(0); v(1) = 2.0 + x(1)*x(2); v(2) = 1.5 + 0.5*x(0)*x(2);
}
static void sym_entries(const Vector &x, double *m /* 11,12,13,22,23,33 */)
{
   m[0] = 2.0 + x(0)*x(0); m[1] = 0.3*x(1); m[2] = 0.2*x(0)*x(2);
   m[3] = 3.0 + x(1)*x(1); m[4] = -0.4*x(2); m[5] = 2.5 + x(0)*x(1);
}
static void sym_fn(const Vector &x, DenseSymmetricMatrix &K)
{
   K.SetSize(3);
   double m[6]; sym_entries(x, m);
   K(0,0) = m[0]; K(0,1) = m[1]; K(0,2) = m[2]; K(1,1) = m[3]; K(1,2) = m[4]; K(2,2) = m[5];
}
static void mat_fn(const Vector &x, DenseMatrix &K)
{
   K.SetSize(3);
   double m[6]; sym_entries(x, m);
   const double s1 = 0.7 + 0.1*x(0), s2 = -0.5*x(1), s3 = 0.25 + 0.2*x(2);   // skew part
   K(0,0) = m[0];      K(0,1) = m[1] + s1; K(0,2) = m[2] + s2;
   K(1,0) = m[1] - s1; K(1,1) = m[3];      K(1,2) = m[4] + s3;
   K(2,0) = m[2] - s2; K(2,1) = m[4] - s3; K(2,2) = m[5];
}

// ctype: 0 vector function, 1 symmetric matrix constant, 2 symmetric matrix function,
//        3 matrix constant, 4 matrix function
static void run_case(const string &out, Mesh &mesh, int p, int ctype)
{
   H1_FECollection fec(p, 3, BasisType::GaussLobatto);
   FiniteElementSpace fes(&mesh, &fec);
   const FiniteElement &el = *fes.GetTypicalFE();
   const IntegrationRule *ir = &DiffusionIntegrator::GetRule(el, el);

   Vector mid(3); mid = 0.4; mid(1) = 0.3; mid(2) = 0.6;
   DenseSymmetricMatrix Ks; sym_fn(mid, Ks);
   DenseMatrix Km; mat_fn(mid, Km);
   VectorFunctionCoefficient vq(3, vec_fn);
   SymmetricMatrixConstantCoefficient smc(Ks);
   SymmetricMatrixFunctionCoefficient smf(3, sym_fn);
   MatrixConstantCoefficient mc(Km);
   MatrixFunctionCoefficient mf(3, mat_fn);

   BilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   DiffusionIntegrator *bfi = nullptr;
   MatrixCoefficient *MQ = nullptr;
   switch (ctype)
   {
      case 0: bfi = new DiffusionIntegrator(vq); break;
      case 1: MQ = &smc; break;
      case 2: MQ = &smf; break;
      case 3: MQ = &mc; break;
      default: MQ = &mf; break;
   }
   if (MQ) { bfi = new DiffusionIntegrator(*MQ); }
   a.AddDomainIntegrator(bfi);
   a.Assemble();

   Dump D(out);
   const DofToQuad &maps = el.GetDofToQuad(*ir, DofToQuad::TENSOR);
   const int D1D = maps.ndof, Q1D = maps.nqpt, NE = fes.GetNE(), ND = el.GetDof();
   D.s("p", p); D.s("ctype", ctype); D.s("D1D", D1D); D.s("Q1D", Q1D); D.s("NE", NE);
   D.s("ndofs", fes.GetNDofs()); D.s("symmetric", bfi->symmetric);
   D.d("B", maps.B); D.d("G", maps.G); D.d("W", ir->GetWeights());
   {
      vector<double> pts(3*ir->GetNPoints());
      for (int q = 0; q < ir->GetNPoints(); q++)
      { pts[3*q] = ir->IntPoint(q).x; pts[3*q+1] = ir->IntPoint(q).y; pts[3*q+2] = ir->IntPoint(q).z; }
      D.d("ip_xyz", pts);
   }
   const auto *R = dynamic_cast<const ElementRestriction*>(fes.GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC));
   MFEM_VERIFY(R, "");
   D.i("gather_map", R->GatherMap()); D.i("offsets", R->Offsets()); D.i("indices", R->Indices());

   const GeometricFactors *geom = mesh.GetGeometricFactors(*ir, GeometricFactors::JACOBIANS | GeometricFactors::COORDINATES);
   const GridFunction *nodes = mesh.GetNodes();
   const FiniteElementSpace *nfes = nodes->FESpace();
   const DofToQuad &nmaps = nfes->GetTypicalFE()->GetDofToQuad(*ir, DofToQuad::TENSOR);
   D.s("nodes_D1D", nmaps.ndof);
   D.d("nodes_B", nmaps.B); D.d("nodes_G", nmaps.G);
   const auto *NR = dynamic_cast<const ElementRestriction*>(nfes->GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC));
   Vector En(NR->Height()); NR->Mult(*nodes, En); D.d("nodes_E", En);
   D.d("J", geom->J);
   D.d("X_q", geom->X);        // physical coordinates of the points, (Q^3, 3, NE)

   // the coefficient as the integrator projects it (bilininteg_diffusion_pa.cpp:127-133)
   {
      QuadratureSpace qs(mesh, *ir);
      CoefficientVector coeff(qs, CoefficientStorage::COMPRESSED);
      if (MQ) { coeff.ProjectTranspose(*MQ); } else { coeff.Project(vq); }
      D.s("coeff_vdim", coeff.GetVDim());
      D.d("coeff", coeff);
   }
   D.d("pa_data", bfi->pa_data);

   const int N = fes.GetVSize();
   Vector x(N), y(N), xE(NE*ND), yE(NE*ND), diag(N), diagE(NE*ND);
   for (int i = 0; i < N; i++) { x(i) = smix(1, i); }
   a.Mult(x, y);
   R->Mult(x, xE); yE = 0.0; bfi->AddMultPA(xE, yE);
   a.AssembleDiagonal(diag);
   diagE = 0.0; bfi->AssembleDiagonalPA(diagE);
   D.d("x", x); D.d("y", y); D.d("xE", xE); D.d("yE", yE); D.d("diag", diag); D.d("diagE", diagE);

   int its = -1;
   if (bfi->symmetric)   // SPD cases: the ex1 solve (all-boundary Dirichlet, f = 1, Jacobi-PCG)
   {
      ConstantCoefficient one(1.0);
      Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess;
      ess_bdr = 1; fes.GetEssentialTrueDofs(ess_bdr, ess);
      D.i("ess_tdof", ess);
      LinearForm b(&fes); b.AddDomainIntegrator(new DomainLFIntegrator(one)); b.Assemble();
      D.d("b", b);
      GridFunction u(&fes); u = 0.0;
      OperatorPtr A; Vector Bv, X;
      a.FormLinearSystem(ess, u, b, A, X, Bv);
      D.d("B_rhs", Bv);
      OperatorJacobiSmoother M(a, ess);
      CGSolver cg;
      cg.SetRelTol(1e-6); cg.SetAbsTol(0.0); cg.SetMaxIter(2000); cg.SetPrintLevel(1);
      cg.SetOperator(*A); cg.SetPreconditioner(M); cg.iterative_mode = false;
      ostringstream oss; oss.precision(17);
      mfem::out.SetStream(oss);
      cg.Mult(Bv, X);
      mfem::out.SetStream(cout);
      vector<double> hist;
      istringstream is(oss.str()); string line;
      while (getline(is, line))
      {
         size_t k = line.find("(B r, r) = ");
         if (k != string::npos) { hist.push_back(strtod(line.c_str() + k + 11, nullptr)); }
      }
      its = cg.GetNumIterations();
      D.d("X", X); D.d("cg_hist", hist); D.s("cg_iters", its); D.s("cg_converged", cg.GetConverged());
   }
   printf("%s: p=%d ctype=%d sym=%d D1D=%d Q1D=%d NE=%d ndofs=%d its=%d |y|=%.15e\n", out.c_str(), p, ctype,
          (int)bfi->symmetric, D1D, Q1D, NE, N, its, y.Norml2());
}

int main(int argc, char **argv)
{
   const string dir = argc > 1 ? argv[1] : "/tmp/golden_raw_mat";
   const string data = "/root/reference/data/";
   const char *cname[5] = {"vec", "symconst", "symfun", "matconst", "matfun"};
   char name[256];
   for (int ct = 0; ct < 5; ct++)
   {
      {
         Mesh mesh = Mesh::MakeCartesian3D(2, 2, 2, Element::HEXAHEDRON);
         snprintf(name, sizeof name, "%s/cart2_%s_p2.gld", dir.c_str(), cname[ct]);
         run_case(name, mesh, 2, ct);
      }
      {
         Mesh mesh((data + "toroid-hex.mesh").c_str(), 1, 1);
         snprintf(name, sizeof name, "%s/toroid-hex_%s_p3.gld", dir.c_str(), cname[ct]);
         run_case(name, mesh, 3, ct);
      }
   }
   {  // one larger order through the non-symmetric path
      Mesh mesh = Mesh::MakeCartesian3D(2, 2, 2, Element::HEXAHEDRON);
      snprintf(name, sizeof name, "%s/cart2_matfun_p5.gld", dir.c_str());
      run_case(name, mesh, 5, 4);
   }
   return 0;
}

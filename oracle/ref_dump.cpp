// TEST INFRASTRUCTURE (fixture generator) -- not part of the product.
//
// Runs the *reference* (serial MFEM, CPU, `-d cpu`) on small cases and dumps every
// intermediate array of the partial-assembly path into a simple binary container
// ("GLD1"), which oracle/pack_fixtures.py converts to tests/golden/*.npz.
//
// It links against a serial libmfem.a.  The reference's own build needs cmake and a
// generated config header, so this repository does NOT build it (see DESIGN.md,
// "Oracle"); this dumper was linked against the serial library the survey stage left
// in /tmp/mfem-build while that existed.  Only the OUTPUT (tests/golden) travels.
//
//   g++ -O2 -std=c++17 -fno-access-control -I/tmp/mfem-build -I/root/reference \
//       oracle/ref_dump.cpp -o /tmp/ref_dump /tmp/mfem-build/libmfem.a -lrt
//
// Arrays dumped follow SURVEY.md section 8c.
#include "mfem.hpp"
#include "fem/integ/bilininteg_diffusion_kernels.hpp"
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
   void i(const string &n, const int *p, int64_t sz) { rec(n, 'i', sz, p, 4); }
   void s(const string &n, int v) { rec(n, 'i', 1, &v, 4); }
};

// Build-owned input generator (SplitMix64): value i of stream `seed` in [0,1).
static double smix(uint64_t seed, uint64_t i)
{
   uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (i + 1) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   z = z ^ (z >> 31);
   return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static double coef_fn(const Vector &x) { return 1.0 + x(0)*x(0) + x(1)*x(2); }

// kind: 0 = mass, 1 = diffusion.  rule: 0 = Gauss-Legendre 2p+3, 1 = Gauss-Lobatto 2p-1,
// 2 = integrator default (IntRule == nullptr, as ex1).
static void run_case(const string &out, Mesh &mesh, int p, int kind, int rule,
                     bool varcoef, bool dumpJ)
{
   const int dim = 3;
   H1_FECollection fec(p, dim, BasisType::GaussLobatto);
   FiniteElementSpace fes(&mesh, &fec);
   const FiniteElement &el = *fes.GetTypicalFE();
   IntegrationRules irs(0, rule == 1 ? Quadrature1D::GaussLobatto : Quadrature1D::GaussLegendre);
   const IntegrationRule *ir = nullptr;
   if (rule == 0) { ir = &irs.Get(Geometry::CUBE, 2*p + 3); }
   if (rule == 1) { ir = &irs.Get(Geometry::CUBE, 2*p - 1); }
   ConstantCoefficient one(1.0);
   FunctionCoefficient fc(coef_fn);
   Coefficient &Q = varcoef ? (Coefficient&)fc : (Coefficient&)one;

   BilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   BilinearFormIntegrator *bfi = nullptr;
   if (kind == 0) { bfi = new MassIntegrator(Q, ir); }
   else { bfi = new DiffusionIntegrator(Q, ir); }
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
   D.s("p", p); D.s("kind", kind); D.s("rule", rule); D.s("varcoef", varcoef);
   D.s("D1D", D1D); D.s("Q1D", Q1D); D.s("NE", NE); D.s("ndofs", fes.GetNDofs());
   D.d("B", maps.B); D.d("G", maps.G); D.d("Bt", maps.Bt); D.d("Gt", maps.Gt);
   D.d("W", ir->GetWeights());
   {
      vector<double> pts(3*ir->GetNPoints());
      for (int q = 0; q < ir->GetNPoints(); q++)
      { pts[3*q] = ir->IntPoint(q).x; pts[3*q+1] = ir->IntPoint(q).y; pts[3*q+2] = ir->IntPoint(q).z; }
      D.d("ip_xyz", pts);
   }
   // mesh topology (for the build's own mesh/fespace generator)
   {
      vector<double> vx(3*mesh.GetNV());
      for (int v = 0; v < mesh.GetNV(); v++) for (int c = 0; c < 3; c++) { vx[3*v+c] = mesh.GetVertex(v)[c]; }
      D.d("vertices", vx);
      Array<int> ev(8*NE), bv(4*mesh.GetNBE()), battr(mesh.GetNBE());
      for (int e = 0; e < NE; e++) { const int *v = mesh.GetElement(e)->GetVertices(); for (int k = 0; k < 8; k++) { ev[8*e+k] = v[k]; } }
      for (int b = 0; b < mesh.GetNBE(); b++)
      {
         const int *v = mesh.GetBdrElement(b)->GetVertices(); for (int k = 0; k < 4; k++) { bv[4*b+k] = v[k]; }
         battr[b] = mesh.GetBdrAttribute(b);
      }
      D.i("elem_vertices", ev); D.i("bdr_vertices", bv); D.i("bdr_attr", battr);
      D.s("nedges", mesh.GetNEdges()); D.s("nfaces", mesh.GetNFaces());
   }
   // element restriction (integer, bit-exact)
   const auto *R = dynamic_cast<const ElementRestriction*>(fes.GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC));
   MFEM_VERIFY(R, "");
   D.i("elem2dof", fes.GetElementToDofTable().GetJ(), (int64_t)NE*ND);
   D.i("dof_map", dynamic_cast<const TensorBasisElement&>(el).GetDofMap());
   D.i("gather_map", R->GatherMap()); D.i("offsets", R->Offsets()); D.i("indices", R->Indices());

   // geometry: nodes -> E-nodes -> J, detJ
   const GeometricFactors *geom = mesh.GetGeometricFactors(*ir, GeometricFactors::JACOBIANS | GeometricFactors::DETERMINANTS);
   const GridFunction *nodes = mesh.GetNodes();
   const FiniteElementSpace *nfes = nodes->FESpace();
   const FiniteElement &nel = *nfes->GetTypicalFE();
   const DofToQuad &nmaps = nel.GetDofToQuad(*ir, DofToQuad::TENSOR);
   D.s("nodes_D1D", nmaps.ndof); D.s("nodes_byvdim", nfes->GetOrdering() == Ordering::byVDIM);
   D.s("nodes_ndofs", nfes->GetNDofs());
   D.d("nodes", *nodes); D.d("nodes_B", nmaps.B); D.d("nodes_G", nmaps.G);
   const auto *NR = dynamic_cast<const ElementRestriction*>(nfes->GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC));
   D.i("nodes_gather_map", NR->GatherMap());
   Vector En(NR->Height()); NR->Mult(*nodes, En); D.d("nodes_E", En);
   if (dumpJ) { D.d("J", geom->J); }
   D.d("detJ", geom->detJ);

   // coefficient at quadrature points + pa_data
   {
      QuadratureSpace qs(mesh, *ir);
      CoefficientVector coeff(qs, CoefficientStorage::COMPRESSED);
      coeff.Project(Q);
      D.d("coeff", coeff);
   }
   if (kind == 0) { D.d("pa_data", static_cast<MassIntegrator*>(bfi)->pa_data); }
   else { D.d("pa_data", static_cast<DiffusionIntegrator*>(bfi)->pa_data); }

   // operator action
   const int N = fes.GetVSize();
   Vector x(N), y(N), xE(NE*ND), yE(NE*ND), diag(N);
   for (int i = 0; i < N; i++) { x(i) = smix(1, i); }
   a.Mult(x, y);
   R->Mult(x, xE); yE = 0.0; bfi->AddMultPA(xE, yE);
   a.AssembleDiagonal(diag);
   Vector diagE(NE*ND); diagE = 0.0; bfi->AssembleDiagonalPA(diagE);
   D.d("x", x); D.d("y", y); D.d("xE", xE); D.d("yE", yE); D.d("diag", diag); D.d("diagE", diagE);

   // linear system: all-boundary Dirichlet, f = 1, Jacobi-PCG (ex1 / bench_ceed Problem)
   Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess;
   ess_bdr = 1; fes.GetEssentialTrueDofs(ess_bdr, ess);
   D.i("ess_tdof", ess);
   LinearForm b(&fes); b.AddDomainIntegrator(new DomainLFIntegrator(one)); b.Assemble();
   D.d("b", b);
   GridFunction u(&fes); u = 0.0;
   OperatorPtr A; Vector Bv, X;
   a.FormLinearSystem(ess, u, b, A, X, Bv);
   D.d("B_rhs", Bv);
   {  // constrained operator action
      Vector cy(N); A->Mult(x, cy); D.d("y_constrained", cy);
   }
   OperatorJacobiSmoother M(a, ess);
   {
      Vector mz(N); M.Mult(x, mz); D.d("jacobi_x", mz);
   }
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
   printf("%s: p=%d kind=%d rule=%d D1D=%d Q1D=%d NE=%d ndofs=%d its=%d |y|=%.15e |X|=%.15e\n", out.c_str(), p, kind, rule,
          D1D, Q1D, NE, N, cg.GetNumIterations(), y.Norml2(), X.Norml2());
}

int main(int argc, char **argv)
{
   const string dir = argc > 1 ? argv[1] : "/tmp/golden_raw";
   const string data = "/root/reference/data/";
   char name[256];
   // Cartesian, BP1/BP3/BP5, p = 1..8 (N = 2; N = 3 for p <= 3)
   for (int p = 1; p <= 8; p++)
   {
      for (int bp = 0; bp < 3; bp++)
      {
         const int kind = bp == 0 ? 0 : 1, rule = bp == 2 ? 1 : 0;
         for (int N = 2; N <= (p <= 3 ? 3 : 2); N++)
         {
            Mesh mesh = Mesh::MakeCartesian3D(N, N, N, Element::HEXAHEDRON);
            snprintf(name, sizeof name, "%s/cart%d_bp%d_p%d.gld", dir.c_str(), N, 2*bp + 1, p);
            run_case(name, mesh, p, kind, rule, false, p <= 3);
         }
      }
   }
   {  // anisotropic box, SFC ordering on a non-cubic grid, default rule (ex1 path)
      Mesh mesh = Mesh::MakeCartesian3D(3, 2, 4, Element::HEXAHEDRON, 1.5, 1.0, 2.0);
      snprintf(name, sizeof name, "%s/box324_diff_p2.gld", dir.c_str());
      run_case(name, mesh, 2, 1, 2, false, true);
      snprintf(name, sizeof name, "%s/box324_mass_p3.gld", dir.c_str());
      run_case(name, mesh, 3, 0, 2, false, true);
   }
   {  // lexicographic element ordering
      Mesh mesh = Mesh::MakeCartesian3D(3, 3, 2, Element::HEXAHEDRON, 1.0, 1.0, 1.0, false);
      snprintf(name, sizeof name, "%s/box332lex_diff_p2.gld", dir.c_str());
      run_case(name, mesh, 2, 1, 2, false, false);
   }
   // curved hex meshes (substitutes for BASELINE config 5), default rule, constant + variable coefficient
   const char *files[2] = {"fichera-q3", "toroid-hex"};
   for (int m = 0; m < 2; m++)
   {
      for (int p = 2; p <= 4; p++)
      {
         Mesh mesh((data + files[m] + ".mesh").c_str(), 1, 1);
         snprintf(name, sizeof name, "%s/%s_diff_p%d.gld", dir.c_str(), files[m], p);
         run_case(name, mesh, p, 1, 2, false, p == 2);
      }
      Mesh mesh((data + files[m] + ".mesh").c_str(), 1, 1);
      snprintf(name, sizeof name, "%s/%s_diffvar_p3.gld", dir.c_str(), files[m]);
      run_case(name, mesh, 3, 1, 2, true, false);
      snprintf(name, sizeof name, "%s/%s_massvar_p2.gld", dir.c_str(), files[m]);
      run_case(name, mesh, 2, 0, 2, true, false);
      snprintf(name, sizeof name, "%s/%s_bp5_p3.gld", dir.c_str(), files[m]);
      run_case(name, mesh, 3, 1, 1, false, false);
   }
   {  // one refined curved case (uniform refinement of curved nodes)
      Mesh mesh((data + "fichera-q3.mesh").c_str(), 1, 1);
      mesh.UniformRefinement();
      snprintf(name, sizeof name, "%s/fichera-q3-r1_diff_p2.gld", dir.c_str());
      run_case(name, mesh, 2, 1, 2, false, false);
   }
   return 0;
}

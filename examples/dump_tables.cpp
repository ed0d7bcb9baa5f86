// dump_tables -- writes the host-side tables of a mesh + H1 space (mesh topology, dof
// numbering, element restriction, essential dofs, basis/quadrature tables, mesh nodes) as raw
// binary files, so the tests can compare them bit for bit with the reference's own tables
// (tests/golden/*.npz).  Host-only: never touches the GPU.
//
//   dump_tables <outdir> cart <nx> <ny> <nz> <sx> <sy> <sz> <sfc 0|1> <p> <rule 0 GL(2p+3) | 1 GLL(2p-1) | 2 default> <kind 0 mass|1 diffusion>
//   dump_tables <outdir> file <mesh file> <refinements> <p> <rule> <kind>
// Two optional trailing arguments "<varcoef 0|1> run" also run the device path through the C++
// API (operator action on the reference's SplitMix input, diagonal, RHS, constrained
// operator, Jacobi, PCG) and write the results next to the tables (needs the GPU).
#include "mfem.hpp"
#include <cstdio>
using namespace mfem;

static std::string outdir;
template <class T> static void dump(const char *name, const T *p, size_t n)
{
   const std::string f = outdir + "/" + name;
   FILE *fp = fopen(f.c_str(), "wb");
   if (!fp || fwrite(p, sizeof(T), n, fp) != n) { MFEM_ABORT("cannot write " << f); }
   fclose(fp);
}

static double smix(uint64_t seed, uint64_t i)
{
   uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (i + 1) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   z = z ^ (z >> 31);
   return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static double coef_fn(const Vector &x) { return 1.0 + x(0)*x(0) + x(1)*x(2); }

// the coefficient functions of tests/golden/mat (oracle/ref_dump_mat.cpp evaluates the same expressions
// through the reference's classes)
static void vec_fn(const Vector &x, Vector &v)
{
   v(0) = 1.0 + x(0)*x(0); v(1) = 2.0 + x(1)*x(2); v(2) = 1.5 + 0.5*x(0)*x(2);
}
static void sym_entries(const Vector &x, double *m)
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
   const double s1 = 0.7 + 0.1*x(0), s2 = -0.5*x(1), s3 = 0.25 + 0.2*x(2);
   K(0,0) = m[0];      K(0,1) = m[1] + s1; K(0,2) = m[2] + s2;
   K(1,0) = m[1] - s1; K(1,1) = m[3];      K(1,2) = m[4] + s3;
   K(2,0) = m[2] - s2; K(2,1) = m[4] - s3; K(2,2) = m[5];
}

int main(int argc, char **argv)
{
   if (argc < 4) { std::cerr << "usage: see the header of dump_tables.cpp\n"; return 2; }
   outdir = argv[1];
   const std::string mode = argv[2];
   Mesh *mesh = nullptr;
   int a = 3, p, rule, kind;
   if (mode == "cart")
   {
      const int nx = atoi(argv[a]), ny = atoi(argv[a+1]), nz = atoi(argv[a+2]);
      const double sx = atof(argv[a+3]), sy = atof(argv[a+4]), sz = atof(argv[a+5]);
      const bool sfc = atoi(argv[a+6]) != 0;
      mesh = new Mesh(Mesh::MakeCartesian3D(nx, ny, nz, Element::HEXAHEDRON, sx, sy, sz, sfc));
      a += 7;
   }
   else
   {
      mesh = new Mesh(argv[a], 1, 1);
      for (int l = 0; l < atoi(argv[a+1]); l++) { mesh->UniformRefinement(); }
      a += 2;
   }
   p = atoi(argv[a]); rule = atoi(argv[a+1]); kind = atoi(argv[a+2]);
   H1_FECollection fec(p, 3);
   FiniteElementSpace fes(mesh, &fec);
   const FiniteElement &el = *fes.GetTypicalFE();
   const int NE = fes.GetNE(), ND = el.GetDof();
   IntegrationRules gll(0, Quadrature1D::GaussLobatto);
   mesh->EnsureNodes();
   const int nodes_order = mesh->GetNodalFESpace()->GetMaxElementOrder();
   const IntegrationRule *ir = (rule == 0) ? &IntRules.Get(Geometry::CUBE, 2*p + 3)
                               : (rule == 1) ? &gll.Get(Geometry::CUBE, 2*p - 1)
                               : (kind == 0) ? &MassIntegrator::GetRule(el, el, nodes_order) : &DiffusionIntegrator::GetRule(el, el);
   const DofToQuad &maps = el.GetDofToQuad(*ir, DofToQuad::TENSOR);
   {
      std::vector<double> vx(3*mesh->GetNV());
      for (int v = 0; v < mesh->GetNV(); v++) for (int c = 0; c < 3; c++) { vx[3*v + c] = mesh->GetVertex(v)[c]; }
      dump("vertices.f64", vx.data(), vx.size());
      std::vector<int> ev(8*NE), bv(4*mesh->GetNBE()), ba(mesh->GetNBE());
      for (int e = 0; e < NE; e++) { std::copy(mesh->GetElementVertices(e), mesh->GetElementVertices(e) + 8, &ev[8*e]); }
      for (int b = 0; b < mesh->GetNBE(); b++) { std::copy(mesh->GetBdrElementVertices(b), mesh->GetBdrElementVertices(b) + 4, &bv[4*b]); ba[b] = mesh->GetBdrAttribute(b); }
      dump("elem_vertices.i32", ev.data(), ev.size()); dump("bdr_vertices.i32", bv.data(), bv.size()); dump("bdr_attr.i32", ba.data(), ba.size());
   }
   const int meta[8] = { NE, fes.GetNDofs(), mesh->GetNEdges(), mesh->GetNFaces(), maps.ndof, maps.nqpt,
                         mesh->GetNodalFESpace()->GetMaxElementOrder() + 1, mesh->GetNodalFESpace()->GetNDofs()
                       };
   dump("meta.i32", meta, 8);
   dump("elem2dof.i32", fes.ElementDofs(0), (size_t)NE*ND);
   dump("dof_map.i32", el.GetDofMap().HostRead(), ND);
   const ElementRestriction *R = fes.GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC);
   dump("gather_map.i32", R->GatherMap().HostRead(), R->GatherMap().Size());
   dump("offsets.i32", R->Offsets().HostRead(), R->Offsets().Size());
   dump("indices.i32", R->Indices().HostRead(), R->Indices().Size());
   Array<int> ess_bdr(mesh->bdr_attributes.Max()), ess;
   ess_bdr = 1;
   fes.GetEssentialTrueDofs(ess_bdr, ess);
   dump("ess_tdof.i32", ess.HostRead(), ess.Size());
   dump("B.f64", maps.B.HostRead(), maps.B.Size()); dump("G.f64", maps.G.HostRead(), maps.G.Size());
   dump("Bt.f64", maps.Bt.HostRead(), maps.Bt.Size()); dump("Gt.f64", maps.Gt.HostRead(), maps.Gt.Size());
   dump("W.f64", ir->GetWeights().HostRead(), ir->GetNPoints());
   const GridFunction *nodes = mesh->GetNodes();
   dump("nodes.f64", nodes->HostRead(), nodes->Size());
   const ElementRestriction *NR = nodes->FESpace()->GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC);
   dump("nodes_gather_map.i32", NR->GatherMap().HostRead(), NR->GatherMap().Size());
   const int nodes_byvdim = nodes->FESpace()->GetOrdering() == Ordering::byVDIM ? 1 : 0;
   dump("nodes_byvdim.i32", &nodes_byvdim, 1);
   const DofToQuad &nm = nodes->FESpace()->GetTypicalFE()->GetDofToQuad(*ir, DofToQuad::TENSOR);
   dump("nodes_B.f64", nm.B.HostRead(), nm.B.Size()); dump("nodes_G.f64", nm.G.HostRead(), nm.G.Size());
   printf("NE=%d ndofs=%d nedges=%d nfaces=%d D1D=%d Q1D=%d ness=%d\n", NE, fes.GetNDofs(), mesh->GetNEdges(), mesh->GetNFaces(),
          maps.ndof, maps.nqpt, ess.Size());
   if (argc > a + 4 && std::string(argv[a+4]) == "run")
   {
      const bool varcoef = atoi(argv[a+3]) != 0;
      Device device("hip");
      ConstantCoefficient one(1.0);
      FunctionCoefficient fc(coef_fn);
      // <varcoef> 7: the same function, handed over as its values at the quadrature points (QuadratureFunctionCoefficient)
      const bool qfmode = atoi(argv[a+3]) == 7;
      QuadratureSpace qspace(*mesh, *ir);
      QuadratureFunction qfun(&qspace);
      if (qfmode) { fc.Project(qfun); }
      QuadratureFunctionCoefficient qfc(qfun);
      Coefficient &Q = qfmode ? (Coefficient&)qfc : varcoef ? (Coefficient&)fc : (Coefficient&)one;
      // <varcoef> 2..6: diffusion with a vector function / symmetric matrix constant / symmetric matrix
      // function / matrix constant / matrix function coefficient (the cases of tests/golden/mat)
      const int ctype = qfmode ? -1 : atoi(argv[a+3]) - 2;
      Vector mid(3); mid(0) = 0.4; mid(1) = 0.3; mid(2) = 0.6;
      DenseSymmetricMatrix Ks; sym_fn(mid, Ks);
      DenseMatrix Km; mat_fn(mid, Km);
      VectorFunctionCoefficient vq(3, vec_fn);
      SymmetricMatrixConstantCoefficient smc(Ks);
      SymmetricMatrixFunctionCoefficient smf(3, sym_fn);
      MatrixConstantCoefficient mc(Km);
      MatrixFunctionCoefficient mf(3, mat_fn);
      MatrixCoefficient *MQs[5] = { nullptr, &smc, &smf, &mc, &mf };
      const bool nonsym = ctype >= 3;
      BilinearForm bf(&fes);
      bf.SetAssemblyLevel(AssemblyLevel::PARTIAL);
      const IntegrationRule *irp = (rule == 2 && !qfmode) ? nullptr : ir;
      BilinearFormIntegrator *bfi = (kind == 0) ? (BilinearFormIntegrator*)new MassIntegrator(Q, irp)
                                    : ctype == 0 ? (BilinearFormIntegrator*)new DiffusionIntegrator(vq, irp)
                                    : ctype > 0 ? (BilinearFormIntegrator*)new DiffusionIntegrator(*MQs[ctype], irp)
                                    : (BilinearFormIntegrator*)new DiffusionIntegrator(Q, irp);
      // trailing "markers": the integrator restricted to the elements of attribute 2, attributes 1 + e%2
      // (tests/unit/fem/test_pa_kernels.cpp "PA Markers")
      const bool use_markers = argc > a + 5 && std::string(argv[a+5]) == "markers";
      Array<int> marker(2); marker[0] = 0; marker[1] = 1;
      if (use_markers)
      {
         for (int e = 0; e < NE; e++) { mesh->SetAttribute(e, 1 + e%2); }
         mesh->SetAttributes();
         bf.AddDomainIntegrator(bfi, marker);
      }
      else { bf.AddDomainIntegrator(bfi); }
      bf.Assemble();
      const int N = fes.GetVSize();
      Vector x(N), y(N), xE(NE*ND), yE(NE*ND), diag(N), diagE(NE*ND);
      for (int i = 0; i < N; i++) { x(i) = smix(1, i); }
      bf.Mult(x, y);
      R->Mult(x, xE); yE = 0.0; bfi->AddMultPA(xE, yE);          // E-vector level (dispatch-table signature)
      bf.AssembleDiagonal(diag);
      diagE = 0.0; bfi->AssembleDiagonalPA(diagE);
      dump("y.f64", y.HostRead(), N); dump("xE.f64", xE.HostRead(), xE.Size()); dump("yE.f64", yE.HostRead(), yE.Size());
      dump("diag.f64", diag.HostRead(), N); dump("diagE.f64", diagE.HostRead(), diagE.Size());
      if (use_markers) { printf("run: markers |y|=%.15e\n", y.Norml2()); delete mesh; return 0; }
      if (nonsym)      // no symmetric solve for the full matrix; the operator through the ConstrainedOperator seam
      {
         Array<int> none;
         OperatorPtr A0; bf.FormSystemMatrix(none, A0);
         Vector cy(N); A0->Mult(x, cy); dump("y_constrained.f64", cy.HostRead(), N);
         // trailing "transpose": MultTranspose of a non-symmetric form aborts, as in the reference
         if (argc > a + 5 && std::string(argv[a+5]) == "transpose") { Vector ty(N); bf.MultTranspose(x, ty); }
         printf("run: non-symmetric D |y|=%.15e\n", y.Norml2());
         delete mesh;
         return 0;
      }
      LinearForm b(&fes);
      b.AddDomainIntegrator(new DomainLFIntegrator(one));
      b.Assemble();
      dump("b.f64", b.HostRead(), N);
      GridFunction u(&fes); u = 0.0;
      OperatorPtr A; Vector Bv, X;
      bf.FormLinearSystem(ess, u, b, A, X, Bv);
      dump("B_rhs.f64", Bv.HostRead(), N);
      Vector cy(N); A->Mult(x, cy); dump("y_constrained.f64", cy.HostRead(), N);
      OperatorJacobiSmoother M(bf, ess);
      Vector mz(N); M.Mult(x, mz); dump("jacobi_x.f64", mz.HostRead(), N);
      CGSolver cg;
      cg.SetRelTol(kind == 0 ? 1e-10 : 1e-6); cg.SetAbsTol(0.0); cg.SetMaxIter(2000); cg.SetPrintLevel(-1);
      cg.SetOperator(*A); cg.SetPreconditioner(M); cg.iterative_mode = false;
      cg.Mult(Bv, X);
      dump("X.f64", X.HostRead(), N);
      dump("cg_hist.f64", cg.ResidualHistory().data(), cg.ResidualHistory().size());
      // the same solve through the generic (unfused) CGSolver path: an operator the fused path does not recognise
      struct Wrap : Operator { const Operator &A; Wrap(const Operator &a) : Operator(a.Height()), A(a) {} void Mult(const Vector &x, Vector &y) const override { A.Mult(x, y); } } wrap(*A);
      CGSolver cg2;
      cg2.SetRelTol(kind == 0 ? 1e-10 : 1e-6); cg2.SetAbsTol(0.0); cg2.SetMaxIter(2000); cg2.SetPrintLevel(-1);
      cg2.SetOperator(wrap); cg2.SetPreconditioner(M); cg2.iterative_mode = false;
      Vector X2(N); cg2.Mult(Bv, X2);
      dump("X_generic.f64", X2.HostRead(), N);
      const int its[4] = { cg.GetNumIterations(), (int)cg.GetConverged(), cg2.GetNumIterations(), (int)cg2.GetConverged() };
      dump("cg_iters.i32", its, 4);
      printf("run: its=%d (generic %d) |y|=%.15e |X|=%.15e\n", its[0], its[2], y.Norml2(), X.Norml2());
   }
   delete mesh;
   return 0;
}

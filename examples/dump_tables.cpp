// dump_tables -- writes the host-side tables of a mesh + H1 space (mesh topology, dof
// numbering, element restriction, essential dofs, basis/quadrature tables, mesh nodes) as raw
// binary files, so the tests can compare them bit for bit with the reference's own tables
// (tests/golden/*.npz).  Host-only: never touches the GPU.
//
//   dump_tables <outdir> cart <nx> <ny> <nz> <sx> <sy> <sz> <sfc 0|1> <p> <rule 0 GL(2p+3) | 1 GLL(2p-1) | 2 default> <kind 0 mass|1 diffusion>
//   dump_tables <outdir> file <mesh file> <refinements> <p> <rule> <kind>
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
   const DofToQuad &nm = nodes->FESpace()->GetTypicalFE()->GetDofToQuad(*ir, DofToQuad::TENSOR);
   dump("nodes_B.f64", nm.B.HostRead(), nm.B.Size()); dump("nodes_G.f64", nm.G.HostRead(), nm.G.Size());
   printf("NE=%d ndofs=%d nedges=%d nfaces=%d D1D=%d Q1D=%d ness=%d\n", NE, fes.GetNDofs(), mesh->GetNEdges(), mesh->GetNFaces(),
          maps.ndof, maps.nqpt, ess.Size());
   delete mesh;
   return 0;
}

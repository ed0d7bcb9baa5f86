// C entry points of the host layer for drivers that are not C++ (include/mfem_amd_host.h).
#include "mfem.hpp"
#include "../../include/mfem_amd_host.h"
#include <vector>
using namespace mfem;

namespace
{
struct HostTables
{
   std::vector<int> gather_map, ess;
   std::vector<double> nodes_E;
};
}

extern "C" int mfem_amd_host_cartesian_tables(int p, int nx, int ny, int nz, double sx, double sy, double sz,
                                              double ox, double oy, double oz, int sfc, mfem_amd_host_tables *out)
{
   if (!out || p < 1 || nx < 1 || ny < 1 || nz < 1)
   {
      std::cerr << "mfem_amd_host_cartesian_tables: bad arguments\n";
      return 1;
   }
   Mesh mesh = Mesh::MakeCartesian3D(nx, ny, nz, Element::HEXAHEDRON, sx, sy, sz, sfc != 0);
   H1_FECollection fec(p, 3);
   FiniteElementSpace fes(&mesh, &fec);
   const ElementRestriction *R = fes.GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC);
   HostTables *h = new HostTables;
   const int ne = fes.GetNE();
   h->gather_map.assign(R->GatherMap().HostRead(), R->GatherMap().HostRead() + R->GatherMap().Size());
   Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess;
   ess_bdr = 1;
   fes.GetEssentialTrueDofs(ess_bdr, ess);
   h->ess.assign(ess.HostRead(), ess.HostRead() + ess.Size());
   // hexahedron vertices (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) ... (fem/geom.cpp) in lexicographic order, x fastest
   static const int lex[8] = {0, 1, 3, 2, 4, 5, 7, 6};
   const double org[3] = {ox, oy, oz};
   h->nodes_E.resize((size_t)ne*24);
   for (int e = 0; e < ne; e++)
   {
      const int *v = mesh.GetElementVertices(e);
      for (int c = 0; c < 3; c++)
      {
         for (int k = 0; k < 8; k++) { h->nodes_E[(size_t)e*24 + c*8 + k] = org[c] + mesh.GetVertex(v[lex[k]])[c]; }
      }
   }
   out->ne = ne; out->ndofs = fes.GetNDofs(); out->ness = (int64_t)h->ess.size(); out->D1D = p + 1;
   out->gather_map = h->gather_map.data(); out->nodes_E = h->nodes_E.data(); out->ess = h->ess.data();
   out->owner = h;
   return 0;
}

extern "C" void mfem_amd_host_tables_free(mfem_amd_host_tables *t)
{
   if (t && t->owner) { delete static_cast<HostTables*>(t->owner); t->owner = nullptr; }
}

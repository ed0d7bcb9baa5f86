// parallel.cpp -- the parallel layer of the MFEM API surface: Mpi, ParMesh, ParFiniteElementSpace, ParLinearForm,
// ParBilinearForm, the parallel PA operator.  One process per GPU; the exchange and the all-reduce are the RCCL
// transport of libmfem_amd_comm.so (include/mfem_amd_comm.h), driven from the C ABI (mfem_amd_pa_par_mult,
// mfem_amd_pcg_pa_par) -- nothing of an iteration runs on the host.
//
// Reference: mesh/pmesh.cpp:106-170 (ParMesh from a serial mesh + partitioning), fem/pfespace.cpp:1804-1851
// (true-dof ownership), :5266-5345 (the device prolongation's tables), general/communication.cpp:120-124 (group
// master = lowest rank), fem/pbilinearform.cpp:293-316, 475- (diagonal, FormLinearSystem), examples/ex1p.cpp.
// The parallel reference needs MPI + hypre + METIS, none on the target image: the semantics are restated from
// the text and pinned to the serial path (a partitioned solve = the serial solve).
#include "mfem.hpp"
#include <limits>
#include <map>

namespace mfem
{

// ---- Mpi ----------------------------------------------------------------------------------------
namespace
{
mfem_amd_world *g_world = nullptr;
int env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
}
int Mpi::WorldRank() { return g_world ? mfem_amd_world_rank(g_world) : env_int("RANK", 0); }
int Mpi::WorldSize() { return g_world ? mfem_amd_world_size(g_world) : env_int("WORLD_SIZE", 1); }
mfem_amd_world *Mpi::World()
{
   if (!g_world)
   {
      Device::Require();      // the RCCL communicator binds to the current device
      if (mfem_amd_world_create(-1, 0, nullptr, 0, 41, -1, &g_world) != 0)
      { MFEM_ABORT("mfem_amd: cannot create the process group: " << mfem_amd_comm_last_error()); }
      atexit([]() { Mpi::Finalize(); });
   }
   return g_world;
}
void Mpi::Finalize()
{
   if (g_world) { mfem_amd_world_destroy(g_world); g_world = nullptr; }
}

// ---- ParMesh ------------------------------------------------------------------------------------
ParMesh::ParMesh(MPI_Comm comm_, Mesh &mesh, const int *partitioning)
   : comm(comm_), rank(Mpi::WorldRank()), nranks(Mpi::WorldSize())
{
   MFEM_VERIFY(mesh.GetNE() > 0, "ParMesh: empty serial mesh");
   serial.Swap(mesh);
   const int ne = serial.GetNE();
   partition.resize((size_t)ne);
   for (int e = 0; e < ne; e++)
   {
      partition[(size_t)e] = partitioning ? partitioning[e] : (int)((int64_t)e*nranks/ne);
      MFEM_VERIFY(partition[(size_t)e] >= 0 && partition[(size_t)e] < nranks, "ParMesh: partitioning[" << e << "] is not a rank");
   }
   BuildLocal();
}

// the rank's elements as a mesh of their own: vertices renumbered in ascending global order (edge orientations,
// which compare vertex ids, stay those of the serial mesh), the boundary elements of the rank's elements, and the
// rank's part of the nodal grid function
void ParMesh::BuildLocal()
{
   Mesh loc;
   local_elems.clear();
   const int ne = serial.GetNE();
   // interior elements first (stable within each class): an element is on the interface when one of its vertices
   // also belongs to an element of another rank -- every shared edge, face and dof lies on such a vertex
   {
      std::vector<char> foreign((size_t)serial.GetNV(), 0);
      for (int e = 0; e < ne; e++)
      {
         if (partition[(size_t)e] == rank) { continue; }
         for (int k = 0; k < 8; k++) { foreign[(size_t)serial.elements[(size_t)8*e + k]] = 1; }
      }
      std::vector<int> iface;
      for (int e = 0; e < ne; e++)
      {
         if (partition[(size_t)e] != rank) { continue; }
         bool on = false;
         for (int k = 0; k < 8; k++) { on = on || foreign[(size_t)serial.elements[(size_t)8*e + k]]; }
         (on ? iface : local_elems).push_back(e);
      }
      n_interior = (int)local_elems.size();
      local_elems.insert(local_elems.end(), iface.begin(), iface.end());
   }
   std::vector<int> vmap((size_t)serial.GetNV(), -1);
   for (int e : local_elems) { for (int k = 0; k < 8; k++) { vmap[(size_t)serial.elements[(size_t)8*e + k]] = 0; } }
   int nv = 0;
   for (size_t v = 0; v < vmap.size(); v++) { if (vmap[v] == 0) { vmap[v] = nv++; } }
   loc.vertices.resize((size_t)3*nv);
   for (size_t v = 0; v < vmap.size(); v++)
   {
      if (vmap[v] >= 0) { for (int c = 0; c < 3; c++) { loc.vertices[(size_t)3*vmap[v] + c] = serial.vertices[3*v + c]; } }
   }
   for (int e : local_elems)
   {
      for (int k = 0; k < 8; k++) { loc.elements.push_back(vmap[(size_t)serial.elements[(size_t)8*e + k]]); }
      loc.elem_attr.push_back(serial.elem_attr[(size_t)e]);
   }
   for (int b = 0; b < serial.GetNBE(); b++)
   {
      const int f = serial.be_to_face[(size_t)b];
      const int e1 = serial.face_elem1[(size_t)f], e2 = serial.face_elem2[(size_t)f];
      const bool mine = (e1 >= 0 && partition[(size_t)e1] == rank) || (e2 >= 0 && partition[(size_t)e2] == rank);
      if (!mine) { continue; }
      for (int k = 0; k < 4; k++) { loc.boundary.push_back(vmap[(size_t)serial.boundary[(size_t)4*b + k]]); }
      loc.bdr_attr.push_back(serial.bdr_attr[(size_t)b]);
   }
   loc.Dim = serial.Dim; loc.spaceDim = serial.spaceDim;
   loc.FinalizeTopology();
   loc.SetAttributes();
   if (const GridFunction *nodes = serial.GetNodes())
   {
      std::vector<real_t> xe, xl;
      const int Dn = internal::ElementNodesLex(serial, xe);
      const size_t blk = (size_t)3*Dn*Dn*Dn;
      xl.resize(blk*local_elems.size());
      for (size_t le = 0; le < local_elems.size(); le++)
      { std::copy(xe.begin() + blk*(size_t)local_elems[le], xe.begin() + blk*((size_t)local_elems[le] + 1), xl.begin() + blk*le); }
      const FiniteElementSpace *nfes = nodes->FESpace();
      loc.SetNodesFromElementValues(Dn - 1, (int)nfes->GetOrdering(), nfes->GetTypicalFE()->Nodes1D(), xl);
   }
   const long seq = sequence;
   Mesh::Swap(loc);
   sequence = seq + 1;
}

void ParMesh::UniformRefinement(int ref_algo)
{
   serial.UniformRefinement(ref_algo);
   // the children of element i are elements 8i .. 8i+7 (Mesh::UniformRefinement)
   std::vector<int> np((size_t)8*partition.size());
   for (size_t i = 0; i < partition.size(); i++) { for (int k = 0; k < 8; k++) { np[8*i + k] = partition[i]; } }
   partition.swap(np);
   BuildLocal();
}

// ---- ParFiniteElementSpace ----------------------------------------------------------------------
ParFiniteElementSpace::ParFiniteElementSpace(ParMesh *pm, const FiniteElementCollection *fec_, int vdim_, int ordering_)
   : FiniteElementSpace(pm, fec_, vdim_, ordering_), pmesh(pm)
{
   std::memset(&cb, 0, sizeof cb);
   BuildSharedTables();
}
ParFiniteElementSpace::~ParFiniteElementSpace()
{
   if (exchange) { mfem_amd_exchange_destroy(exchange); }
}

int ParFiniteElementSpace::LocalDof(int64_t g) const
{
   auto it = std::lower_bound(g2l.begin(), g2l.end(), std::make_pair(g, -1));
   return (it != g2l.end() && it->first == g) ? it->second : -1;
}

// Ownership and the exchange lists from the serial mesh and the partition (the same on every rank): a dof's group = the
// ranks whose elements touch it, its owner = the lowest of them (general/communication.cpp:120-124); lists per neighbour
// in ascending global dof id on both sides.  The serial SPACE is never built: its numbering is evaluated element by
// element (FiniteElementSpace::ElementDofsOf) for the rank's own elements and for the elements of other ranks that share a
// vertex with them -- every element that touches one of the rank's dofs is among those -- so the work and the memory
// beyond the serial mesh itself are those of the rank's part and its surface, whatever the number of ranks.
void ParFiniteElementSpace::BuildSharedTables()
{
   const int me = pmesh->GetMyRank();
   const Mesh &sm = pmesh->SerialMesh();
   const H1_FECollection *h1c = dynamic_cast<const H1_FECollection*>(FEColl());
   MFEM_VERIFY(h1c, "mfem_amd: only H1_FECollection spaces");
   const std::vector<int> &part = pmesh->Partitioning(), &lel = pmesh->LocalElements();
   const int nd = GetTypicalFE()->GetDof(), gne = sm.GetNE(), nl = GetNDofs();
   global_ndofs = FiniteElementSpace::NDofsOf(sm, h1c->GetOrder());
   // local -> global: the j-th dof of an element is the same node in both numberings
   l2g64.assign((size_t)nl, -1);
   std::vector<int64_t> gd((size_t)nd);
   for (size_t le = 0; le < lel.size(); le++)
   {
      const int *ld = ElementDofs((int)le);
      FiniteElementSpace::ElementDofsOf(sm, *h1c, lel[le], gd.data());
      for (int j = 0; j < nd; j++) { l2g64[(size_t)ld[j]] = gd[(size_t)j]; }
   }
   for (int i = 0; i < nl; i++) { MFEM_VERIFY(l2g64[(size_t)i] >= 0, "local dof without an element"); }
   // global -> local, for the dofs that can be shared at all: those of the rank's interface elements (the local elements
   // after the first InteriorElements(): a dof another rank touches lies on an entity whose vertices that rank touches)
   g2l.clear();
   for (int le = pmesh->InteriorElements(); le < (int)lel.size(); le++)
   {
      const int *ld = ElementDofs(le);
      for (int j = 0; j < nd; j++) { g2l.emplace_back(l2g64[(size_t)ld[j]], ld[j]); }
   }
   std::sort(g2l.begin(), g2l.end());
   g2l.erase(std::unique(g2l.begin(), g2l.end()), g2l.end());
   // the elements of other ranks around the rank's part
   std::vector<char> mine((size_t)sm.GetNV(), 0);
   for (int e : lel) { for (int k = 0; k < 8; k++) { mine[(size_t)sm.GetElementVertices(e)[k]] = 1; } }
   std::vector<int> halo;
   for (int e = 0; e < gne; e++)
   {
      if (part[(size_t)e] == me) { continue; }
      const int *v = sm.GetElementVertices(e);
      bool near = false;
      for (int k = 0; k < 8; k++) { near = near || mine[(size_t)v[k]]; }
      if (near) { halo.push_back(e); }
   }
   std::vector<int> owner((size_t)nl, me);
   for (int e : halo)
   {
      const int r = part[(size_t)e];
      FiniteElementSpace::ElementDofsOf(sm, *h1c, e, gd.data());
      for (int j = 0; j < nd; j++)
      {
         const int li = LocalDof(gd[(size_t)j]);
         if (li >= 0 && r < owner[(size_t)li]) { owner[(size_t)li] = r; }
      }
   }
   // upper: (sharing rank, global dof) for the dofs I own that other ranks touch; lower: my ghosts by owner
   std::vector<std::pair<int, int64_t>> up, lo;
   for (int e : halo)
   {
      const int r = part[(size_t)e];
      FiniteElementSpace::ElementDofsOf(sm, *h1c, e, gd.data());
      for (int j = 0; j < nd; j++)
      {
         const int li = LocalDof(gd[(size_t)j]);
         if (li >= 0 && owner[(size_t)li] == me) { up.emplace_back(r, gd[(size_t)j]); }
      }
   }
   std::sort(up.begin(), up.end());
   up.erase(std::unique(up.begin(), up.end()), up.end());
   n_owned = 0;
   for (int i = 0; i < nl; i++)
   {
      const int o = owner[(size_t)i];
      if (o == me) { n_owned += GetVDim(); }
      else { lo.emplace_back(o, l2g64[(size_t)i]); }
   }
   std::sort(lo.begin(), lo.end());
   // vector spaces (fem/pfespace.cpp:5266-5345 builds its tables on vdofs): every shared dof carries its vd components,
   // next to each other in the messages, at the vdofs the space's ordering gives them
   const int vd = GetVDim();
   auto flatten = [&](const std::vector<std::pair<int, int64_t>> &prs, std::vector<int> &ranks, std::vector<int> &sizes, Array<int> &ids)
   {
      ranks.clear(); sizes.clear();
      ids.SetSize((int)prs.size()*vd);
      int *p = ids.HostWrite();
      for (size_t k = 0; k < prs.size(); k++)
      {
         if (ranks.empty() || ranks.back() != prs[k].first) { ranks.push_back(prs[k].first); sizes.push_back(0); }
         sizes.back() += vd;
         for (int c = 0; c < vd; c++) { p[k*vd + c] = DofToVDof(LocalDof(prs[k].second), c); }
      }
   };
   flatten(up, up_ranks, up_sizes, shr_ltdof);
   flatten(lo, lo_ranks, lo_sizes, ext_ldof);
   // CSR of the deterministic accumulate (fem/pfespace.cpp:5475-5494): every distinct owned-shared dof and the
   // positions of its copies in the receive buffer, ascending (= ascending neighbour rank)
   const int ns = shr_ltdof.Size();
   std::vector<std::pair<int, int>> byd((size_t)ns);
   for (int k = 0; k < ns; k++) { byd[(size_t)k] = {shr_ltdof[k], k}; }
   std::sort(byd.begin(), byd.end());
   unq_shr_j.SetSize(ns);
   std::vector<int> ud, ui;
   for (int k = 0; k < ns; k++)
   {
      if (k == 0 || byd[(size_t)k].first != byd[(size_t)k - 1].first) { ud.push_back(byd[(size_t)k].first); ui.push_back(k); }
      unq_shr_j[k] = byd[(size_t)k].second;
   }
   ui.push_back(ns);
   unq_ltdof.SetSize((int)ud.size()); unq_shr_i.SetSize((int)ui.size());
   for (size_t k = 0; k < ud.size(); k++) { unq_ltdof[(int)k] = ud[k]; }
   for (size_t k = 0; k < ui.size(); k++) { unq_shr_i[(int)k] = ui[k]; }

   if (exchange) { mfem_amd_exchange_destroy(exchange); exchange = nullptr; }
   if (mfem_amd_exchange_create(Mpi::World(), (int)up_ranks.size(), up_ranks.data(), up_sizes.data(), (int)lo_ranks.size(),
                                lo_ranks.data(), lo_sizes.data(), ns ? shr_ltdof.Read() : nullptr,
                                ext_ldof.Size() ? ext_ldof.Read() : nullptr, unq_ltdof.Size(),
                                unq_ltdof.Size() ? unq_ltdof.Read() : nullptr, unq_shr_i.Read(),
                                ns ? unq_shr_j.Read() : nullptr, &exchange, &cb) != 0)
   { MFEM_ABORT("mfem_amd: cannot create the shared-dof exchange: " << mfem_amd_comm_last_error()); }
}

void ParFiniteElementSpace::GetEssentialTrueDofs(const Array<int> &bdr_attr_is_ess, Array<int> &ess_tdof_list, int component) const
{
   // the boundary elements of the rank's own elements (the local mesh has them), and -- a local dof on the domain boundary may
   // lie on boundary faces of other ranks' elements only -- the serial mesh's boundary elements that touch a vertex of the
   // rank's part, numbered as the serial space numbers them and looked up among the rank's interface dofs; ascending vdofs
   const Mesh &sm = pmesh->SerialMesh();
   const H1_FECollection *h1c = dynamic_cast<const H1_FECollection*>(FEColl());
   const int vd = GetVDim(), ne1 = h1c->GetOrder() - 1, nbd = 4 + 4*ne1 + ne1*ne1;
   Array<int> marker;
   FiniteElementSpace::GetEssentialVDofs(bdr_attr_is_ess, marker, component);
   std::vector<int> list;
   for (int i = 0; i < marker.Size(); i++) { if (marker[i]) { list.push_back(i); } }
   std::vector<char> mine((size_t)sm.GetNV(), 0);
   for (int e : pmesh->LocalElements()) { for (int k = 0; k < 8; k++) { mine[(size_t)sm.GetElementVertices(e)[k]] = 1; } }
   std::vector<int64_t> bd((size_t)nbd);
   for (int b = 0; b < sm.GetNBE(); b++)
   {
      const int attr = sm.GetBdrAttribute(b);
      if (attr < 1 || attr > bdr_attr_is_ess.Size() || !bdr_attr_is_ess[attr - 1]) { continue; }
      const int *v = sm.GetBdrElementVertices(b);
      const int nmine = mine[(size_t)v[0]] + mine[(size_t)v[1]] + mine[(size_t)v[2]] + mine[(size_t)v[3]];
      if (nmine == 0) { continue; }
      const int f = sm.GetBdrElementFaceIndex(b), e1 = sm.GetFaceElement1(f);
      if (e1 >= 0 && pmesh->Partitioning()[(size_t)e1] == pmesh->GetMyRank()) { continue; }     // the local mesh has it
      FiniteElementSpace::BdrElementDofsOf(sm, *h1c, b, bd.data());
      for (int i = 0; i < nbd; i++)
      {
         const int li = LocalDof(bd[(size_t)i]);
         if (li < 0) { continue; }
         if (component < 0) { for (int c = 0; c < vd; c++) { list.push_back(DofToVDof(li, c)); } }
         else { list.push_back(DofToVDof(li, component)); }
      }
   }
   std::sort(list.begin(), list.end());
   list.erase(std::unique(list.begin(), list.end()), list.end());
   ess_tdof_list.SetSize((int)list.size());
   for (size_t k = 0; k < list.size(); k++) { ess_tdof_list[(int)k] = list[k]; }
}

void ParFiniteElementSpace::Forward(Vector &v) const
{
   MFEM_VERIFY(v.Size() == GetVSize(), "ParFiniteElementSpace::Forward: wrong vector size");
   if (cb.forward(cb.ctx, v.ReadWrite(), nullptr) != 0) { MFEM_ABORT("mfem_amd: shared-dof exchange (P) failed: " << mfem_amd_comm_last_error()); }
}
void ParFiniteElementSpace::Reverse(Vector &v) const
{
   MFEM_VERIFY(v.Size() == GetVSize(), "ParFiniteElementSpace::Reverse: wrong vector size");
   if (cb.reverse(cb.ctx, v.ReadWrite(), nullptr) != 0) { MFEM_ABORT("mfem_amd: shared-dof exchange (P^T) failed: " << mfem_amd_comm_last_error()); }
}

// the forward exchange runs beside the interior elements (mfem_amd_pa_operator::e_split): whole batches of the blocked layout
static int64_t interior_split(const mfem_amd_pa_operator &op, int n_interior)
{
   if (op.vdim > 0 || op.kind == 2) { return 0; }
   const int64_t per = (op.kind ? 6 : 1)*(int64_t)op.Q1D*op.Q1D*op.Q1D;
   const int64_t nbz = op.pa_layout ? mfem_amd_pa_data_size(op.kind, 1, op.Q1D, 1)/per : 1;
   return (n_interior/nbz)*nbz;
}

// ---- ParPAOperator --------------------------------------------------------------------------------
ParPAOperator::ParPAOperator(const PABilinearFormExtension &ext, const ParFiniteElementSpace &pf, const Array<int> &ess_list)
   : Operator(pf.GetVSize()), pfes(&pf), pa(&ext)
{
   MFEM_VERIFY(ext.FillOperator(cop), "mfem_amd: the parallel operator needs a form with one assembled domain integrator");
   ess_list.Copy(ess);
   marker.SetSize(height); marker = (unsigned char)0;
   unsigned char *m = marker.HostWrite();
   for (int i = 0; i < ess.Size(); i++) { m[ess[i]] = 1; }
   const Array<int> &gh = pf.GhostDofs();
   for (int i = 0; i < gh.Size(); i++) { if (m[gh[i]]) { m[gh[i]] = 2; } }       // essential ghost copy: y = 0
   const int ness = ess.Size();
   if (ness && cop.vdim == 0)                  // (vector spaces apply the vdof marker in the gather instead)
   {
      const ElementRestriction *R = ext.GetElementRestriction();
      masked_gather.SetSize(R->GatherMap().Size());
      MFEM_AMD_CALL(mfem_amd_mask_gather_map(R->GatherMap().Size(), R->GatherMap().Read(), marker.Read(), masked_gather.Write(), nullptr));
      cop.masked_gather_map = masked_gather.Read();
   }
   cop.ness = ness; cop.ess = ness ? ess.Read() : nullptr; cop.ess_marker = marker.Read();
   SetSplit();
}
// ... and the ghost -> owner exchange beside the transpose-gather (mfem_amd_pa_operator::reverse_overlap): the rank's ghost dofs
static void ghost_rows(mfem_amd_pa_operator &op, const ParFiniteElementSpace &pf)
{
   if (op.vdim > 0 || op.kind == 2) { return; }
   const Array<int> &gh = pf.GhostDofs();
   op.reverse_overlap = 1;
   op.n_ghost_rows = gh.Size();
   op.ghost_rows = gh.Size() ? gh.Read() : nullptr;
}
void ParPAOperator::SetSplit() const
{
   cop.e_split = interior_split(cop, pfes->GetParMesh()->InteriorElements());
   ghost_rows(cop, *pfes);
}
// as ConstrainedOperator::RefreshOperator (fem.cpp): the form's buffers may have moved since construction
void ParPAOperator::RefreshOperator() const
{
   const int64_t ness = cop.ness;
   MFEM_VERIFY(pa->FillOperator(cop), "mfem_amd: the form behind the parallel operator lost its partial assembly");
   cop.ness = ness; cop.ess = ness ? ess.Read() : nullptr; cop.ess_marker = marker.Read();
   if (ness && cop.vdim == 0) { cop.masked_gather_map = masked_gather.Read(); }
   SetSplit();
}
void ParPAOperator::Mult(const Vector &x, Vector &y) const
{
   MFEM_VERIFY(x.Size() == Width() && y.Size() == Height(), "incompatible sizes");
   RefreshOperator();
   MFEM_AMD_CALL(mfem_amd_pa_par_mult(&cop, pfes->Comm(), const_cast<Vector&>(x).ReadWrite(), y.Write(), nullptr));
}

// ---- ParBilinearForm ------------------------------------------------------------------------------
void ParBilinearForm::Mult(const Vector &x, Vector &y) const
{
   Vector xx(x);
   const PABilinearFormExtension *ext = GetPAExtension();
   mfem_amd_pa_operator op;
   if (ext && ext->FillOperator(op) && op.vdim == 0 && xx.Read() != y.Read())
   {
      // the unconstrained P^T A P in one library call: the owner -> ghost exchange beside the interior elements
      op.e_split = interior_split(op, pfes->GetParMesh()->InteriorElements());
      ghost_rows(op, *pfes);
      y.SetSize(xx.Size());
      MFEM_AMD_CALL(mfem_amd_pa_par_mult(&op, pfes->Comm(), xx.ReadWrite(), y.Write(), nullptr));
      return;
   }
   pfes->Forward(xx);
   BilinearForm::Mult(xx, y);
   pfes->Reverse(y);
}
void ParBilinearForm::AssembleDiagonal(Vector &diag) const
{
   BilinearForm::AssembleDiagonal(diag);
   pfes->Reverse(diag);
   const Array<int> &gh = pfes->GhostDofs();
   if (gh.Size())
   {
      MFEM_AMD_CALL(mfem_amd_set_subvector(gh.Size(), gh.Read(), nullptr, std::numeric_limits<double>::infinity(),
                                           diag.ReadWrite(), nullptr));
   }
}
// fem/pbilinearform.cpp FormLinearSystem -> Operator::FormLinearSystem (linalg/operator.cpp:114-129) with P, P^T
void ParBilinearForm::FormLinearSystem(const Array<int> &ess_tdof_list, Vector &x, Vector &b, OperatorHandle &A,
                                       Vector &X, Vector &B, int copy_interior)
{
   const PABilinearFormExtension *ext = GetPAExtension();
   MFEM_VERIFY(ext, "mfem_amd: FormLinearSystem needs AssemblyLevel::PARTIAL");
   const Array<int> &gh = pfes->GhostDofs();
   B = b;
   pfes->Reverse(B);                                        // B = P^T b
   X = x;
   if (!copy_interior) { X.SetSubVectorComplement(ess_tdof_list, 0.0); }
   if (gh.Size()) { X.SetSubVector(gh, 0.0); }              // true-dof vectors hold zero in the ghost entries
   ParPAOperator *op = new ParPAOperator(*ext, *pfes, ess_tdof_list);
   // EliminateRHS (linalg/operator.cpp:559-584): B -= A w with w = X on the essential dofs, then B = X there
   const int ness = ess_tdof_list.Size();
   if (ness)
   {
      Vector w(X.Size()), z(X.Size());
      w = 0.0;
      MFEM_AMD_CALL(mfem_amd_set_subvector(ness, ess_tdof_list.Read(), X.Read(), 0.0, w.ReadWrite(), nullptr));
      Mult(w, z);                                           // the unconstrained P^T A P
      B -= z;
      MFEM_AMD_CALL(mfem_amd_set_subvector(ness, ess_tdof_list.Read(), X.Read(), 0.0, B.ReadWrite(), nullptr));
   }
   A.Reset(op);
}
void ParBilinearForm::RecoverFEMSolution(const Vector &X, const Vector &, Vector &x)
{
   x = X;
   pfes->Forward(x);
}

} // namespace mfem

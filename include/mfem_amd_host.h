/* mfem_amd_host.h -- C entry points of the C++ host layer (libmfem_amd_cxx.so) for drivers that are not C++.
 *
 * The C++ host layer (mfem_amd/host/mfem.hpp) builds what the reference's callers build on the host: the mesh in
 * MFEM's element order, the H1 space in MFEM's dof numbering, the ElementRestriction tables.  A driver written in
 * another language (bench.py) gets the same tables through this call, so that what it times is the numbering the
 * reference's own drivers (examples/ex1.cpp, tests/benchmarks/bench_ceed.cpp) produce -- not a synthetic one.
 * Host-only: nothing here touches the GPU. */
#ifndef MFEM_AMD_HOST_H
#define MFEM_AMD_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfem_amd_host_tables
{
   int64_t ne, ndofs, ness;
   int D1D;
   int *gather_map;   /* ne*D1D^3: ElementRestriction::GatherMap(), lexicographic local order (fem/restriction.cpp:86-99) */
   double *nodes_E;   /* ne*3*8: the hexahedra's corners as the E-vector of the trilinear nodal space, (2,2,2,3,ne), x fastest */
   int *ess;          /* ness: GetEssentialTrueDofs with every boundary attribute marked (bench_ceed.cpp:99-100) */
   void *owner;       /* internal */
} mfem_amd_host_tables;

/* Mesh::MakeCartesian3D(nx, ny, nz, HEXAHEDRON, sx, sy, sz, sfc_ordering) (mesh/mesh.cpp:3684-3964: sfc = 1 is the
 * reference's default, the generalised Hilbert curve of mesh/ncmesh.cpp:5435-5634), H1_FECollection(p, 3),
 * FiniteElementSpace (fem/fespace.cpp:3392-3500).  The box starts at (ox, oy, oz).  Returns 0, or 1 with a message on stderr. */
int mfem_amd_host_cartesian_tables(int p, int nx, int ny, int nz, double sx, double sy, double sz,
                                   double ox, double oy, double oz, int sfc, mfem_amd_host_tables *out);
void mfem_amd_host_tables_free(mfem_amd_host_tables *t);

#ifdef __cplusplus
}
#endif
#endif

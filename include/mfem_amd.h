/* mfem_amd.h -- C ABI of the MI355X-native partial-assembly (AssemblyLevel::PARTIAL)
 * operator-apply + Jacobi-PCG path of MFEM (H1, hexahedra, scalar fields).
 *
 * This is the drop-in boundary: every entry point replaces one reference function (cited
 * as file:line relative to the MFEM source root) and is what an MFEM-side binding would
 * call from the corresponding C++ virtual / dispatch-table slot.
 *
 * Conventions
 *  - Unless marked HOST, pointers are device pointers (hipMalloc'ed, or
 *    Vector::Read()/Write()/ReadWrite() pointers of an MFEM build with the HIP backend).
 *  - B, G (the tiny 1D DofToQuad tables) are HOST pointers (Array::HostRead()): they are
 *    passed to the kernels by value (kernel-argument segment -> scalar registers).
 *    Layout is MFEM's DofToQuad::TENSOR layout: B[q + Q1D*d], G[q + Q1D*d]
 *    (fem/fe/fe_base.cpp:2622-2673).  W (quadrature weights, Q1D^3 doubles,
 *    W[qx + Q1D*(qy + Q1D*qz)], fem/intrules.cpp:56-84) is a DEVICE pointer.
 *  - Sizes that can exceed 2^31 are int64_t (the reference uses 32-bit int and cannot hold
 *    BP3 at 1e8 dofs, linalg/dtensor.hpp:32).  Index arrays are int32 like the reference's.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream, as the reference
 *    launches, general/forall.hpp:689-718).  All entry points are asynchronous on `stream`
 *    and never allocate, free or synchronise (safe under HIP graph capture), except where
 *    noted.
 *  - Return value: 0 on success, non-zero on error; mfem_amd_last_error() gives the
 *    message.  The C++ layer turns non-zero into abort-with-message, MFEM's convention
 *    (general/error.cpp, MFEM_VERIFY / MFEM_GPU_CHECK general/hip.hpp:32-39).
 *  - Limits: 2 <= D1D <= 10, D1D*Q1D <= 140, Q1D <= 14 (MFEM's HIP limits are
 *    MAX_D1D = MAX_Q1D = 10, general/forall.hpp:58-61).  Violations return an error, like
 *    MFEM_VERIFY(D1D <= MAX_D1D) (fem/integ/bilininteg_diffusion_kernels.hpp:1005-1006).
 */
#ifndef MFEM_AMD_H
#define MFEM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime -------------------------------------------------------------------- */

/* Select the HIP device for the calling thread and warm the library up.
 * Counterpart: Device::Configure("hip") (general/device.hpp:224). */
int mfem_amd_init(int device);
/* Last error message of the calling thread (never NULL). */
const char *mfem_amd_last_error(void);
/* Non-zero: report on stderr the first use of a non-specialised (fallback) kernel, like
 * MFEM_REPORT_KERNELS (fem/kernel_reporter.hpp:54-98). Also enabled by that env var. */
void mfem_amd_report_kernels(int on);
/* 1 if (D1D,Q1D) has a compile-time specialised apply kernel, else 0 (the dispatch-table
 * query, fem/kernel_dispatch.hpp:166-181).  kind: 0 = mass, 1 = diffusion.  Registered: D1D = 2..9 with
 * Q1D = D1D or D1D + 1, plus the reference's over-integrated "others" (bilininteg_diffusion_kernels.cpp:62-63,
 * bilininteg_mass_kernels.cpp:59,66-68): (4,6), (5,8) for both, (2,4), (3,6) for mass.  MFEM_AMD_TRACE=1 in
 * the environment wraps the stages of the path in roctx ranges (rocprofv3 --marker-trace). */
int mfem_amd_has_specialization(int kind, int D1D, int Q1D);

/* ---- ElementRestriction (fem/restriction.cpp) ------------------------------------ */

/* ElementRestriction::Mult, scalar field (fem/restriction.cpp:109-129):
 * yE[i] = +-x[gather_map[i]], i < nedofs = NE*D1D^3 (negative entries encode -1-idx and a
 * sign flip). */
int mfem_amd_restrict_gather(int64_t nedofs, const int *gather_map, const double *x,
                             double *yE, void *stream);
/* Same, vector field with vd components (mesh nodes): x ordered byNODES (byvdim = 0) or
 * byVDIM (1); yE laid out (nd, vd, NE) (fem/restriction.cpp:115-127). */
int mfem_amd_restrict_gather_vdim(int64_t nedofs, int nd, int vd, int byvdim,
                                  int64_t ndofs, const int *gather_map, const double *x,
                                  double *yE, void *stream);
/* ElementRestriction::TAddMultTranspose<ADD> (add = 0/1) and AbsMultTranspose (absv = 1)
 * (fem/restriction.cpp:152-180, 196-221): the deterministic transpose-offset gather
 * y[i] (+)= sum_{j in [offsets[i], offsets[i+1])} +-xE[indices[j]], summed left to right
 * starting from 0, exactly the reference's order. */
int mfem_amd_restrict_scatter(int64_t ndofs, const int *offsets, const int *indices,
                              const double *xE, double *y, int add, int absv,
                              void *stream);

/* Chunk-pipelined form of the same transpose gather (fem/restriction.cpp:152-180 applied chunk of
 * elements by chunk): a sub-row is the part of an L-row whose entries lie in one element chunk;
 * rows[i] = L-row | (1<<30: start from y[row], the partial sum the row's previous sub-row left, instead
 * of 0) | (1<<31: more sub-rows follow).  Entries [offsets[i], offsets[i+1]) of `indices` address xE.
 * Sub-rows of a row must be submitted in element order, in launches ordered on `stream`: the additions
 * are then the reference's, in the reference's order (bit-identical to mfem_amd_restrict_scatter). */
int mfem_amd_restrict_scatter_subrows(int64_t nsub, const int *rows, const int *offsets,
                                      const int *indices, const double *xE, double *y,
                                      void *stream);

/* ---- geometric factors and PA setup --------------------------------------------- */

/* GeometricFactors::Compute, JACOBIANS (mesh/mesh.cpp:15178-15231; Derivatives3D,
 * fem/qinterp/grad.hpp:232-374): nodesE (Dn,Dn,Dn,3,NE) -> J (Q,Q,Q,3,3,NE),
 * J(q,r,c,e) = dx_r/dxi_c.  Bn, Gn: HOST, nodal basis at the rule's points. */
int mfem_amd_geom_jacobians_3d(int Dn, int Q1D, int64_t ne, const double *Bn,
                               const double *Gn, const double *nodesE, double *J,
                               void *stream);
/* GeometricFactors DETERMINANTS (fem/qinterp/det.hpp:209-282): detJ (Q,Q,Q,NE) from J. */
int mfem_amd_geom_det_3d(int Q1D, int64_t ne, const double *J, double *detJ, void *stream);
/* PADiffusionSetup3D (fem/integ/bilininteg_diffusion_kernels.cpp:243-367).  Scalar or
 * vector coefficient (coeffDim 1 or 3) and symmetric matrix coefficient (coeffDim 6, entries
 * (1,1),(1,2),(1,3),(2,2),(2,3),(3,3); :297-348): D (Q,Q,Q,6,NE), component order
 * (1,1),(2,1),(3,1),(2,2),(3,2),(3,3).  Full matrix coefficient (coeffDim 9, the layout
 * CoefficientVector::ProjectTranspose leaves, fem/coefficient.cpp:2026-2056): D (Q,Q,Q,9,NE),
 * row-major planes, `symmetric` = 0 in the apply / diagonal entry points below.
 * C: device; const_c != 0 -> C has coeffDim entries, else (coeffDim,Q,Q,Q,NE). */
int mfem_amd_diffusion_setup_3d(int Q1D, int coeffDim, int64_t ne, const double *W,
                                const double *J, const double *C, int const_c, double *D,
                                void *stream);
/* Mass setup (fem/integ/bilininteg_mass_pa.cpp:72-77): D(q,e) = W(q)*c*detJ(q,e). */
int mfem_amd_mass_setup_3d(int Q1D, int64_t ne, const double *W, const double *detJ,
                           const double *C, int const_c, double *D, void *stream);
/* Fused nodes -> D: AssemblePA without materialising J (9*Q^3*NE doubles, 1.5x pa_data;
 * SURVEY a-6).  kind 0 = mass (D (Q^3,NE)), 1 = diffusion (D (Q^3,6,NE), or
 * (Q^3,9,NE) for coeffDim 9, pa_layout 0 only); coeffDim 1, 3, 6 or 9 (diffusion), 1 (mass).  detJ_out may be NULL; if not, (Q^3,NE) determinants are also
 * written (needed by the RHS).  Same arithmetic per point as the two-step path. */
int mfem_amd_pa_setup_from_nodes_3d(int kind, int Dn, int Q1D, int coeffDim, int64_t ne,
                                    const double *Bn, const double *Gn, const double *W,
                                    const double *nodesE, const double *C, int const_c,
                                    int pa_layout, double *D, double *detJ_out, void *stream);

/* pa_data layouts.  0: the reference's (q, component, element) (fem/integ/
 * bilininteg_diffusion_kernels.cpp:257, bilininteg_mass_pa.cpp:72-77) -- what the
 * dispatch-table level entry points below take, so MFEM's own pa_data can be passed.
 * 1: blocked for HBM3E, the same doubles permuted so that every load instruction of a
 * workgroup of the fused kernel reads one contiguous 16-byte-per-lane run (see
 * mfem_amd/csrc/common.h).  pa_data is a private member of the integrator
 * (fem/bilininteg.hpp:2213), so its layout is free when both AssemblePA and AddMultPA run
 * here.  No counterpart in the reference. */
int64_t mfem_amd_pa_data_size(int kind, int pa_layout, int Q1D, int64_t ne); /* in doubles */
int mfem_amd_pa_data_to_blocked(int kind, int Q1D, int64_t ne, const double *D,
                                double *Dblocked, void *stream);
int mfem_amd_pa_data_from_blocked(int kind, int Q1D, int64_t ne, const double *Dblocked,
                                  double *D, void *stream);

/* DomainLFIntegrator, device path (fem/integ/lininteg_domain.cpp:22-59 -> DLFEvalAssemble3D,
 * fem/integ/lininteg_domain_kernels.hpp): yE(dx,dy,dz,e) += sum_q B(qx,dx) B(qy,dy) B(qz,dz)
 * F(q,e) with F = W * f * detJ at the quadrature points ((Q^3, NE), what
 * mfem_amd_pa_setup_from_nodes_3d(kind 0, pa_layout 0) produces for the coefficient f).
 * Follow with mfem_amd_restrict_scatter.  B: HOST. */
int mfem_amd_domain_lf_3d(int D1D, int Q1D, int64_t ne, const double *B, const double *F,
                          double *yE, void *stream);

/* ---- PA operator action ---------------------------------------------------------- */

/* DiffusionIntegrator::AddMultPA -> ApplyPAKernels::Run -> SmemPADiffusionApply3D<D,Q> /
 * fallback PADiffusionApply3D (fem/integ/bilininteg_diffusion_pa.cpp:40-74,
 * fem/integ/bilininteg_diffusion_kernels.hpp:988-1211, 795-985).  Signature mirrors
 * DiffusionIntegrator::ApplyKernelType (fem/bilininteg.hpp:2181-2185):
 * yE += G^T D G xE per element.  symmetric != 0: 6-plane D, the specialised kernels;
 * symmetric == 0: the 9-plane D of a full matrix coefficient (bilininteg_diffusion_kernels.hpp:
 * 1119-1127), which runs through the runtime-(D,Q) LDS kernel for every order. */
int mfem_amd_pa_diffusion_apply_3d(int D1D, int Q1D, int64_t ne, int symmetric,
                                   const double *B, const double *G, const double *D,
                                   const double *xE, double *yE, void *stream);
/* MassIntegrator::AddMultPA -> SmemPAMassApply3D<D,Q> / PAMassApply3D_Element
 * (fem/integ/bilininteg_mass_pa.cpp:140-169, bilininteg_mass_kernels.hpp:1119-1144,
 * 656-805); MassIntegrator::ApplyKernelType (fem/bilininteg.hpp:2387-2393). yE += ... */
int mfem_amd_pa_mass_apply_3d(int D1D, int Q1D, int64_t ne, const double *B,
                              const double *D, const double *xE, double *yE, void *stream);
/* AssembleDiagonalPA: SmemPADiffusionDiagonal3D / SmemPAMassAssembleDiagonal3D
 * (fem/integ/bilininteg_diffusion_kernels.hpp:368-484, bilininteg_mass_kernels.hpp:324-408;
 * DiagonalKernelType fem/bilininteg.hpp:2187-2189).  diagE += diag(A_e).  symmetric == 0:
 * 9-plane D, k = 3 i + j (:428-431), pa_layout 0 only. */
int mfem_amd_pa_diffusion_diag_3d(int D1D, int Q1D, int64_t ne, int symmetric,
                                  const double *B, const double *G, const double *D,
                                  int pa_layout, double *diagE, void *stream);
int mfem_amd_pa_mass_diag_3d(int D1D, int Q1D, int64_t ne, const double *B,
                             const double *D, int pa_layout, double *diagE, void *stream);

/* PABilinearFormExtension::MultInternal, H1 branch, one integrator
 * (fem/bilinearform_ext.cpp:487-564): yE = A_E (gather x) with the L->E gather fused into
 * the element kernel (no localX round trip, no memset of localY).  kind 0 mass / 1
 * diffusion / 2 diffusion with the non-symmetric 9-plane D (pa_layout 0 only; the operator
 * struct below takes the same kinds).  gather_map entries < 0 here mean "use 0.0" (masked essential dofs, see
 * mfem_amd_mask_gather_map) -- scalar H1 gather maps are never negative in the reference.
 * Follow with mfem_amd_restrict_scatter[_dot] for the E->L sum. */
int mfem_amd_pa_apply_L2E(int kind, int pa_layout, int D1D, int Q1D, int64_t ne,
                          const double *B, const double *G, const double *D,
                          const int *gather_map, const double *x, double *yE, void *stream);

/* Element-attribute markers of a domain integrator (BilinearForm::AddDomainIntegrator(bfi, marker)):
 * PABilinearFormExtension::AddMultWithMarkers -> AddWithMarkers_ (fem/bilinearform_ext.cpp:807-833,
 * 753-774): yE(:,e) += xE(:,e) for the elements with markers[attributes[e]-1] != 0, xE being the
 * integrator's action on a zeroed temporary E-vector.  markers (nmarkers ints), attributes (ne ints):
 * device. */
int mfem_amd_evec_add_marked(int64_t ne, int nd, int nmarkers, const int *markers,
                             const int *attributes, const double *xE, double *yE, void *stream);
/* The same selection for the diagonal (fem/bilinearform_ext.cpp:374-398): dE(:,e) = 0 for the
 * elements that are NOT marked. */
int mfem_amd_evec_zero_unmarked(int64_t ne, int nd, int nmarkers, const int *markers,
                                const int *attributes, double *dE, void *stream);

/* ---- ConstrainedOperator (linalg/operator.cpp:511-646) --------------------------- */

/* marker[i] = 1 for i in list (constraint_list -> byte marker), others untouched. */
int mfem_amd_set_marker(int64_t nlist, const int *list, unsigned char *marker, void *stream);
/* masked[i] = marker[gather_map[i]] ? -1 : gather_map[i]: folds `z = x; z[ess] = 0`
 * (linalg/operator.cpp:603-608) into the fused gather. */
int mfem_amd_mask_gather_map(int64_t nedofs, const int *gather_map,
                             const unsigned char *marker, int *masked, void *stream);
/* y[list[i]] = x[list[i]] (linalg/operator.cpp:624-629), or y[list[i]] = value if x NULL
 * (:608). */
int mfem_amd_set_subvector(int64_t nlist, const int *list, const double *x, double value,
                           double *y, void *stream);

/* ---- Vector / reductions (linalg/vector.cpp, general/reducers.hpp) ---------------- */

/* z = a*x + b*y (add(v1,alpha,v2,v) linalg/vector.cpp:436-473, subtract :570; aliasing of
 * z with x or y allowed).  y may be NULL when b == 0 (copy / scale). */
int mfem_amd_vec_axpby(int64_t n, double a, const double *x, double b, const double *y,
                       double *z, void *stream);
/* x[i] = value (Vector::operator=(double) linalg/vector.cpp:232). */
int mfem_amd_vec_fill(int64_t n, double value, double *x, void *stream);
/* Bytes of device workspace the reductions below need (independent of n). */
int64_t mfem_amd_reduce_workspace_bytes(void);
/* *result = sum x[i]*y[i] (Vector::operator* linalg/vector.cpp:1079-1152 -> reduce
 * general/reducers.hpp:531-593).  Deterministic: fixed chunking that depends on n only,
 * wavefront shuffle + LDS tree, final pass by one workgroup; `result` is a device double.
 * ws: device workspace of mfem_amd_reduce_workspace_bytes(). */
int mfem_amd_vec_dot(int64_t n, const double *x, const double *y, double *result,
                     void *ws, void *stream);
/* mfem_amd_vec_dot for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): the same two kernels and summation order, hence the same
 * bits, with the partial sums kept in a 64 KiB device array that the library owns (one per GPU, held until
 * the process ends) instead of a caller workspace.  No counterpart in the reference.  Calls no hipMalloc /
 * hipFree.  Limits: n <= MFEM_AMD_DOT_NOALLOC_MAX_N (larger n is refused with an error); devices 0..63.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams) race --
 * order them on one stream.  The array keeps the last call's partial sums; every entry a call reads it has
 * written first. */
#define MFEM_AMD_DOT_NOALLOC_MAX_N ((int64_t)1 << 33)
int mfem_amd_vec_dot_noalloc(int64_t n, const double *x, const double *y, double *result,
                             void *stream);
/* OperatorJacobiSmoother::Setup (linalg/solvers.cpp:401-425):
 * dinv[i] = damping/diag[i]; dinv[ess[i]] = damping. */
int mfem_amd_jacobi_setup(int64_t n, const double *diag, double damping, int64_t ness,
                          const int *ess, double *dinv, void *stream);
/* OperatorJacobiSmoother::Mult (linalg/solvers.cpp:427-453): z = dinv .* r, one pass. */
int mfem_amd_jacobi_apply(int64_t n, const double *dinv, const double *r, double *z,
                          void *stream);

/* ---- Jacobi-PCG on the PA operator, scalars resident on the device ---------------- */

/* ---- vector H1 spaces (vdim components of one scalar space) ------------------------------ */

/* ElementRestriction::MultTranspose / AddMultTranspose / AbsMultTranspose for vdim > 1
 * (fem/restriction.cpp:152-180,196-221): y[vdof(i,c)] (+)= sum_j xE[(idx_j % nd) +
 * nd*(c + vdim*(idx_j / nd))], same order as the scalar rows.  nd = D1D^3.  byvdim as
 * Ordering::byVDIM (fem/fespace.hpp). */
int mfem_amd_restrict_scatter_vdim(int64_t ndofs, int nd, int vdim, int byvdim, const int *offsets,
                                   const int *indices, const double *xE, double *y, int add,
                                   int absv, void *stream);

/* Fused ElementRestriction::Mult + VectorDiffusionIntegrator / VectorMassIntegrator::AddMultPA
 * (fem/integ/bilininteg_vecdiffusion_pa.hpp:108-170, bilininteg_vecmass_pa.hpp; kind as
 * mfem_amd_pa_apply_L2E): yE(D,D,D,vdim,NE) = A_c x_c for every component c, x an L-vector of
 * vdim*ndofs entries.  D: the SCALAR integrator's pa_data (d_comp_stride = 0: scalar coefficient,
 * read once for all components -- the reference stores and reads vdim identical copies), or one
 * array per component d_comp_stride doubles apart (vector coefficient; pa_layout 0 then means
 * the reference's (Q^3,NC,vdim,NE)).  ess_marker: optional vdim*ndofs bytes, nonzero entries of
 * x read as 0 (ConstrainedOperator z[ess] = 0, linalg/operator.cpp:604-612). */
int mfem_amd_pa_apply_L2E_vdim(int kind, int pa_layout, int D1D, int Q1D, int64_t ne, int vdim,
                               int byvdim, int64_t ndofs, const double *B, const double *G,
                               const double *D, int64_t d_comp_stride, const int *gather_map,
                               const unsigned char *ess_marker, const double *x, double *yE,
                               void *stream);

/* VectorDiffusionIntegrator::AddMultPA at the dispatch-table seam
 * (ApplyKernelType, fem/bilininteg.hpp; SmemPAVectorDiffusionApply3D,
 * fem/integ/bilininteg_vecdiffusion_pa.hpp:108-170): yE += A xE on (D,D,D,vdim,NE) E-vectors;
 * D is the reference's pa_data (Q,Q,Q,9,vdim,NE): dim*dim planes allocated per component, the
 * symmetric six filled and read (bilininteg_vecdiffusion_pa.cpp:133-134,228-290).
 * coeff_vdim 1 or vdim; matrix coefficients (vdim^2) are rejected.  Extension: coeff_vdim = 0
 * takes the SCALAR integrator's (Q,Q,Q,6,NE) data, shared by all components. */
int mfem_amd_pa_vector_diffusion_apply_3d(int D1D, int Q1D, int64_t ne, int vdim, int coeff_vdim,
                                          const double *B, const double *G, const double *D,
                                          const double *xE, double *yE, void *stream);
/* VectorMassIntegrator::AddMultPA (fem/integ/bilininteg_vecmass_pa.hpp): D is (Q,Q,Q,coeff_vdim,NE)
 * (bilininteg_vecmass_pa.cpp:118-160). */
int mfem_amd_pa_vector_mass_apply_3d(int D1D, int Q1D, int64_t ne, int vdim, int coeff_vdim,
                                     const double *B, const double *D, const double *xE,
                                     double *yE, void *stream);

/* y[vdof(i,c)] = scale[c]*x[i], i < ndofs, c < vdim (scale: HOST, vdim doubles, NULL = ones):
 * spreads a scalar-space vector over the components of a vector space -- the diagonal of a
 * vector integrator with a scalar coefficient (PAVectorDiffusionDiagonal3D,
 * fem/integ/bilininteg_vecdiffusion_pa.cpp) and VectorDomainLFIntegrator with a constant vector
 * coefficient (fem/lininteg.cpp) are the scalar results per component. */
int mfem_amd_vec_replicate_vdim(int64_t ndofs, int vdim, int byvdim, const double *scale,
                                const double *x, double *y, void *stream);

/* Problem description for mfem_amd_pcg_pa (all device pointers unless marked HOST). */
typedef struct mfem_amd_pa_operator
{
   int kind;                 /* 0 mass, 1 diffusion */
   int pa_layout;            /* 0 MFEM layout, 1 blocked (mfem_amd_pa_data_to_blocked) */
   int D1D, Q1D;
   int64_t ne, ndofs;
   const double *B, *G;      /* HOST */
   const double *pa_data;
   const int *gather_map;    /* plain gather map (used when ness == 0) */
   const int *masked_gather_map; /* from mfem_amd_mask_gather_map (used when ness > 0) */
   const int *offsets, *indices;
   int64_t ness;
   const int *ess;           /* constraint_list */
   const unsigned char *ess_marker; /* ndofs bytes: 1 essential, 2 essential ghost copy (multi-GPU) */
   double *yE;               /* work: NE*D1D^3 doubles (times vdim for a vector space) */
   /* Vector H1 spaces (VectorMassIntegrator / VectorDiffusionIntegrator; BP2/BP4/BP6).  Zero =
    * scalar space, the fields above mean what they always did.  With vdim >= 1 set: vectors
    * have vdim*ndofs entries ordered byNODES (byvdim = 0) or byVDIM (1); gather_map / offsets /
    * indices stay the scalar space's tables; ess / ess_marker index vdofs (ess_marker: vdim*ndofs
    * bytes) and masked_gather_map is not used (the marker is applied in the gather). */
   int vdim, byvdim;
   int64_t pa_comp_stride;   /* 0: one pa_data shared by all components (scalar coefficient);
                                else doubles between the components' pa_data arrays */
   int64_t direct_rows;      /* 0, or what mfem_amd_interior_rows returned for gather_map / offsets: the last direct_rows
                                L-rows are element-interior dofs, which the element kernel writes itself (scalar spaces) */
   const int *packed_indices;/* `indices` remapped by mfem_amd_pack_indices: the table of the packed work E-vector that goes
                                with the direct rows (NULL: no direct rows, the plain two-kernel form) */
   int64_t e_split;          /* partitioned runs (scalar spaces): the local elements [0, e_split) touch no ghost dof -- their
                                part of the element kernel runs while the owner -> ghost exchange is in flight on a second
                                stream; a multiple of the blocked layout's batch (mfem_amd_pa_data_size(kind, 1, Q1D, 1) /
                                (C*Q1D^3) elements).  0: the exchange first, then all elements */
   int reverse_overlap;      /* partitioned runs (scalar spaces): 1 = ghost_rows / n_ghost_rows below are set and the ghost -> owner
                                exchange may run beside the transpose-gather; 0: it follows the transpose-gather */
   const int *ghost_rows;    /* the n_ghost_rows local L-dofs that other ranks own (the exchange's ext_ldof list; device; may be */
   int64_t n_ghost_rows;     /* empty).  With reverse_overlap and a communicator that offers reverse_begin / reverse_end, those
                                rows are summed first and travel to their owners on a second stream while the transpose-gather
                                sums all rows (the reference, fem/pfespace.cpp:5503-5539, has no such overlap) */
} mfem_amd_pa_operator;

/* Element-interior dofs belong to one element, so their rows of G^T are single entries: when they are the LAST
 * NE*(D1D-2)^3 L-dofs, numbered element by element in lexicographic order -- as the reference numbers every H1 space
 * (vertex, edge, face, then interior dofs: fem/fespace.cpp, FiniteElementSpace::BuildDofToArrays and the H1 dof maps of
 * fem/fe/fe_h1.cpp) -- the element kernel can store them straight into y and ElementRestriction::MultTranspose
 * (fem/restriction.cpp:152-180) only has to sum the rows before them: 22 % (p = 4) to 47 % (p = 8) of the E-vector
 * never goes through memory.  Same values, bit for bit.  This checks the property on the device (every interior entry
 * of gather_map and the row lengths in offsets) and returns NE*(D1D-2)^3 in *direct_rows (HOST) or 0.
 * work: one int of device memory.  Synchronises the stream. */
int mfem_amd_interior_rows(int D1D, int64_t ne, int64_t ndofs, const int *gather_map, const int *offsets,
                           int *work, int64_t *direct_rows, void *stream);

/* The packed form of the work E-vector between the two kernels of the apply (fem/bilinearform_ext.cpp:540-558: localY
 * written by AddMultPA, read by ElementRestriction::MultTranspose, fem/restriction.cpp:152-180).  With the element-interior
 * rows gone (mfem_amd_interior_rows) the reference's (D,D,D,NE) layout would be written with holes -- every sector of it
 * still travels -- and read a sector at a time by row blocks that want one class of entries (MFEM numbers vertex, edge,
 * face dofs class after class).  The packed form keeps each element's NS = D1D^3 - (D1D-2)^3 surviving entries contiguous,
 * class by class: 8 vertex entries, 12 edges (D1D-2 each), 6 faces ((D1D-2)^2 each).  mfem_amd_pack_indices remaps the nnz
 * entries of `indices` once (signs kept; element-interior entries, which nothing reads, map to the element's first
 * entry): rows are summed from the same values in the same order -- bit-identical y.  packed_indices: nnz ints (device). */
int mfem_amd_pack_indices(int D1D, int64_t nnz, const int *indices, int *packed_indices, void *stream);

/* The element kernel of mfem_amd_pa_apply_L2L alone, for callers that run the two halves of the apply themselves (timing,
 * overlap with an exchange): mfem_amd_pa_apply_L2E with the element-interior rows stored to
 * y_interior[e*(D1D-2)^3 + lexicographic interior index] (= y + ndofs - direct_rows for a space that passed
 * mfem_amd_interior_rows) and everything else to the packed E-vector yE_packed (NE*NS doubles), which
 * mfem_amd_restrict_scatter then reads over the first ndofs - direct_rows rows through packed_indices. */
int mfem_amd_pa_apply_L2E_packed(int kind, int pa_layout, int D1D, int Q1D, int64_t ne,
                                 const double *B, const double *G, const double *D,
                                 const int *gather_map, const double *x, double *yE_packed,
                                 double *y_interior, void *stream);

/* PABilinearFormExtension::Mult on L-vectors (fem/bilinearform_ext.cpp:487-564: ElementRestriction::Mult, the
 * integrator's AddMultPA, ElementRestriction::MultTranspose) as two kernels: y = G^T A_E G x.
 * direct_rows: 0 or the value from mfem_amd_interior_rows, then with packed_indices from mfem_amd_pack_indices (NULL
 * otherwise).  yE: NE*D1D^3 doubles of work space. */
int mfem_amd_pa_apply_L2L(int kind, int pa_layout, int D1D, int Q1D, int64_t ne, int64_t ndofs,
                          const double *B, const double *G, const double *D, const int *gather_map,
                          const int *offsets, const int *indices, const int *packed_indices,
                          int64_t direct_rows, const double *x, double *yE, double *y, void *stream);

/* ConstrainedOperator(PABilinearFormExtension)::Mult, DIAG_ONE
 * (linalg/operator.cpp:586-646 over fem/bilinearform_ext.cpp:487-564):
 * y = A z with z = x, z[ess] = 0; y[ess] = x[ess].  Two kernels (fused gather+apply,
 * transpose-gather with the essential rows patched in). */
int mfem_amd_pa_constrained_mult(const mfem_amd_pa_operator *op, const double *x, double *y,
                                 void *stream);

typedef struct mfem_amd_pcg_result
{
   int iterations;      /* CGSolver::GetNumIterations */
   int converged;       /* CGSolver::GetConverged */
   double initial_nom;  /* (B r, r) at iteration 0 */
   double final_nom;    /* (B r, r) at the last iteration; final_norm = sqrt */
   int history_count;   /* entries written to hist: iterations + 1, or iterations when the solve ended on a
                           zero / non-finite (A d, d) after the iteration count was advanced (solvers.cpp:1011-1025) */
} mfem_amd_pcg_result;

/* Bytes of device workspace mfem_amd_pcg_pa needs for n unknowns (3 vectors, reduction
 * partials, the residual history and the scalar state). */
int64_t mfem_amd_pcg_workspace_bytes(int64_t n, int max_iter);
/* CGSolver::Mult with iterative_mode = false (linalg/solvers.cpp:869-1050), preconditioned
 * by OperatorJacobiSmoother (dinv != NULL) or not (dinv == NULL, bench_ceed.cpp:117-121).
 * Stopping rule (:919,988): stop when (B r, r) <= max((B r, r)_0*rel_tol^2, abs_tol^2);
 * max_iter as the reference.  Vector updates are fused (kernel A
 * {r -= alpha z; z = dinv r; partial r.z}, kernel B {x += alpha d; d = z + beta d}, d.Ad in
 * the transpose-gather epilogue: 11 vector passes per iteration); alpha, beta, the residual norms and the stop flag live
 * on the device, so no host synchronisation happens inside the loop: the host polls the
 * flag every `check_every` iterations (>= 1) and kernels launched after convergence are
 * no-ops, so the iteration count is exactly the reference's.
 * hist: HOST, may be NULL, else max_iter+1 doubles = the "(B r, r)" sequence the reference
 * prints (linalg/solvers.cpp:898-901, 982-986).
 * This entry point synchronises `stream` (it returns the result to the host). */
int mfem_amd_pcg_pa(const mfem_amd_pa_operator *op, const double *dinv, const double *b,
                    double *x, double rel_tol, double abs_tol, int max_iter,
                    int check_every, void *workspace, mfem_amd_pcg_result *result,
                    double *hist, void *stream);

/* ---- multi-GPU: shared-dof exchange + global reductions --------------------------- */

/* DeviceConformingProlongationOperator's device kernels (fem/pfespace.cpp):
 * ExtractSubVector (:5347) buf[i] = x[idx[i]]; SetSubVector (:5371) y[idx[i]] = buf[i];
 * AddSubVector (:5475-5494) y[unq_ldof[i]] += sum_{j in [unq_shr_i[i], unq_shr_i[i+1])}
 * shr_buf[unq_shr_j[j]], in that fixed order (deterministic). */
int mfem_amd_shared_pack(int64_t n, const int *idx, const double *x, double *buf, void *stream);
int mfem_amd_shared_unpack(int64_t n, const int *idx, const double *buf, double *y, void *stream);
int mfem_amd_shared_accumulate(int64_t nuniq, const int *unq_ldof, const int *unq_shr_i,
                               const int *unq_shr_j, const double *shr_buf, double *y,
                               void *stream);

/* Communicator seen by the solver: three callbacks, all asynchronous on `stream` (a
 * hipStream_t).  The reference's counterpart is GroupCommunicator + MPI_Allreduce
 * (general/communication.cpp:724-1042, linalg/vector.hpp:773-779); here the transport is
 * RCCL over xGMI (torch.distributed, backend "nccl", in the Python host; see
 * mfem_amd/parallel.py) or anything else that honours the contract.  Vectors live on the
 * local L-dofs with non-owned (ghost) entries kept at 0 between operator applications.
 *   forward(v): P    -- ghost entries of v := the owner's value (pfespace.cpp:5401-5447)
 *   reverse(v): P^T  -- owner entries += ghost entries of the sharing ranks, in fixed
 *                       neighbour order; then the local ghost entries := 0 (:5503-5539)
 *   allreduce_sum(buf, n): in-place sum over ranks of n device doubles.
 *   reverse_begin(v) / reverse_end(v): optional (NULL: not offered), the two halves of reverse(v) for callers that
 *                       overlap them with other work: begin packs the ghost entries of v and moves them to their
 *                       owners' receive buffers (v is only read); end adds what arrived to the owner entries in the
 *                       same fixed order and zeroes the ghosts.  begin; end on one stream == reverse.  The two may be
 *                       handed DIFFERENT streams; the caller orders end after begin with an event.
 * Return 0 on success. */
typedef struct mfem_amd_comm
{
   void *ctx;
   int (*allreduce_sum)(void *ctx, double *dev_buf, int n, void *stream);
   int (*forward)(void *ctx, double *v, void *stream);
   int (*reverse)(void *ctx, double *v, void *stream);
   int (*reverse_begin)(void *ctx, const double *v, void *stream);
   int (*reverse_end)(void *ctx, double *v, void *stream);
} mfem_amd_comm;

/* RAPOperator(P, ConstrainedOperator(A), P)::Mult (linalg/operator.cpp:433,586-646):
 * forward(x) [ghosts of x are overwritten], y = A x on local dofs, reverse(y).
 * op->ess_marker: 1 = essential & owned (y = x), 2 = essential ghost copy (y = 0). */
int mfem_amd_pa_par_mult(const mfem_amd_pa_operator *op, const mfem_amd_comm *comm, double *x,
                         double *y, void *stream);
/* CGSolver(MPI_Comm)::Mult (linalg/solvers.cpp:44-54,869-1050) on the partitioned
 * operator: mfem_amd_pcg_pa with every dot product all-reduced (one double each, the
 * reference's message shape) and the operator wrapped in the exchange.  b must hold 0 in
 * ghost entries; x is returned consistent on owners and ghosts.  All ranks take identical
 * decisions (they see the same reduced scalars). */
int mfem_amd_pcg_pa_par(const mfem_amd_pa_operator *op, const mfem_amd_comm *comm,
                        const double *dinv, const double *b, double *x, double rel_tol,
                        double abs_tol, int max_iter, int check_every, void *workspace,
                        mfem_amd_pcg_result *result, double *hist, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MFEM_AMD_H */

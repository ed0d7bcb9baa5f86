// Sum-factorised PA operator action for H1 hexahedra on gfx950 (MI355X).
//
// Replaces: SmemPADiffusionApply3D<D,Q> / PADiffusionApply3D
//           (fem/integ/bilininteg_diffusion_kernels.hpp:988-1211, 795-985),
//           SmemPAMassApply3D<D,Q> / PAMassApply3D_Element
//           (fem/integ/bilininteg_mass_kernels.hpp:1119-1144, 807-1033, 656-805),
//           and, in the fused L->E form, ElementRestriction::Mult
//           (fem/restriction.cpp:109-129) + the `localY = 0` memset
//           (fem/bilinearform_ext.cpp:543-548).
//
// Design (MI355X-first, not the reference's one-element-per-block Q^3-thread layout):
//  * A workgroup holds NBZ elements; each element is worked by a Q x Q thread sheet
//    (thread = (qx,qy)), so Q*Q*NBZ ~ 200..256 threads = 4 full 64-lane wavefronts even at
//    p = 1 (the reference launches 27 threads per block at p = 1).
//  * The z direction lives in registers: every thread keeps its z-column of the x/y
//    contracted fields (3*D doubles), runs the qz loop -- contraction to the quadrature
//    point, the 3x3 symmetric D matvec, transposed contraction back to D z-values --
//    entirely in VGPRs.  Only the x and y contractions go through LDS.
//  * B and G travel in the kernel-argument segment (BasisT by value): the z-direction
//    factors are wave-uniform and compile-time indexed -> scalar loads + SGPR operands of
//    v_fma_f64; thread-indexed rows are staged once in LDS.
//  * pa_data (the dominant HBM stream, 48*Q^3 B per element) is fetched into registers at
//    the very top of the kernel, before the first barrier, so the whole stream is in
//    flight while the x/y contractions run; it is read exactly once, in the reference's
//    (q, component, element) layout.
//  * Logical block ids are remapped so every XCD (private L2) works on one contiguous
//    element range.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace mfa
{

// Register budget hint (waves per SIMD the kernel should fit, 2nd __launch_bounds__ argument
// on AMD): at p <= 4 diffusion fits 3 waves/SIMD = 3 resident workgroups per CU.
constexpr int minw_for(int KIND, int D, int PF = 0, int RB = 1, bool VEC = false, int Q = 0)
{
   // (5,8): only the dispatched form (two z-slices streamed) fits three waves; the override-only forms get two
   if (KIND && D == 5 && Q >= 8 && PF != 2) { return 2; }
#ifndef MFA_MINW_PF          // (tuning builds: the register budget of the streamed diffusion kernels at D <= 6, in waves per SIMD)
#define MFA_MINW_PF 3
#endif
   if (PF > 0) { return KIND ? (D <= 6 ? MFA_MINW_PF : 2) : (D <= 7 ? 4 : 3); }   // streamed pa_data: fewer live registers
   // D = 5 diffusion fits the 168 VGPRs of three waves per SIMD only in its register-blocked scalar form; the
   // component loop (VEC) and the one-output-per-thread form (RB = 0: MFEM layout, AddMultPA) spilled 12-96 B per
   // lane under that cap: two waves for them, no scratch (profiles/r2/d5_spill_probe.txt: BP4 p=4 320 -> 274 us,
   // BP6 p=4 190 -> 204 us, MFEM-layout BP3 p=4 407 -> 436 us)
   if (KIND && D == 5 && (VEC || RB == 0 || (RB == 2 && Q == 6))) { return 2; }   // (5,6) RB = 2: override-only form
   return KIND ? (D <= 5 ? 3 : D <= 7 ? 2 : 1) : (D <= 7 ? 3 : 2);
}

// Staged stores (kernel comment at the copy-out): which specialisations send their packed E-vector entries and interior rows
// through LDS and out as whole lines.  Measured A/B on one box, whole apply at 1e7 dofs on the reference's numbering
// (profiles/r3/staged_stores_ab.txt): BP1 p = 2..7 +1.3..4.0 %, p = 8 -1.4 %; BP3 p = 2 +0.8, p = 3 -1.5, p = 4 +2.5, p = 5, 6
// +0.3 / 0, p = 7 -1.0, p = 8 +0.9 %; BP5 p = 2, 3 -0.6 / -3.5, p = 4..7 +0.8 / 3.3 / 3.8 / 0.5, p = 8 -4.7 %; D = 8 re-measured with
// the whole-column schedule it got later (staged_stores_ab.txt, end): (8,9) staged 0.279 -> 0.466 ms, (8,8) -1.5 %, mass +3.5 %.
// MFA_STAGE_STORE=0 | 1 (build) forces it off / on everywhere.
#ifndef MFA_STAGE_STORE
#define MFA_STAGE_STORE -1
#endif
constexpr bool stage_store_for(int KIND, int D, int Q)
{
   if (MFA_STAGE_STORE >= 0) { return MFA_STAGE_STORE != 0; }
   // Re-measured on the one-trip kernels (staged_stores_ab.txt, end; whole apply): every shape that had lost with the looping
   // form gains now that the copy-out has registers -- (4,5) +2.3 %, (3,3) +3.8 %, (8,9) +1.5 % (it was 0.466 ms against 0.279
   // with the loop), (9,9) +4 %, mass (9,10) +4..6 %, (4,4) +0.3 % -- except (8,8): -1.2 %
   return !(KIND == 1 && D == 8 && Q == 8);
}

// KIND: 0 mass, 1 diffusion.  GATHER: x is an L-vector read through gmap (entries < 0 read
// as 0.0), else x is an E-vector.  ACCUM: yE += (AddMultPA semantics) else yE =.
//
// Persistent workgroups: the grid is (resident workgroups per CU) x (CUs); workgroup w
// walks a contiguous range of element batches.  The loads of batch k+1 (gather indices
// at the top of the iteration, then the x values and the pa_data z-columns right after
// the z loop of batch k, into the registers batch k has just finished with) stay in
// flight across the remaining barriers of batch k and the x/y contractions of batch k+1,
// so HBM is being read during every phase except the z loop itself.  Issue order matters:
// vmcnt retires in order, so the (small, needed-first) x gather is issued before the
// pa_data stream.
// ABL (ablation builds only: -DMFA_ABLATION, tools/ablate.hip): 0 = the kernel; 1 = memory traffic only (loads +
// stores, no contractions); 2 = contractions only (pa_data not read).  With 2, further bits switch
// parts of the contractions off (timing only, results meaningless): 512 = basis factors are
// literals (no scalar loads), 1024 = no z loop, 2048 = no y contraction, 4096 = no transposed y.
// PF > 0 (blocked layout only): pa_data is streamed through the z loop instead of being held
// whole in registers -- PF z-slices (6 doubles per thread each) are in flight at any time;
// slice qz+PF is requested when slice qz is consumed, and in the persistent schedule the ring
// runs on into the next batch, so every wave keeps a steady stream of requests up through all
// phases.  Frees 2*NC*(Q-PF) registers per thread: one more wave per SIMD at p >= 4.
#ifndef MFA_VGPR_CAP
#define MFA_VGPR_CAP
#endif
// Where an element's Q x Q thread sheet tiles the 64-lane wavefront exactly (Q = 4: four elements
// per wave, Q = 8: one), every LDS region of an element is written and read by ONE wave only:
// the phases need no workgroup barrier, only program order within the wave (LDS instructions of a
// wave execute in order) -- a compiler-level fence.  MFA_WAVE_SYNC=0 restores the barriers.
// basis factors as scalar operands (ablation builds, ABL & 512: literals instead)
// MFA_ABLATION (defined by tools/ablate*.hip before they include this file) compiles the ablation branches in;
// the library is built without it and contains none of them.
#ifdef MFA_ABLATION
#define MFA_ABL(mask) ((ABL & (mask)) != 0)
#define RBB(lo, UP, idx, off) ((ABL & 512) ? (0.25 + 0.0625*(idx)) : bg.lo[(idx) + (off)])
#define ZBB(lo, idx, off) ((ABL & 512) ? (0.25 + 0.0625*(idx)) : bg.lo[(idx) + (off)])
#else
#define MFA_ABL(mask) false
#define RBB(lo, UP, idx, off) bg.lo[(idx) + (off)]
#define ZBB(lo, idx, off) bg.lo[(idx) + (off)]
#endif
#ifndef MFA_WAVE_SYNC
#define MFA_WAVE_SYNC 1
#endif
#ifndef MFA_SPLIT_O
#define MFA_SPLIT_O 1
#endif
// In-place transposed y contraction (diffusion, RB = 1, one batch per workgroup; kernel comment at INPL): which
// specialisations drop LDS region V.  It pays where LDS was what limited the resident workgroups and the registers allow one
// more (profiles/r3/inplace_ab.txt, whole apply at 1e7 dofs): (6,7) 2 -> 3 workgroups per CU, BP3 p = 5 0.611 -> 0.643;
// (6,6) 0.602 -> 0.639 (with RB 1 instead of 2), (7,7) 0.599 -> 0.639, (8,8) 3 -> 4 workgroups, 0.615 -> 0.652.  Where the
// registers are the limit anyway it only adds bank conflicts to the last phase's row reads: (7,8) 0.665 -> 0.649, (5,5) with
// RB 1 0.577 -> 0.570, (5,6) / (4,5) / (8,9) +-0.  MFA_INPLACE=0 | 1 (build) forces it off / on for every eligible kernel.
#ifndef MFA_INPLACE
#define MFA_INPLACE -1
#endif
constexpr bool inplace_for(int D, int Q)
{
   if (MFA_INPLACE >= 0) { return MFA_INPLACE != 0; }
   if (Q == D) { return D >= 6; }        // ((9,9): 89 -> 54 KB, 1 -> 2 workgroups per CU: 263 -> 206 us, BP5 p = 8 0.48 -> 0.61)
   return D == 6 && Q == 7;
}
#ifndef MFA_ONE_TRIP
#define MFA_ONE_TRIP 1
#endif
#ifndef MFA_O_EARLY
#define MFA_O_EARLY 0
#endif
#ifndef MFA_STAGE_W16
#define MFA_STAGE_W16 -1
#endif
// tuning builds: -DMFA_ONLY_D=<D1D> compiles the specialisations of one D1D only (seconds instead of minutes)
#ifndef MFA_ONLY_D
#define MFA_ONLY_D 0
#endif
// pa_data and the gather map are read exactly once per application.  When pa_data is larger than the Infinity Cache
// both are loaded with the non-temporal policy (template argument NTP), which leaves the caches to what IS re-used --
// the gathered x values and the E-vector lines the E->L kernel re-reads.  Timed where it matters, in the alternating
// sequence of the two apply kernels (profiles/r2/pa_nt_in_context.txt): BP3 p = 2..5 4.3 / 4.0 / 3.4 / 1.6 % faster
// than cached loads for pa_data and another 4.6 % at p = 5 for the map, BP1 p = 5..7 4 %, nowhere slower at 1e7 dofs;
// the kernel timed in a loop of its own had suggested a per-shape table (profiles/r2/nt_probe.txt).  A pa_data that
// fits the cache (1e6 dofs: 160 MB at BP3 p = 4) is better left in it between applications (5 %), so the launcher
// picks the instantiation by size; a run-time switch inside the kernel cost what the policy gains (3 % at p = 3, 4).
// The streamed (PF > 0) kernels keep cached loads: non-temporal slices cost (8,9) / (9,10) 15-30 %.
// MFEM_AMD_APPLY_NT=0|1 forces the choice (tuning aid).
template <bool NT> __device__ __forceinline__ double load_pa(const double *p)
{
   return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ __forceinline__ double2 load_pa(const double2 *p)
{
   if (!NT) { return *p; }
   typedef double v2d __attribute__((ext_vector_type(2)));
   const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
   double2 r; r.x = v.x; r.y = v.y;
   return r;
}
template <bool WS>
__device__ __forceinline__ void phase_sync()
{
   if (WS)
   {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   }
   else { __syncthreads(); }
}

// RB (register-blocked x / transposed-y / transposed-x contractions): instead of one output per
// thread and sweep (one LDS read per FMA), a thread owns one LDS row -- (dy,dz) for the x
// contractions, (qx,dz) for the transposed y contraction -- reads it once and produces every
// output along the contracted direction from registers, with the basis factors as scalar
// operands.  Same sums in the same order (bit-identical results), ~half the LDS traffic, which is
// the busiest unit of the kernel at p >= 5 (profiles/r1/kbench_bp3_p5_pmc.txt: LDS active 59 %).
// LDS rows are padded to an odd number of doubles so the row-per-thread accesses are
// conflict-free.
//
// VEC (vector H1 spaces, VectorMassIntegrator / VectorDiffusionIntegrator with a scalar or vector
// coefficient: fem/integ/bilininteg_vecdiffusion_pa.hpp:108-170 applies the scalar operator to
// every component): the workgroup loops over the vd components of its elements with the gather
// indices -- and, when all components share one pa_data (scalar coefficient), the pa_data
// columns -- held in registers: pa_data is read once for vd components, where the reference
// stores and reads vd identical copies.  E-vectors are (D,D,D,vd,NE) as in the reference.
typedef __attribute__((address_space(1))) double gdouble;   // global memory, for addresses built from integers

template <int KIND, int D, int Q, int NBZ, bool GATHER, bool ACCUM, bool BLK = false, int ABL = 0, bool PERSIST = true, int PF = 0, int RB = 0, bool VEC = false, bool NTP = false>
__global__ void __launch_bounds__(Q*Q*NBZ, minw_for(KIND, D, PF, RB, VEC, Q)) MFA_VGPR_CAP
k_pa_apply(const int64_t ne, const int64_t nbatches, const int sched, const BasisTT<D*Q> bg,
           const double *__restrict__ Dq, const double *__restrict__ x,
           const int *__restrict__ gmap, double *__restrict__ yE, const VecArgs va)
{
   static_assert(!VEC || (!PERSIST && PF == 0 && ABL == 0), "the component loop runs in the one-batch-per-workgroup schedule");
   constexpr int NC = KIND ? 6 : 1;       // stored components per quadrature point
   constexpr int NF = KIND ? 3 : 1;       // fields carried through the z loop
   constexpr int NX = KIND ? 2 : 1;       // fields after the x contraction (B x, G x)
   constexpr bool DIRECT = GATHER && !ACCUM && !VEC && ABL == 0 && D >= 3;   // va.ydirect honoured
   constexpr bool STG = DIRECT && stage_store_for(KIND, D, Q);               // ... through LDS, as whole lines
   // (16 bytes a lane in the copy-out; MFA_STAGE_W16 = 0 forces 8.  The (5,6) diffusion kernel kept 8-byte lanes while its
   // looping form had no register to spare -- the wide form cost it a spilled x value; without the loop 16 bytes are 0.25 %
   // faster there too, profiles/r3/staged_stores_ab.txt, end)
   constexpr bool STG16 = MFA_STAGE_W16 != 0;
   // LDS strides (doubles): rows that one thread walks (sX, sBX/sGX, sV) are padded to an odd
   // length; the (qy,qx) planes of sT keep dense rows and get a small pad as a whole for even Q.
   // (Measured: padding or transposing the planes so that the row-per-thread reads are entirely
   // conflict-free -- they cost 20 % conflict cycles at p = 5, profiles/r1/kbench_bp3_p5_rb_pmc.txt
   // -- is no faster, and at p = 4 the extra LDS costs a resident workgroup: 347 -> 382 us.)
   constexpr int DP = RB ? (D | 1) : D, QP = RB ? (Q | 1) : Q;
   constexpr int PLPAD = ((Q - Q*Q) % 32 + 32) % 32;
   constexpr int QQ = Q*Q, PL = QQ + ((RB && PLPAD <= 8) ? PLPAD : 0);
   auto tix = [](const int plane, const int qy, const int qx) { return plane*PL + qy*Q + qx; };
   constexpr int D3 = D*D*D, DDQ = D*D*QP, DQQ = D*PL;
   constexpr int Q3 = Q*Q*Q;
   constexpr int SXN = D*D*DP;
   // RB == 2: the x-contracted fields live in region V, so region A holds sX, then the
   // y-contracted z-columns (sR), then sT
   constexpr int SA1 = (RB == 2) ? SXN : SXN + NX*DDQ;
   constexpr int SA = (SA1 > NF*DQQ) ? SA1 : NF*DQQ;
   constexpr int SV = NX*DDQ;
   constexpr int NXV = (D3 + QQ - 1)/QQ;  // x values staged per thread
   // INPL: thread (qx,dz) of the transposed y contraction reads its three columns of sT (region A) and nobody else's, and it
   // has read them all before it stores: its D outputs per field can go where the first D of its Q inputs were, and region V
   // is not needed at all (RB = 1 keeps the x-contracted fields in region A as well) -- a third less LDS per element, one
   // more resident workgroup where LDS was the limit.  The third field's planes are dead by the last phase: the staged
   // stores use them.  One batch per workgroup only (no next batch whose x values would land on region A's last readers).
   constexpr bool INPL = KIND == 1 && RB == 1 && !PERSIST && MFA_ONE_TRIP && !VEC && ABL == 0 && inplace_for(D, Q);
   constexpr int ESTRIDE = SA + (INPL ? 0 : SV);      // doubles of LDS per element
   constexpr int STGO = INPL ? 2*DQQ : 0;             // where the staged stores collect an element's values
   static_assert(!INPL || STGO + D3 <= SA, "staging area inside the third field's planes");
   __shared__ double smem[NBZ][ESTRIDE];
   __shared__ double sB[Q*D], sG[Q*D];

   const int a = threadIdx.x, b = threadIdx.y, tz = threadIdx.z;
   const int t2 = a + Q*b;
   // contiguous batch range of this workgroup; XCD-aware (workgroups w, w+8, ... share an L2)
   // batch schedule: SCHED 0 = contiguous range per workgroup (XCD-aware); 1 = round-robin
   // (workgroup w takes batches w, w+G, ...: at any time the grid works on G consecutive
   // batches, so the chip reads/writes one moving window of memory, like a streaming copy)
   // PERSIST = false: one batch per workgroup (grid = nbatches), nothing prefetched.
   const int SCHED = PERSIST ? sched : 1;
   const int64_t G_ = gridDim.x;
   // one batch per workgroup with sched == 2: workgroups b, b+8, ... (one XCD, one L2) take consecutive batches
   // sched >= 4 (one batch per workgroup): runs of `sched` consecutive batches go to one XCD -- workgroups w, w+8, ... of
   // a window of 8*sched -- so that neighbouring batches share an L2 while every XCD stays in the same window of memory
   int64_t L = (PERSIST ? SCHED != 0 : sched != 2) ? (int64_t)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
   if (!PERSIST && sched >= 4)
   {
      const int64_t per = 8*(int64_t)sched, bid = blockIdx.x;
      if (bid < (G_/per)*per) { L = ((bid/per)*8 + (bid & 7))*sched + ((bid >> 3) % sched); }
   }
   const int64_t b0 = SCHED ? L : (L*nbatches)/G_;
   // One batch per workgroup: a trip count the compiler can see (MFA_ONE_TRIP).  As a loop over [b0, nbatches) in steps of
   // the grid size -- one trip at run time, unknown at compile time -- the body "contains stores, no loads, and uses values
   // loaded outside", and for such a loop the compiler drains every outstanding load in the preheader (s_waitcnt vmcnt(0)):
   // the whole pa_data column had to land before the x values could even be staged.  Without the loop the waits are
   // counted: the x / y contractions run while pa_data is still arriving, and the z loop starts on its first slices.
   constexpr bool ONE = !PERSIST && MFA_ONE_TRIP;
   const int64_t b1 = ONE ? b0 + 1 : (SCHED ? nbatches : ((L + 1)*nbatches)/G_);
   const int64_t bstep = ONE ? 1 : (SCHED ? G_ : 1);
   if (b0 >= (ONE ? nbatches : b1)) { return; }              // uniform over the workgroup, before any barrier

   double *sA = smem[tz];
   double *sV = sA + SA;
   double *sX = sA, *sBX = (RB == 2) ? sV : sA + SXN, *sGX = sBX + DDQ;  // phase 1 view
   double *sT = sA;                                     // phase 2 view of region A

   for (int i = t2 + QQ*tz; i < Q*D; i += QQ*NBZ)
   {
      sB[i] = bg.b[i];
      if (KIND) { sG[i] = bg.g[i]; }
   }
   // DIRECT: where the D entries of store-phase row (dz,dy) go in the packed E-vector (common.h: packed_pos) -- first entry,
   // distance to the last, first of the D-2 middle ones (inner rows: offset of the interior run in the L-vector), packed
   // into one int per row.  Computed here, where registers are free, and read back from LDS in the store phase: derived
   // there (or hoisted here by the compiler) the same integers cost the (5,6) kernel its third wave's registers.
   __shared__ int sTab[DIRECT ? D*D : 1];
   if (DIRECT && va.ydirect && tz == 0 && t2 < D*D)
   {
      constexpr int DI = D - 2;
      const int rz = t2/D, ry = t2 % D;
      const bool inner = rz >= 1 && rz <= DI && ry >= 1 && ry <= DI;
      const int p0 = packed_pos(D, rz, ry, 0), pl = packed_pos(D, rz, ry, D - 1);
      const int pm = inner ? ((rz - 1)*DI + (ry - 1))*DI : packed_pos(D, rz, ry, 1);
      sTab[t2] = p0 | ((pl - p0) << 10) | (pm << 20) | (inner ? (1 << 30) : 0);
   }
   constexpr bool WS = MFA_WAVE_SYNC && (64 % QQ == 0) && (ABL == 0 || MFA_ABL(2));
   if (WS) { __syncthreads(); }           // the basis tables are shared by all waves: the only workgroup barrier

   static_assert(PF == 0 || (BLK && ABL == 0 && PF < Q), "streamed pa_data needs the blocked layout");
   double O[Q][NC];
   double On[PF > 0 ? PF : 1][NC];        // first PF slices of the next batch (persistent schedule)
   double xv[NXV];
   int gi[NXV];
   int comp = 0;                          // VEC: component being worked on

   // ---- loaders (plain loads into registers; they survive the barriers) ----
   // Every load is unconditional: out-of-range threads / elements read a clamped, valid
   // address and their values are never stored.  Straight-line loads (no exec-mask
   // branches) let the compiler place counted s_waitcnt vmcnt(N) instead of vmcnt(0), so
   // the prefetched pa_data stays in flight while the gathered x values are consumed.
   const int64_t elast = ne - 1;
   auto load_gi = [&](const int64_t e)
   {
#ifdef MFA_ABLATION
      if (ABL & 8)
      {
#pragma unroll
         for (int k = 0; k < NXV; k++) { gi[k] = 0; }
         return;
      }
#endif
      if (GATHER)
      {
         const int *gm = gmap + (e < elast ? e : elast)*D3;
#pragma unroll
         for (int k = 0; k < NXV; k++)
         {
            const int i = t2 + k*QQ;
            // the gather map is read once per application too: same policy as pa_data (BP3 p = 5 4.6 %, BP1 p = 5..7 4 %
            // on the apply, profiles/r2/pa_nt_in_context.txt)
            const int *gp = &gm[i < D3 ? i : D3 - 1];
            gi[k] = NTP ? __builtin_nontemporal_load(gp) : *gp;
         }
      }
   };
   auto load_xv = [&](const int64_t e, const int cc = 0)
   {
#ifdef MFA_ABLATION
      if (ABL & 8)
      {
#pragma unroll
         for (int k = 0; k < NXV; k++) { xv[k] = 1.0; }
         return;
      }
#endif
      const int64_t ec = e < elast ? e : elast;
      const double *xe = x + (VEC ? ec*va.vd + cc : ec)*D3;
#pragma unroll
      for (int k = 0; k < NXV; k++)
      {
         const int i = t2 + k*QQ;
         if (GATHER && VEC)
         {
            const int64_t g = gi[k] >= 0 ? gi[k] : 0;
            const int64_t idx = va.byvdim ? g*va.vd + cc : g + cc*va.ndofs;
            const double v = x[idx];
            const bool masked = va.marker ? (va.marker[idx] != 0) : false;
            xv[k] = (gi[k] >= 0 && !masked) ? v : 0.0;
         }
         else if (GATHER)
         {
            // (one batch per workgroup: the mask is applied when the value is staged -- here it would wait for the value
            // in the middle of the pa_data requests)
            const double v = x[gi[k] >= 0 ? gi[k] : 0];
            xv[k] = (ONE || gi[k] >= 0) ? v : 0.0;
         }
         else { xv[k] = xe[i < D3 ? i : D3 - 1]; }
      }
   };
   auto load_O = [&](const int64_t e, const int64_t bt, const int qz0 = 0, const int qz1 = Q)
   {
#ifdef MFA_ABLATION
      if (ABL == 2)
      {
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
#pragma unroll
            for (int c = 0; c < NC; c++) { O[qz][c] = xv[0] + (qz*NC + c); }
         }
         return;
      }
#endif
      if (BLK)
      {
         // blocked layout (common.h): whole-workgroup contiguous runs, 16 B per lane
         if (KIND)
         {
            const double2 *Db = reinterpret_cast<const double2*>(Dq + (VEC ? comp*va.d_comp_stride : 0))
                                + bt*(int64_t)(3*Q*NBZ*QQ) + tz*QQ + b*Q + a;
#pragma unroll
            for (int qz = 0; qz < Q; qz++)
            {
               if (qz < qz0 || qz >= qz1) { continue; }
#pragma unroll
               for (int cp = 0; cp < 3; cp++)
               {
                  const double2 v = load_pa<NTP>(&Db[(qz*3 + cp)*(NBZ*QQ)]);
                  O[qz][(2*cp) % NC] = v.x; O[qz][(2*cp + 1) % NC] = v.y;
               }
            }
         }
         else
         {
            const double *Db = Dq + (VEC ? comp*va.d_comp_stride : 0) + bt*(int64_t)(Q*NBZ*QQ) + tz*QQ + b*Q + a;
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { if (qz >= qz0 && qz < qz1) { O[qz][0] = load_pa<NTP>(&Db[qz*(NBZ*QQ)]); } }
         }
         return;
      }
      const int64_t ec = e < elast ? e : elast;
      const int64_t blk = (VEC && va.d_comp_stride) ? ec*va.vd + comp : ec;   // (Q^3,NC,vd,NE) or (Q^3,NC,NE)
      const double *De = Dq + blk*((VEC && va.d_block) ? va.d_block : (int64_t)(NC*Q3)) + b*Q + a;
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         if (qz < qz0 || qz >= qz1) { continue; }
#pragma unroll
         for (int c = 0; c < NC; c++) { O[qz][c] = De[c*Q3 + qz*QQ]; }
      }
   };

   // one z-slice of the blocked pa_data of batch bt (PF > 0)
   auto load_slice = [&](const int64_t bt, const int qz, double (&dst)[NC])
   {
      // (plain loads: a policy switch inside the streaming ring cost the (8,9) / (9,10) kernels 15-30 %)
      if (KIND)
      {
         const double2 *Db = reinterpret_cast<const double2*>(Dq) + bt*(int64_t)(3*Q*NBZ*QQ) + tz*QQ + b*Q + a;
#pragma unroll
         for (int cp = 0; cp < 3; cp++)
         {
            const double2 v = Db[(qz*3 + cp)*(NBZ*QQ)];
            dst[(2*cp) % NC] = v.x; dst[(2*cp + 1) % NC] = v.y;
         }
      }
      else { dst[0] = Dq[bt*(int64_t)(Q*NBZ*QQ) + tz*QQ + b*Q + a + qz*(NBZ*QQ)]; }
   };

   // the gathered x values go to LDS (region sX)
   auto stage_x = [&]()
   {
#pragma unroll
      for (int k = 0; k < NXV; k++)
      {
         // unconditional: a thread past the end holds the LAST entry's value (the loaders clamp the same way) and writes it
         // where its owner does.  Behind an `if` the compiler sinks that request to the store -- below the pa_data stream,
         // and drains the stream there (s_waitcnt vmcnt(0)) -- once loads and store share a block (MFA_ONE_TRIP)
         const int i = (t2 + k*QQ < D3) ? t2 + k*QQ : D3 - 1;
         sX[RB ? (i/D)*DP + i % D : i] = (ONE && GATHER && !VEC && gi[k] < 0) ? 0.0 : xv[k];
      }
   };
   // One batch per workgroup, whole pa_data columns in registers: the x values can be staged between the two halves of the
   // pa_data requests, so that their registers are free again before the second half is allocated.  It removes the one
   // spilled x value of the (5,6) kernel (whose reload waits for EVERY outstanding load, vmcnt(0)) -- and still loses there:
   // the second half of the stream is requested one memory latency later (profiles/r3/split_o_ab.txt, whole apply, two
   // runs each: p = 3 0.563 vs 0.582, p = 4 0.590 vs 0.605, p = 5 0.570 vs 0.580, p = 6 0.620 vs 0.632).  Only D = 3, with
   // four short columns per thread, gains (p = 2: 0.560 vs 0.537), and keeps it.
   constexpr bool SPLIT = MFA_SPLIT_O && D == 3 && !PERSIST && !VEC && PF == 0 && ABL == 0 && KIND == 1;
   {
      const int64_t e = b0*NBZ + tz;
      load_gi(e);
      // tuning builds (-DMFA_O_EARLY=<slices>, < 0: all): request that many z-slices of pa_data BEFORE the gather waits for
      // its indices, so that the index latency of a one-batch workgroup is not spent with nothing in flight
      constexpr int OE = (!PERSIST && PF == 0 && !SPLIT && !VEC && ABL == 0 && BLK && GATHER) ? (MFA_O_EARLY < 0 || MFA_O_EARLY > Q ? Q : MFA_O_EARLY) : 0;
      if (OE > 0) { load_O(e, b0, 0, OE); }
      load_xv(e);
      if (SPLIT) { load_O(e, b0, 0, Q/2); stage_x(); load_O(e, b0, Q/2, Q); }
      else if (PF == 0) { load_O(e, b0, OE, Q); }
      else
      {
#pragma unroll
         for (int s = 0; s < PF; s++) { load_slice(b0, s, O[s]); }
      }
      // (with the body in the same block as these requests the scheduler would sink some of them to their first use --
      // behind the pa_data stream, and wait for all of it there)
      if (ONE) { __builtin_amdgcn_sched_barrier(0); }
   }

   for (int64_t bt = b0; bt < b1; bt += bstep)
   {
      const int64_t e = bt*NBZ + tz;
      const bool active = e < ne;
      const int64_t btn = (bt + bstep < b1) ? bt + bstep : bt;  // last batch: harmless re-read of itself
      const int64_t en = btn*NBZ + tz;
      // opaque 0: keeps the per-thread basis rows (LDS reads below) from being hoisted out of
      // the batch loop and held in ~90 VGPRs for the whole kernel
      int vz;
      asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
      for (comp = 0; comp < (VEC ? va.vd : 1); comp++)
      {
         // VEC: the x values of the next component are requested right after this one's have been
         // staged (below), so only the first gather of a workgroup is waited for
         if (VEC && comp > 0 && va.d_comp_stride) { load_O(e, bt); }

         if (!SPLIT) { stage_x(); }
         if (PERSIST) { load_gi(en); }       // next batch's gather indices: in flight from here
         if (VEC && comp + 1 < va.vd) { load_xv(e, comp + 1); }
         phase_sync<WS>();
#ifdef MFA_ABLATION
         if (ABL & 1)
         {
            double acc = sX[t2 % D];
#pragma unroll
            for (int qz = 0; qz < Q; qz++)
            {
#pragma unroll
               for (int c = 0; c < NC; c++) { acc += O[qz][c]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_xv(en);
            load_O(en, btn);
            if (ABL & 16)
            {
               const int tid = t2 + QQ*tz;
               const int nst = (ABL & 128) ? 64 : QQ*NBZ;   // 128: only wave 0 stores
               if (tid < nst)
               for (int i = tid; i < NBZ*D3; i += nst)
               {
                  if (bt*NBZ*D3 + i < ne*D3)
                  {
                     if (ABL & 32) { __builtin_nontemporal_store(acc, &yE[bt*NBZ*D3 + i]); }
                     else if (ABL & 64) { yE[(blockIdx.x*NBZ*D3 + i) % (ne*D3)] = acc; }
                     else { yE[bt*NBZ*D3 + i] = acc; }
                  }
               }
            }
            else if (active && !(ABL & 4))
            {
               for (int i = t2; i < D3; i += QQ) { yE[e*D3 + i] = acc; }
            }
            if ((ABL & 4) && acc == 1.2345e-300) { yE[0] = acc; }
            __syncthreads();
            continue;
         }
#endif

         // --- x contraction: thread (qx = a, dy = b); RB: thread t2 = row (dz,dy), all qx
         int zofs, zx, zy, zt;
         asm volatile("s_mov_b32 %0, 0" : "=s"(zx));
         if (RB)
         {
            if (t2 < D*D)
            {
               double sx[D];
#pragma unroll
               for (int dx = 0; dx < D; dx++) { sx[dx] = sX[t2*DP + dx]; }
#pragma unroll
               for (int qx = 0; qx < Q; qx++)
               {
                  double u = 0.0, v = 0.0;
#pragma unroll
                  for (int dx = 0; dx < D; dx++)
                  {
                     u += RBB(b, B, qx*D + dx, zx)*sx[dx];
                     if (KIND) { v += RBB(g, G, qx*D + dx, zx)*sx[dx]; }
                  }
                  sBX[t2*QP + qx] = u;
                  if (KIND) { sGX[t2*QP + qx] = v; }
               }
            }
         }
         else if (b < D)
         {
            double Ba[D], Ga[D];
#pragma unroll
            for (int dx = 0; dx < D; dx++) { Ba[dx] = sB[a*D + dx + vz]; Ga[dx] = KIND ? sG[a*D + dx + vz] : 0.0; }
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double u = 0.0, v = 0.0;
#pragma unroll
               for (int dx = 0; dx < D; dx++)
               {
                  const double s = sX[(dz*D + b)*DP + dx];
                  u += Ba[dx]*s;
                  if (KIND) { v += Ga[dx]*s; }
               }
               sBX[(dz*D + b)*QP + a] = u;
               if (KIND) { sGX[(dz*D + b)*QP + a] = v; }
            }
         }
         phase_sync<WS>();

         // --- y contraction into register z-columns: thread (qx = a, qy = b)
         double r1[D], r2[D], r3[D];
         if (RB == 2)
         {
            // thread t2 = (qx, dz) reads its 2*D inputs once and produces the Q outputs along y
            // into sR (the layout of sT); every thread then picks up its own z-column
            int zr;
            asm volatile("s_mov_b32 %0, 0" : "=s"(zr));
            if (t2 < Q*D)
            {
               const int qx = t2 % Q, dz = t2 / Q;
               double bx[D], gx[D];
#pragma unroll
               for (int dy = 0; dy < D; dy++)
               {
                  bx[dy] = sBX[(dz*D + dy)*QP + qx];
                  gx[dy] = KIND ? sGX[(dz*D + dy)*QP + qx] : 0.0;
               }
#pragma unroll
               for (int qy = 0; qy < Q; qy++)
               {
                  double u = 0.0, v = 0.0, w = 0.0;
#pragma unroll
                  for (int dy = 0; dy < D; dy++)
                  {
                     w += bg.b[qy*D + dy + zr]*bx[dy];
                     if (KIND)
                     {
                        u += bg.b[qy*D + dy + zr]*gx[dy];
                        v += bg.g[qy*D + dy + zr]*bx[dy];
                     }
                  }
                  sT[tix(dz, qy, qx)] = w;
                  if (KIND)
                  {
                     sT[tix(D + dz, qy, qx)] = v;
                     sT[tix(2*D + dz, qy, qx)] = u;
                  }
               }
            }
            phase_sync<WS>();
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               r3[dz] = sT[tix(dz, b, a)];
               r2[dz] = KIND ? sT[tix(D + dz, b, a)] : 0.0;
               r1[dz] = KIND ? sT[tix(2*D + dz, b, a)] : 0.0;
            }
         }
#ifdef MFA_ABLATION
         else if (ABL & 2048)
         {
#pragma unroll
            for (int dz = 0; dz < D; dz++) { r1[dz] = sBX[(dz*D + (b % D))*QP + a]; r2[dz] = r1[dz] + 1.0; r3[dz] = r1[dz] + 2.0; }
         }
#endif
         else
         {
            double Bb[D], Gb[D];
#pragma unroll
            for (int dy = 0; dy < D; dy++) { Bb[dy] = sB[b*D + dy + vz]; Gb[dy] = KIND ? sG[b*D + dy + vz] : 0.0; }
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double u = 0.0, v = 0.0, w = 0.0;
#pragma unroll
               for (int dy = 0; dy < D; dy++)
               {
                  const double bx = sBX[(dz*D + dy)*QP + a];
                  w += Bb[dy]*bx;
                  if (KIND)
                  {
                     const double gx = sGX[(dz*D + dy)*QP + a];
                     u += Bb[dy]*gx;
                     v += Gb[dy]*bx;
                  }
               }
               r1[dz] = u; r2[dz] = v; r3[dz] = w;
            }
         }

         // --- z loop in registers: to the quadrature point, D matvec, back to dofs in z
         // The wave-uniform basis factors are re-read from the kernel-argument segment every
         // iteration (scalar loads, scalar cache): `zofs` is an opaque 0 that keeps the
         // compiler from hoisting all 2*Q*D of them out of the batch loop, where they would
         // not fit the SGPR file and spill to VGPR lanes.
         asm volatile("s_mov_b32 %0, 0" : "=s"(zofs));
         double t1[D], t2r[D], t3[D];
#pragma unroll
         for (int dz = 0; dz < D; dz++) { t1[dz] = 0.0; t2r[dz] = 0.0; t3[dz] = 0.0; }
#ifdef MFA_ABLATION
         if (ABL & 1024)
         {
#pragma unroll
            for (int dz = 0; dz < D; dz++) { t1[dz] = r1[dz] + O[dz % Q][0]; t2r[dz] = r2[dz]; t3[dz] = r3[dz]; }
         }
         else
#endif
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            if (PF > 0)
            {
               // request the slice PF steps ahead: of this batch, or (ring) of the next one
               if (qz + PF < Q) { load_slice(bt, qz + PF, O[qz + PF < Q ? qz + PF : 0]); }
               else if (PERSIST) { load_slice(btn, qz + PF - Q, On[qz + PF >= Q ? qz + PF - Q : 0]); }
            }
            if (KIND)
            {
               double gX = 0.0, gY = 0.0, gZ = 0.0;
#pragma unroll
               for (int dz = 0; dz < D; dz++)
               {
                  const double bz = ZBB(b, qz*D + dz, zofs), gz = ZBB(g, qz*D + dz, zofs);
                  gX += bz*r1[dz];
                  gY += bz*r2[dz];
                  gZ += gz*r3[dz];
               }
               const double O11 = O[qz][0], O12 = O[qz][1], O13 = O[qz][2];
               const double O22 = O[qz][NC > 3 ? 3 : 0], O23 = O[qz][NC > 4 ? 4 : 0], O33 = O[qz][NC > 5 ? 5 : 0];
               const double fX = (O11*gX) + (O12*gY) + (O13*gZ);
               const double fY = (O12*gX) + (O22*gY) + (O23*gZ);
               const double fZ = (O13*gX) + (O23*gY) + (O33*gZ);
#pragma unroll
               for (int dz = 0; dz < D; dz++)
               {
                  const double bz = ZBB(b, qz*D + dz, zofs), gz = ZBB(g, qz*D + dz, zofs);
                  t1[dz] += bz*fX;
                  t2r[dz] += bz*fY;
                  t3[dz] += gz*fZ;
               }
            }
            else
            {
               double u = 0.0;
#pragma unroll
               for (int dz = 0; dz < D; dz++) { u += ZBB(b, qz*D + dz, zofs)*r3[dz]; }
               u *= O[qz][0];
#pragma unroll
               for (int dz = 0; dz < D; dz++) { t3[dz] += ZBB(b, qz*D + dz, zofs)*u; }
            }
         }
         // all reads of sBX/sGX done; region A is reused as sT (RB == 2: every thread overwrites
         // exactly the sR entries it read itself -- no barrier)
         if (RB != 2) { phase_sync<WS>(); }
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            sT[tix(dz, b, a)] = t3[dz];
            if (KIND)
            {
               sT[tix(D + dz, b, a)] = t2r[dz];
               sT[tix(2*D + dz, b, a)] = t1[dz];
            }
         }
         // --- next batch: x values first (needed first; vmcnt retires in order), then pa_data.
         // The scheduling fence keeps these loads below the z loop: hoisted above it they would
         // need a second set of 2*NC*Q registers and cost a wave of occupancy.
         __builtin_amdgcn_sched_barrier(0);
         if (PERSIST)
         {
            load_xv(en);
            if (PF == 0) { load_O(en, btn); }
         }
         phase_sync<WS>();

         // --- transposed y contraction: thread (qx = a, dy = b); RB: thread t2 = (qx, dz), all dy
         asm volatile("s_mov_b32 %0, 0" : "=s"(zy));
#ifdef MFA_ABLATION
         if (RB && (ABL & 4096))
         {
            if (t2 < Q*D)
            {
               const int qx = t2 % Q, dz = t2 / Q;
#pragma unroll
               for (int dy = 0; dy < D; dy++)
               {
                  sV[(dz*D + dy)*QP + qx] = sT[tix(dz, dy, qx)];
                  if (KIND) { sV[DDQ + (dz*D + dy)*QP + qx] = sT[tix(D + dz, dy, qx)]; }
               }
            }
         }
         else
#endif
         if (RB)
         {
            if (t2 < Q*D)
            {
               const int qx = t2 % Q, dz = t2 / Q;
               double v23[D], v1[D];
#pragma unroll
               for (int dy = 0; dy < D; dy++) { v23[dy] = 0.0; v1[dy] = 0.0; }
#pragma unroll
               for (int qy = 0; qy < Q; qy++)
               {
                  const double T0 = sT[tix(dz, qy, qx)];
                  const double T1 = KIND ? sT[tix(D + dz, qy, qx)] : 0.0;
                  const double T2 = KIND ? sT[tix(2*D + dz, qy, qx)] : 0.0;
#pragma unroll
                  for (int dy = 0; dy < D; dy++)
                  {
                     v23[dy] += RBB(b, B, qy*D + dy, zy)*T0;
                     if (KIND)
                     {
                        v23[dy] += RBB(g, G, qy*D + dy, zy)*T1;
                        v1[dy] += RBB(b, B, qy*D + dy, zy)*T2;
                     }
                  }
               }
#pragma unroll
               for (int dy = 0; dy < D; dy++)
               {
                  if (INPL) { sT[tix(dz, dy, qx)] = v23[dy]; sT[tix(D + dz, dy, qx)] = v1[dy]; }
                  else
                  {
                     sV[(dz*D + dy)*QP + qx] = v23[dy];
                     if (KIND) { sV[DDQ + (dz*D + dy)*QP + qx] = v1[dy]; }
                  }
               }
            }
         }
         else if (b < D)
         {
            double Bc[Q], Gc[Q];
#pragma unroll
            for (int qy = 0; qy < Q; qy++) { Bc[qy] = sB[qy*D + b + vz]; Gc[qy] = KIND ? sG[qy*D + b + vz] : 0.0; }
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double v1 = 0.0, v23 = 0.0;
#pragma unroll
               for (int qy = 0; qy < Q; qy++)
               {
                  v23 += Bc[qy]*sT[tix(dz, qy, a)];
                  if (KIND)
                  {
                     v23 += Gc[qy]*sT[tix(D + dz, qy, a)];
                     v1 += Bc[qy]*sT[tix(2*D + dz, qy, a)];
                  }
               }
               sV[(dz*D + b)*QP + a] = v23;
               if (KIND) { sV[DDQ + (dz*D + b)*QP + a] = v1; }
            }
         }
         phase_sync<WS>();

         // --- transposed x contraction and store: thread (dx = a, dy = b); RB: thread t2 = row
         // (dz,dy), all dx -- D consecutive doubles of yE per thread
         asm volatile("s_mov_b32 %0, 0" : "=s"(zt));
         if (RB)
         {
            if (t2 < D*D && active)
            {
               double V23[Q], V1[Q];
#pragma unroll
               for (int qx = 0; qx < Q; qx++)
               {
                  if (INPL) { V23[qx] = sT[tix(t2/D, t2 % D, qx)]; V1[qx] = sT[tix(D + t2/D, t2 % D, qx)]; }
                  else { V23[qx] = sV[t2*QP + qx]; V1[qx] = KIND ? sV[DDQ + t2*QP + qx] : 0.0; }
               }
               double yv[D];
#pragma unroll
               for (int dx = 0; dx < D; dx++)
               {
                  double y = 0.0;
#pragma unroll
                  for (int qx = 0; qx < Q; qx++)
                  {
                     // transposed tables: this phase walks q fastest, and scalar operands want contiguous runs
                     y += RBB(bt, B, dx*Q + qx, zt)*V23[qx];
                     if (KIND) { y += RBB(gt, G, dx*Q + qx, zt)*V1[qx]; }
                  }
                  yv[dx] = y;
               }
               double *ye = yE + (VEC ? e*va.vd + comp : e)*D3 + t2*D;
               // rows of element-interior dofs belong to this element alone: their sums ARE these values, so they
               // go straight to the L-vector and never through the E-vector (uniform branch on a kernel argument);
               // what is left of the element goes to the packed E-vector (common.h: packed_pos): this row's two end
               // entries and its run of D-2 middle entries -- an x-edge, a row of a z- or y-face, or (inner rows)
               // the interior run
               if (STG && va.ydirect)
               {
                  // staged stores (below): the row's values go to their packed positions in the element's dead region A --
                  // [0, NS) the surviving entries, [NS, D^3) the interior run -- and leave the workgroup as two contiguous runs
                  constexpr int NSP = packed_size(D);
                  const int code = sTab[t2];
                  const int p0 = code & 1023, dl = (code >> 10) & 1023, pm = ((code >> 20) & 1023) + (((code >> 30) & 1) ? NSP : 0);
#pragma unroll
                  for (int dx = 0; dx < D; dx++) { sA[STGO + (dx == 0 ? p0 : (dx == D - 1 ? p0 + dl : pm + dx - 1))] = yv[dx]; }
               }
               else if (DIRECT && va.ydirect)
               {
                  // One store per value, its address chosen per lane by mask arithmetic: the rows of one wave differ in
                  // kind, and a branch would issue both forms of the row one after the other.  (Not `c ? p : q` on the
                  // pointers: hipcc 7.2 turns that back into exec-masked blocks and sinks one of the moves out of its
                  // block -- lanes of dy = 0 then stored through their neighbour row's address.)  The row's positions come
                  // from the table filled at the top of the kernel.
                  constexpr int DI = D - 2;
                  const int code = sTab[t2];
                  const unsigned p0 = code & 1023, dl = (code >> 10) & 1023, pm = (code >> 20) & 1023;
                  const uint64_t inner = (uint64_t)(-(int64_t)((code >> 30) & 1));
                  // (an opaque copy of the element's slot: the two base addresses are formed here, after the z loop, instead of
                  // being hoisted above it)
                  int ez;
                  asm volatile("v_mov_b32 %0, %1" : "=v"(ez) : "v"(tz));
                  const int64_t eo = bt*NBZ + ez;
                  const uint64_t pb = reinterpret_cast<uint64_t>(yE + eo*packed_size(D));
                  const uint64_t pd = reinterpret_cast<uint64_t>(va.ydirect + eo*(DI*DI*DI));
                  const uint64_t a0 = pb + 8u*p0;
                  const uint64_t am = (pb ^ ((pb ^ pd) & inner)) + 8u*pm;
#pragma unroll
                  for (int dx = 0; dx < D; dx++)
                  {
#ifdef MFA_PROBE_NOSTORE   // probe builds only (tools/r3_job13.sh): what the store phase costs -- the values stay live, nothing is written
                     if (yv[dx] != 1.2345e-300) { continue; }
#endif
                     *reinterpret_cast<gdouble*>(dx == 0 ? a0 : (dx == D - 1 ? a0 + 8u*(unsigned)dl : am + 8u*(dx - 1))) = yv[dx];
                  }
               }
               else if (D % 2 == 0)
               {
                  double2 *ye2 = reinterpret_cast<double2*>(ye);      // e*D3 + t2*D is even: 16-B aligned
#pragma unroll
                  for (int dx = 0; dx < D/2; dx++)
                  {
                     double2 o;
                     if (ACCUM) { o = ye2[dx]; o.x += yv[2*dx]; o.y += yv[2*dx + 1]; }
                     else { o.x = yv[2*dx]; o.y = yv[2*dx + 1]; }
                     ye2[dx] = o;
                  }
               }
               else
               {
#pragma unroll
                  for (int dx = 0; dx < D; dx++)
                  {
                     if (ACCUM) { ye[dx] += yv[dx]; }
                     else { ye[dx] = yv[dx]; }
                  }
               }
            }
         }
         else if (a < D && b < D && active)
         {
            double Bc[Q], Gc[Q];
#pragma unroll
            for (int qx = 0; qx < Q; qx++) { Bc[qx] = sB[qx*D + a + vz]; Gc[qx] = KIND ? sG[qx*D + a + vz] : 0.0; }
            double *ye = yE + (VEC ? e*va.vd + comp : e)*D3 + b*D + a;
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double y = 0.0;
#pragma unroll
               for (int qx = 0; qx < Q; qx++)
               {
                  y += Bc[qx]*sV[(dz*D + b)*QP + qx];
                  if (KIND) { y += Gc[qx]*sV[DDQ + (dz*D + b)*QP + qx]; }
               }
               if (STG && va.ydirect)
               {
                  constexpr int DI = D - 2, NSP = packed_size(D);
                  const int pp = packed_pos(D, dz, b, a);
                  sA[pp < 0 ? NSP + ((dz - 1)*DI + (b - 1))*DI + (a - 1) : pp] = y;      // staged stores (below)
               }
               else if (DIRECT && va.ydirect)
               {
                  constexpr int DI = D - 2;
                  const bool inner = a >= 1 && a <= DI && b >= 1 && b <= DI && dz >= 1 && dz <= DI;
                  const int pp = packed_pos(D, dz, b, a);
                  const uint64_t pe = reinterpret_cast<uint64_t>(yE + e*packed_size(D) + (pp < 0 ? 0 : pp));
                  const uint64_t pd = reinterpret_cast<uint64_t>(va.ydirect + e*(DI*DI*DI) + ((dz - 1)*DI + (b - 1))*DI + (a - 1));
                  *reinterpret_cast<gdouble*>(pe ^ ((pe ^ pd) & (uint64_t)(-(int64_t)inner))) = y;     // (mask, not ?: -- see the RB form)
               }
               else if (ACCUM) { ye[dz*D*D] += y; }
               else { ye[dz*D*D] = y; }
            }
         }
         if (STG && va.ydirect)
         {
            // Staged stores: a row thread's D values land at scattered packed positions, 8 bytes a lane -- tens of partial
            // write requests per wave instruction, and the memory pipeline of a CU moves requests, not bytes
            // (profiles/r3/store_probe.txt).  From LDS the workgroup's elements leave as two contiguous runs instead: the
            // batch's packed E-vector entries (NBZ*NS doubles) and its interior rows of y (NBZ*(D-2)^3), whole lines per
            // wave instruction.  Same values at the same addresses.
            constexpr int NSP = packed_size(D), NIP = D3 - NSP, STRIDE = ESTRIDE, NT = QQ*NBZ;
            __syncthreads();
            // (an opaque copy of the thread's index: everything the copy-out derives from it is formed here, after the z loop,
            // instead of being hoisted above it, where the (5,6) kernel has no register to spare)
            int tid;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(t2 + QQ*tz));
            const int64_t e0 = bt*NBZ;
            const int nact = (int)(ne - e0 < NBZ ? ne - e0 : NBZ);
            const double *sm = &smem[0][0];
            // 16 bytes a lane (8-byte accesses run at 0.5-0.7 of that rate, MI355X_MICROARCH.md): pairs (i, i+1) from the first
            // 16-byte-aligned double of the run on; its possible leading double and the run's last one go alone
            auto copy_out = [&](double *dst, const int n, auto per_c, auto off_c)
            {
               constexpr int per = decltype(per_c)::value, off = decltype(off_c)::value;
               auto at = [&](const int i) { const int ee = i/per; return sm[ee*STRIDE + STGO + off + (i - ee*per)]; };
               if (!STG16)
               {
#pragma unroll
                  for (int k = 0; k < (NBZ*per + NT - 1)/NT; k++)
                  {
                     const int i = tid + k*NT;
                     if (i < n) { dst[i] = at(i); }
                  }
                  return;
               }
               const int mis = (int)((reinterpret_cast<uint64_t>(dst) >> 3) & 1);
               const int npairs = (n - mis)/2;
#pragma unroll
               for (int k = 0; k < (NBZ*per/2 + NT - 1)/NT; k++)
               {
                  const int j = tid + k*NT;
                  if (j < npairs)
                  {
                     const int i = mis + 2*j;
                     double2 v; v.x = at(i); v.y = at(i + 1);
                     *reinterpret_cast<double2*>(dst + i) = v;
                  }
               }
               if (tid == 0 && mis && n > 0) { dst[0] = at(0); }
               if (tid == 1 && ((n - mis) & 1) && n > mis) { dst[n - 1] = at(n - 1); }
            };
            copy_out(yE + e0*NSP, nact*NSP, std::integral_constant<int, NSP>{}, std::integral_constant<int, 0>{});
            copy_out(va.ydirect + e0*NIP, nact*NIP, std::integral_constant<int, NIP>{}, std::integral_constant<int, NSP>{});
            if (PERSIST) { __syncthreads(); }      // the next batch's x values are staged in region A
         }
         // no barrier needed here: the next iteration writes sX (region A), whose last readers
         // (the transposed y contraction) are behind the barrier above; sV is next written
         // three barriers from now.
      }
      if (PF > 0 && PERSIST)
      {
#pragma unroll
         for (int s = 0; s < PF; s++)
         {
#pragma unroll
            for (int c = 0; c < NC; c++) { O[s][c] = On[s][c]; }
         }
      }
   }
}

// ------------------------------------------------------------------------------------
// Generic fallback (runtime D1D, Q1D), one workgroup per element, everything through LDS.
// Counterpart of the reference's non-specialised PADiffusionApply3D / PAMassApply3D
// fallbacks (kernel_dispatch.hpp:176-180).
// ------------------------------------------------------------------------------------
template <int KIND, bool GATHER, bool ACCUM>
__global__ void __launch_bounds__(256)
k_pa_apply_generic(const int D, const int Q, const int64_t ne, const BasisT<MAX_DQ> bg,
                   const double *__restrict__ Dq, const double *__restrict__ x,
                   const int *__restrict__ gmap, double *__restrict__ yE, const VecArgs va, const int comp)
{
   // KIND 2: diffusion with the full 3x3 D of a non-symmetric matrix coefficient (9 planes, row-major)
   constexpr int NC = KIND == 2 ? 9 : (KIND ? 6 : 1), NF = KIND ? 3 : 1, NX = KIND ? 2 : 1;
   extern __shared__ __align__(16) double dyn[];
   const int D3 = D*D*D, DDQ = D*D*Q, DQQ = D*Q*Q, Q3 = Q*Q*Q, QQ = Q*Q;
   double *sB = dyn, *sG = sB + Q*D;
   double *sX = sG + Q*D;            // D3
   double *s1 = sX + D3;             // NX*DDQ  [k][dz][dy][qx]
   double *s2 = s1 + NX*DDQ;         // NF*DQQ  [k][dz][qy][qx]
   double *s3 = s2 + NF*DQQ;         // NF*Q3   [k][qz][qy][qx]
   const int tid = threadIdx.x, nt = blockDim.x;
   const int64_t e = blockIdx.x;
   for (int i = tid; i < Q*D; i += nt) { sB[i] = bg.b[i]; sG[i] = bg.g[i]; }
   // va.vd == 0: scalar space; else component `comp` of a vector space (one launch per component)
   const int64_t eblk = va.vd ? e*va.vd + comp : e;            // E-vector block (D^3,vd,NE)
   const int64_t dblk = (va.vd && va.d_comp_stride) ? eblk : e;   // pa_data block
   for (int i = tid; i < D3; i += nt)
   {
      double v;
      if (GATHER)
      {
         const int g = gmap[e*D3 + i];
         if (va.vd)
         {
            const int64_t gg = g >= 0 ? g : 0;
            const int64_t idx = va.byvdim ? gg*va.vd + comp : gg + comp*va.ndofs;
            v = (g >= 0 && !(va.marker && va.marker[idx])) ? x[idx] : 0.0;
         }
         else { v = (g >= 0) ? x[g] : 0.0; }
      }
      else { v = x[eblk*D3 + i]; }
      sX[i] = v;
   }
   __syncthreads();
   for (int i = tid; i < DDQ; i += nt)
   {
      const int qx = i % Q, dy = (i/Q) % D, dz = i/(Q*D);
      double u = 0.0, v = 0.0;
      for (int dx = 0; dx < D; dx++)
      {
         const double s = sX[(dz*D + dy)*D + dx];
         u += sB[qx*D + dx]*s;
         v += sG[qx*D + dx]*s;
      }
      s1[i] = u;
      if (KIND) { s1[DDQ + i] = v; }
   }
   __syncthreads();
   for (int i = tid; i < DQQ; i += nt)
   {
      const int qx = i % Q, qy = (i/Q) % Q, dz = i/QQ;
      double u = 0.0, v = 0.0, w = 0.0;
      for (int dy = 0; dy < D; dy++)
      {
         const double bx = s1[(dz*D + dy)*Q + qx];
         w += sB[qy*D + dy]*bx;
         if (KIND)
         {
            const double gx = s1[DDQ + (dz*D + dy)*Q + qx];
            u += sB[qy*D + dy]*gx;
            v += sG[qy*D + dy]*bx;
         }
      }
      s2[i] = w;
      if (KIND) { s2[DQQ + i] = v; s2[2*DQQ + i] = u; }
   }
   __syncthreads();
   for (int i = tid; i < Q3; i += nt)
   {
      const int qq = i % QQ, qz = i/QQ;
      const double *De = Dq + dblk*((va.vd && va.d_block) ? va.d_block : (int64_t)(NC*Q3)) + i;
      if (KIND)
      {
         double gX = 0.0, gY = 0.0, gZ = 0.0;
         for (int dz = 0; dz < D; dz++)
         {
            const double bz = sB[qz*D + dz], gz = sG[qz*D + dz];
            gX += bz*s2[2*DQQ + dz*QQ + qq];
            gY += bz*s2[DQQ + dz*QQ + qq];
            gZ += gz*s2[dz*QQ + qq];
         }
         // plane order of the reference (bilininteg_diffusion_kernels.hpp:1119-1127)
         const double O11 = De[0], O12 = De[Q3], O13 = De[2*(int64_t)Q3];
         const double O21 = KIND == 2 ? De[3*(int64_t)Q3] : O12;
         const double O22 = De[(KIND == 2 ? 4 : 3)*(int64_t)Q3], O23 = De[(KIND == 2 ? 5 : 4)*(int64_t)Q3];
         const double O31 = KIND == 2 ? De[6*(int64_t)Q3] : O13;
         const double O32 = KIND == 2 ? De[7*(int64_t)Q3] : O23;
         const double O33 = De[(KIND == 2 ? 8 : 5)*(int64_t)Q3];
         s3[i] = (O31*gX) + (O32*gY) + (O33*gZ);
         s3[Q3 + i] = (O21*gX) + (O22*gY) + (O23*gZ);
         s3[2*Q3 + i] = (O11*gX) + (O12*gY) + (O13*gZ);
      }
      else
      {
         double u = 0.0;
         for (int dz = 0; dz < D; dz++) { u += sB[qz*D + dz]*s2[dz*QQ + qq]; }
         s3[i] = u*De[0];
      }
   }
   __syncthreads();
   for (int i = tid; i < DQQ; i += nt)   // back in z: [k][dz][qy][qx]
   {
      const int qq = i % QQ, dz = i/QQ;
      double t1 = 0.0, t2 = 0.0, t3 = 0.0;
      for (int qz = 0; qz < Q; qz++)
      {
         const double bz = sB[qz*D + dz], gz = sG[qz*D + dz];
         if (KIND)
         {
            t3 += gz*s3[qz*QQ + qq];
            t2 += bz*s3[Q3 + qz*QQ + qq];
            t1 += bz*s3[2*Q3 + qz*QQ + qq];
         }
         else { t3 += bz*s3[qz*QQ + qq]; }
      }
      s2[i] = t3;
      if (KIND) { s2[DQQ + i] = t2; s2[2*DQQ + i] = t1; }
   }
   __syncthreads();
   for (int i = tid; i < DDQ; i += nt)   // back in y: [k][dz][dy][qx]
   {
      const int qx = i % Q, dy = (i/Q) % D, dz = i/(Q*D);
      double v1 = 0.0, v23 = 0.0;
      for (int qy = 0; qy < Q; qy++)
      {
         v23 += sB[qy*D + dy]*s2[dz*QQ + qy*Q + qx];
         if (KIND)
         {
            v23 += sG[qy*D + dy]*s2[DQQ + dz*QQ + qy*Q + qx];
            v1 += sB[qy*D + dy]*s2[2*DQQ + dz*QQ + qy*Q + qx];
         }
      }
      s1[i] = v23;
      if (KIND) { s1[DDQ + i] = v1; }
   }
   __syncthreads();
   for (int i = tid; i < D3; i += nt)
   {
      const int dx = i % D, dyz = i/D;
      double y = 0.0;
      for (int qx = 0; qx < Q; qx++)
      {
         y += sB[qx*D + dx]*s1[dyz*Q + qx];
         if (KIND) { y += sG[qx*D + dx]*s1[DDQ + dyz*Q + qx]; }
      }
      const int dy = dyz % D, dz = dyz/D, DI = D - 2;
      if (GATHER && !ACCUM && va.ydirect && dx >= 1 && dx <= DI && dy >= 1 && dy <= DI && dz >= 1 && dz <= DI)
      {
         va.ydirect[(int64_t)blockIdx.x*(DI*DI*DI) + ((dz - 1)*DI + (dy - 1))*DI + (dx - 1)] = y;
      }
      else if (GATHER && !ACCUM && va.ydirect) { yE[e*packed_size(D) + packed_pos(D, dz, dy, dx)] = y; }
      else if (ACCUM) { yE[eblk*D3 + i] += y; }
      else { yE[eblk*D3 + i] = y; }
   }
}

// ------------------------------------------------------------------------------------
// Dispatch: (D1D,Q1D) -> specialised launcher, like MFEM's KernelDispatchTable
// (fem/kernel_dispatch.hpp:131-181).  Registered: D1D = 2..9 with Q1D in {D1D, D1D+1}
// (BP5 collocated GLL and BP1/BP3/ex1 Gauss-Legendre p+2), i.e. p = 1..8.
// ------------------------------------------------------------------------------------
template <class BG>
static void fill_basis(BG &bg, int D, int Q, const double *B, const double *G)
{
   for (int q = 0; q < Q; q++)
   {
      for (int d = 0; d < D; d++)
      {
         bg.b[q*D + d] = B[q + Q*d];
         bg.g[q*D + d] = G ? G[q + Q*d] : 0.0;
      }
   }
}
template <int N>
static void fill_basis(BasisTT<N> &bg, int D, int Q, const double *B, const double *G)
{
   for (int q = 0; q < Q; q++)
   {
      for (int d = 0; d < D; d++)
      {
         bg.b[q*D + d] = bg.bt[d*Q + q] = B[q + Q*d];
         bg.g[q*D + d] = bg.gt[d*Q + q] = G ? G[q + Q*d] : 0.0;
      }
   }
}

// Schedule of the element kernel per (operator, order); tuned on MI355X (profiles/r1).
// MFEM_AMD_APPLY_MODE=0|1|2 overrides it (tuning aid).
static int apply_mode(int KIND, int D, int Q)
{
   static int forced = -2;
   if (forced == -2)
   {
      const char *v = getenv("MFEM_AMD_APPLY_MODE");
      forced = (v && *v >= '0' && *v <= '2') ? (*v - '0') : -1;
   }
   if (forced >= 0) { return forced; }
   // measured (profiles/r1/sweep_modes.txt): one batch per workgroup wins up to p = 7 (BP3);
   // the persistent round-robin schedule wins where registers allow one wave per SIMD only
   // ((6,6) was persistent with streamed pa_data after round 1's isolated sweep; in the alternating apply the plain
   // form is 2 % faster: profiles/r2/ctx_sweep_bp5.txt)
   // ((8,8) too until the re-sweep on the final tree of round 3 -- staged stores, packed E-vector: one batch per workgroup with
   // whole columns is 6 % faster there now, profiles/r3/ctx_sweep_final.txt)
   // ((9,10) too once the one-batch form lost its one-trip loop: whole columns fit 248 VGPRs without the loop-carried
   // addresses -- 258 us against 296 for the persistent streamed form, profiles/r3/ctx_sweep_one_trip.txt; (9,9) stays
   // persistent, 262 against 297 -- until it could contract in place, inplace_for: 206 us in the one-batch form).  Every
   // dispatched kernel is a one-batch-per-workgroup kernel now; the persistent forms stay behind the override.
   return 0;
}

typedef int (*apply_fn)(int64_t ne, const double *B, const double *G, const double *Dq,
                        const double *x, const int *gmap, double *yE, double *ydir, hipStream_t s);

// pa_data prefetch distance in z-slices (0 = hold all Q slices in registers); MFEM_AMD_APPLY_PF
// overrides it (tuning aid; only the values compiled in are accepted).
static int apply_pf(int KIND, int D, int Q)
{
   static int forced = -2;
   if (forced == -2)
   {
      const char *v = getenv("MFEM_AMD_APPLY_PF");
      forced = (v && *v >= '0' && *v <= '9') ? (*v - '0') : -1;
   }
   if (forced >= 0) { return forced; }
   // measured (profiles/r1/sweep_pf_*.txt): streaming pays where the whole-column prefetch costs
   // a wave of occupancy -- diffusion at p = 7, 8 (p = 8: 501 -> 351 us at 1e7 dofs), and the
   // collocated (Q = D) kernels at p = 5, 7, 8; everywhere else holding the column is as fast or faster
   if (KIND == 1)
   {
      // (D = 9 streamed two / three slices through the persistent schedule until the one-trip and in-place forms)
      // (D = 8 streamed two slices until round 3's re-sweep in context: with the packed E-vector and the staged stores the
      // whole-column, non-temporal form is 6-7 % faster at (8,9) and (8,8), profiles/r3/ctx_sweep_final.txt)
      if (D == 5 && Q == 8) { return 2; }        // eight whole z-slices do not fit three waves' registers (40-92 B of scratch)
   }
   return 0;
}

constexpr int rb_table(int KIND, int D, int Q)
{
   if (KIND == 0) { return D >= 4 ? 2 : 1; }
   if (Q == D) { return D == 5 ? 2 : 1; }       // ((6,6), (9,9): level 1 since they can contract in place, inplace_for)
   // ((3,4): level 2 holds 91 VGPRs, five workgroups per CU instead of four: 950 against 964 us, rb_resweep_one_trip.txt)
   return D >= 9 ? 2 : (D >= 4 ? 1 : (D == 3 && Q == 4 ? 2 : 0));
}

// register-blocked contractions (kernel comment): MFEM_AMD_APPLY_RB=0|1|2 overrides (tuning aid)
static int apply_rb(int KIND, int D, int Q)
{
   static int forced = -2;
   if (forced == -2)
   {
      const char *v = getenv("MFEM_AMD_APPLY_RB");
      forced = (v && *v >= '0' && *v <= '2') ? (*v - '0') : -1;
   }
   if (forced >= 0) { return forced; }
   // measured (profiles/r1/sweep_rb_*.txt): level 1 pays from p = 3 on; level 2 adds a little for
   // mass and for the largest elements, where its smaller register footprint matters
   return rb_table(KIND, D, Q);
}

// the non-temporal instantiation for a blocked pa_data of `bytes`: once it cannot stay in the 256 MiB Infinity Cache beside
// the vectors and index tables of the application
static int apply_nt(int64_t bytes)
{
   static int forced = -2;
   if (forced == -2)
   {
      const char *v = getenv("MFEM_AMD_APPLY_NT");
      forced = (v && (*v == '0' || *v == '1')) ? (*v - '0') : -1;
   }
   if (forced >= 0) { return forced; }
   return bytes > ((int64_t)192 << 20) ? 1 : 0;
}

template <int KIND, int D, int Q, bool GATHER, bool ACCUM, bool BLK, int PF = 0, int RB = 0, bool RESOLVED = false>
static int launch_apply(int64_t ne, const double *B, const double *G, const double *Dq,
                        const double *x, const int *gmap, double *yE, double *ydir, hipStream_t s)
{
   constexpr int NBZ = nbz_for(KIND, Q);
   constexpr bool HOT = BLK && GATHER && !ACCUM && Q > 3;
   if (HOT && !RESOLVED)
   {
      // the hot path has streamed and register-blocked variants
      const int pf = apply_pf(KIND, D, Q);
      const int rb = apply_rb(KIND, D, Q);
      // compiled: every RB level with PF = 0, and the table's RB level with PF = 2, 3 (the environment
      // overrides are tuning aids; the other combinations are refused)
#define MFA_VARIANT(P, R) \
      if (pf == P && rb == R) \
      { \
         constexpr bool built = (P == 0) || (R == rb_table(KIND, D, Q)); \
         MFA_REQUIRE(built, "pa_apply: the variant PF=%d RB=%d of (%d,%d) is not compiled in", P, R, D, Q); \
         return launch_apply<KIND, D, Q, GATHER, ACCUM, BLK, (HOT && built) ? P : 0, (HOT && built) ? R : 0, true>(ne, B, G, Dq, x, gmap, yE, ydir, s); \
      }
      MFA_VARIANT(0, 1) MFA_VARIANT(2, 0) MFA_VARIANT(2, 1) MFA_VARIANT(3, 0) MFA_VARIANT(3, 1)
      MFA_VARIANT(0, 2) MFA_VARIANT(2, 2) MFA_VARIANT(3, 2)
#ifdef MFA_DEEP_RING          // tuning builds: deeper rings (the table's RB level only)
      MFA_VARIANT(4, 1) MFA_VARIANT(4, 2)
#endif
#undef MFA_VARIANT
   }
   const int mode = apply_mode(KIND, D, Q);   // 0 one batch per workgroup, 1 persistent/contiguous, 2 persistent/round-robin
   // non-temporal instantiations exist for the blocked-layout gather kernels that hold whole pa_data columns
   constexpr bool NTV = BLK && GATHER && !ACCUM && PF == 0;
   const bool nt = NTV && apply_nt((int64_t)sizeof(double)*blocked_size(KIND, Q, ne));
   auto kern = nt ? k_pa_apply<KIND, D, Q, NBZ, GATHER, ACCUM, BLK, 0, true, PF, RB, false, NTV>
                  : k_pa_apply<KIND, D, Q, NBZ, GATHER, ACCUM, BLK, 0, true, PF, RB>;
   auto kern1 = nt ? k_pa_apply<KIND, D, Q, NBZ, GATHER, ACCUM, BLK, 0, false, PF, RB, false, NTV>
                   : k_pa_apply<KIND, D, Q, NBZ, GATHER, ACCUM, BLK, 0, false, PF, RB>;
   // resident workgroups on the current device (queried once per instantiation and device)
   static int resident[64] = {0};
   int dev = 0;
   if (int rc = check_hip(hipGetDevice(&dev), "hipGetDevice")) { return rc; }
   if (dev < 0 || dev >= 64) { dev = 0; }
   if (resident[dev] == 0)
   {
      int per_cu = 0, cus = 0;
      if (int rc = check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, Q*Q*NBZ, 0),
                             "hipOccupancyMaxActiveBlocksPerMultiprocessor")) { return rc; }
      if (int rc = check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev),
                             "hipDeviceGetAttribute")) { return rc; }
      if (per_cu < 1) { per_cu = 1; }
      resident[dev] = per_cu*cus;
   }
   BasisTT<D*Q> bg;
   fill_basis(bg, D, Q, B, G);
   const int64_t nb = (ne + NBZ - 1)/NBZ;
   if (nb == 0) { return 0; }
   MFA_REQUIRE(!ydir || (GATHER && !ACCUM && D >= 3), "pa_apply: interior rows can only be written by the gather kernels, D1D >= 3");
   VecArgs va{};
   va.ydirect = ydir;
   if (mode == 0)
   {
      MFA_REQUIRE(nb < (int64_t)2147483647, "pa_apply: too many workgroups (%lld)", (long long)nb);
      // Runs of 64 consecutive batches per XCD once the L-vector no longer fits the Infinity Cache (8 B x ~NE (D-1)^3 dofs beyond
      // 192 MiB): neighbouring batches then share their gathered x lines through one L2 instead of each XCD fetching them from
      // HBM -- whole apply at p = 4 +0.6 / +1.9 / +3.9 % at 3e7 / 5e7 / 1e8 dofs, PCG +2.8 / +4.1 / +4.3 % (p = 2, 6 at 1e8: +2.2 / +0.7 %;
      // profiles/r3/xcd_runs_large.txt).  Below that the re-fetched lines are cache hits that cost nothing and the round-robin
      // order is as fast or faster (xcd_runs.txt, xcd_traffic_probe.txt).  MFEM_AMD_APPLY_XCD forces: 0 round-robin, 1
      // XCD-contiguous batches, >= 4 runs of that many batches per XCD.
      static const int xcd_env = getenv("MFEM_AMD_APPLY_XCD") ? atoi(getenv("MFEM_AMD_APPLY_XCD")) : -1;
      const bool big_x = (int64_t)sizeof(double)*ne*(D - 1)*(D - 1)*(D - 1) > ((int64_t)192 << 20);
      const int xcd = xcd_env >= 4 ? xcd_env : xcd_env > 0 ? 2 : xcd_env == 0 ? 1 : (big_x ? 64 : 1);
      kern1<<<dim3((unsigned)nb), dim3(Q, Q, NBZ), 0, s>>>(ne, nb, xcd, bg, Dq, x, gmap, yE, va);
   }
   else
   {
      const unsigned grid = (unsigned)(nb < resident[dev] ? nb : resident[dev]);
      kern<<<dim3(grid), dim3(Q, Q, NBZ), 0, s>>>(ne, nb, mode == 2 ? 1 : 0, bg, Dq, x, gmap, yE, va);
   }
   MFA_CHECK_LAUNCH("k_pa_apply");
   return 0;
}

template <int KIND, bool GATHER, bool ACCUM, bool BLK>
static apply_fn find_apply(int D, int Q)
{
#define MFA_CASE(DD, QQ) \
   if constexpr (MFA_ONLY_D == 0 || DD == MFA_ONLY_D) { if (D == DD && Q == QQ) { return launch_apply<KIND, DD, QQ, GATHER, ACCUM, BLK>; } }
   MFA_CASE(2, 2) MFA_CASE(2, 3) MFA_CASE(3, 3) MFA_CASE(3, 4) MFA_CASE(4, 4) MFA_CASE(4, 5)
   MFA_CASE(5, 5) MFA_CASE(5, 6) MFA_CASE(6, 6) MFA_CASE(6, 7) MFA_CASE(7, 7) MFA_CASE(7, 8)
   MFA_CASE(8, 8) MFA_CASE(8, 9) MFA_CASE(9, 9) MFA_CASE(9, 10)
   // the reference's "others" (bilininteg_diffusion_kernels.cpp:62-63, bilininteg_mass_kernels.cpp:59,66-68):
   // over-integrated rules, e.g. the mass matrix of an order-3 space on an order-3 curved mesh
   MFA_CASE(4, 6) MFA_CASE(5, 8)
   if constexpr (KIND == 0) { MFA_CASE(2, 4) MFA_CASE(3, 6) }
#undef MFA_CASE
   return nullptr;
}

template <int KIND, bool GATHER, bool ACCUM, bool BLK>
static int run_apply(int D, int Q, int64_t ne, const double *B, const double *G,
                     const double *Dq, const double *x, const int *gmap, double *yE,
                     hipStream_t s, double *ydir = nullptr)
{
   MFA_REQUIRE(D >= 2 && D <= MAX_D1D && Q >= 1 && Q <= MAX_Q1D && D*Q <= MAX_DQ,
               "pa_apply: unsupported (D1D,Q1D) = (%d,%d); limits D1D<=%d, Q1D<=%d, D1D*Q1D<=%d",
               D, Q, MAX_D1D, MAX_Q1D, MAX_DQ);
   MFA_REQUIRE(B && Dq && x && yE && (KIND == 0 || G) && (!GATHER || gmap), "pa_apply: null pointer argument");
   if (apply_fn f = find_apply<KIND, GATHER, ACCUM, BLK>(D, Q)) { return f(ne, B, G, Dq, x, gmap, yE, ydir, s); }
   MFA_REQUIRE(!BLK, "pa_apply: the blocked pa_data layout needs a specialised (D1D,Q1D); (%d,%d) is not registered", D, Q);
   report_fallback(KIND ? "PADiffusionApply3D" : "PAMassApply3D", D, Q);
   if (ne == 0) { return 0; }
   MFA_REQUIRE(ne < (int64_t)2147483647, "pa_apply: too many elements for the fallback kernel");
   BasisT<MAX_DQ> bg;
   fill_basis(bg, D, Q, B, G);
   constexpr int NF = KIND ? 3 : 1, NX = KIND ? 2 : 1;
   const size_t lds = sizeof(double)*(size_t)(2*Q*D + D*D*D + NX*D*D*Q + NF*D*Q*Q + NF*Q*Q*Q);
   MFA_REQUIRE(lds <= 160*1024, "pa_apply: fallback kernel needs %zu B of LDS (> 160 KiB)", lds);
   auto kern = k_pa_apply_generic<KIND, GATHER, ACCUM>;
   if (lds > 64*1024)
   {
      if (int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                             "hipFuncSetAttribute")) { return rc; }
   }
   MFA_REQUIRE(!ydir || (GATHER && !ACCUM && D >= 3), "pa_apply: interior rows can only be written by the gather kernels, D1D >= 3");
   VecArgs va{};
   va.ydirect = ydir;
   kern<<<dim3((unsigned)ne), dim3(256), lds, s>>>(D, Q, ne, bg, Dq, x, gmap, yE, va, 0);
   MFA_CHECK_LAUNCH("k_pa_apply_generic");
   return 0;
}

// Non-symmetric D (matrix coefficient, 9 planes): the reference has no specialisation of its own for
// it either (one kernel, `symmetric` a runtime flag); here it runs through the LDS fallback kernel.
template <bool GATHER, bool ACCUM>
static int run_apply_nonsym(int D, int Q, int64_t ne, const double *B, const double *G,
                            const double *Dq, const double *x, const int *gmap, double *yE,
                            hipStream_t s)
{
   MFA_REQUIRE(D >= 2 && D <= MAX_D1D && Q >= 1 && Q <= MAX_Q1D && D*Q <= MAX_DQ,
               "pa_apply: unsupported (D1D,Q1D) = (%d,%d); limits D1D<=%d, Q1D<=%d, D1D*Q1D<=%d",
               D, Q, MAX_D1D, MAX_Q1D, MAX_DQ);
   MFA_REQUIRE(B && G && Dq && x && yE && (!GATHER || gmap), "pa_apply: null pointer argument");
   if (ne == 0) { return 0; }
   MFA_REQUIRE(ne < (int64_t)2147483647, "pa_apply: too many elements for the fallback kernel");
   BasisT<MAX_DQ> bg;
   fill_basis(bg, D, Q, B, G);
   const size_t lds = sizeof(double)*(size_t)(2*Q*D + D*D*D + 2*D*D*Q + 3*D*Q*Q + 3*Q*Q*Q);
   MFA_REQUIRE(lds <= 160*1024, "pa_apply: fallback kernel needs %zu B of LDS (> 160 KiB)", lds);
   auto kern = k_pa_apply_generic<2, GATHER, ACCUM>;
   if (lds > 64*1024)
   {
      if (int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                             "hipFuncSetAttribute")) { return rc; }
   }
   kern<<<dim3((unsigned)ne), dim3(256), lds, s>>>(D, Q, ne, bg, Dq, x, gmap, yE, VecArgs{}, 0);
   MFA_CHECK_LAUNCH("k_pa_apply_generic (non-symmetric D)");
   return 0;
}


// ---- vector spaces: one launch, the workgroup loops over the components -------------------
typedef int (*apply_vec_fn)(int64_t ne, const double *B, const double *G, const double *Dq,
                            const double *x, const int *gmap, double *yE, VecArgs va, hipStream_t s);

template <int KIND, int D, int Q, bool GATHER, bool ACCUM, bool BLK>
static int launch_apply_vec(int64_t ne, const double *B, const double *G, const double *Dq,
                            const double *x, const int *gmap, double *yE, VecArgs va, hipStream_t s)
{
   constexpr int NBZ = nbz_for(KIND, Q);
   constexpr int RBV = Q > 3 ? rb_table(KIND, D, Q) : 0;
   constexpr bool NTV = BLK && GATHER && !ACCUM;
   const bool nt = NTV && apply_nt((int64_t)sizeof(double)*blocked_size(KIND, Q, ne)*(va.d_comp_stride ? va.vd : 1));
   auto kern = nt ? k_pa_apply<KIND, D, Q, NBZ, GATHER, ACCUM, BLK, 0, false, 0, RBV, true, NTV>
                  : k_pa_apply<KIND, D, Q, NBZ, GATHER, ACCUM, BLK, 0, false, 0, RBV, true>;
   BasisTT<D*Q> bg;
   fill_basis(bg, D, Q, B, G);
   const int64_t nb = (ne + NBZ - 1)/NBZ;
   if (nb == 0) { return 0; }
   MFA_REQUIRE(nb < (int64_t)2147483647, "pa_apply: too many workgroups (%lld)", (long long)nb);
   kern<<<dim3((unsigned)nb), dim3(Q, Q, NBZ), 0, s>>>(ne, nb, 1, bg, Dq, x, gmap, yE, va);
   MFA_CHECK_LAUNCH("k_pa_apply (vector)");
   return 0;
}

template <int KIND, bool GATHER, bool ACCUM, bool BLK>
static apply_vec_fn find_apply_vec(int D, int Q)
{
#define MFA_CASE(DD, QQ) \
   if constexpr (MFA_ONLY_D == 0 || DD == MFA_ONLY_D) { if (D == DD && Q == QQ) { return launch_apply_vec<KIND, DD, QQ, GATHER, ACCUM, BLK>; } }
   MFA_CASE(2, 2) MFA_CASE(2, 3) MFA_CASE(3, 3) MFA_CASE(3, 4) MFA_CASE(4, 4) MFA_CASE(4, 5)
   MFA_CASE(5, 5) MFA_CASE(5, 6) MFA_CASE(6, 6) MFA_CASE(6, 7) MFA_CASE(7, 7) MFA_CASE(7, 8)
   MFA_CASE(8, 8) MFA_CASE(8, 9) MFA_CASE(9, 9) MFA_CASE(9, 10)
   // the reference's "others" (bilininteg_diffusion_kernels.cpp:62-63, bilininteg_mass_kernels.cpp:59,66-68):
   // over-integrated rules, e.g. the mass matrix of an order-3 space on an order-3 curved mesh
   MFA_CASE(4, 6) MFA_CASE(5, 8)
   if constexpr (KIND == 0) { MFA_CASE(2, 4) MFA_CASE(3, 6) }
#undef MFA_CASE
   return nullptr;
}

template <int KIND, bool GATHER, bool ACCUM, bool BLK>
static int run_apply_vec(int D, int Q, int64_t ne, const double *B, const double *G,
                         const double *Dq, const double *x, const int *gmap, double *yE,
                         VecArgs va, hipStream_t s)
{
   MFA_REQUIRE(D >= 2 && D <= MAX_D1D && Q >= 1 && Q <= MAX_Q1D && D*Q <= MAX_DQ,
               "pa_apply: unsupported (D1D,Q1D) = (%d,%d); limits D1D<=%d, Q1D<=%d, D1D*Q1D<=%d",
               D, Q, MAX_D1D, MAX_Q1D, MAX_DQ);
   MFA_REQUIRE(B && Dq && x && yE && (KIND == 0 || G) && (!GATHER || gmap), "pa_apply: null pointer argument");
   MFA_REQUIRE(va.vd >= 1 && va.vd <= 16, "pa_apply: vdim must be in 1..16 (got %d)", va.vd);
   if (apply_vec_fn f = find_apply_vec<KIND, GATHER, ACCUM, BLK>(D, Q)) { return f(ne, B, G, Dq, x, gmap, yE, va, s); }
   MFA_REQUIRE(!BLK, "pa_apply: the blocked pa_data layout needs a specialised (D1D,Q1D); (%d,%d) is not registered", D, Q);
   report_fallback(KIND ? "PAVectorDiffusionApply3D" : "PAVectorMassApply3D", D, Q);
   if (ne == 0) { return 0; }
   MFA_REQUIRE(ne < (int64_t)2147483647, "pa_apply: too many elements for the fallback kernel");
   BasisT<MAX_DQ> bg;
   fill_basis(bg, D, Q, B, G);
   constexpr int NF = KIND ? 3 : 1, NX = KIND ? 2 : 1;
   const size_t lds = sizeof(double)*(size_t)(2*Q*D + D*D*D + NX*D*D*Q + NF*D*Q*Q + NF*Q*Q*Q);
   MFA_REQUIRE(lds <= 160*1024, "pa_apply: fallback kernel needs %zu B of LDS (> 160 KiB)", lds);
   auto kern = k_pa_apply_generic<KIND, GATHER, ACCUM>;
   if (lds > 64*1024)
   {
      if (int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                             "hipFuncSetAttribute")) { return rc; }
   }
   for (int c = 0; c < va.vd; c++)
   {
      kern<<<dim3((unsigned)ne), dim3(256), lds, s>>>(D, Q, ne, bg, Dq, x, gmap, yE, va, c);
      MFA_CHECK_LAUNCH("k_pa_apply_generic (vector)");
   }
   return 0;
}

int pa_apply_L2E_vec(int kind, int layout, int D, int Q, int64_t ne, const double *B, const double *G,
                     const double *Dq, const int *gmap, const double *x, double *yE, VecArgs va, hipStream_t s)
{
   MFA_REQUIRE(kind == 0 || kind == 1, "pa_apply (vector space): kind must be 0 (mass) or 1 (diffusion)");
   if (layout)
   {
      if (kind == 0) { return run_apply_vec<0, true, false, true>(D, Q, ne, B, G, Dq, x, gmap, yE, va, s); }
      return run_apply_vec<1, true, false, true>(D, Q, ne, B, G, Dq, x, gmap, yE, va, s);
   }
   if (kind == 0) { return run_apply_vec<0, true, false, false>(D, Q, ne, B, G, Dq, x, gmap, yE, va, s); }
   return run_apply_vec<1, true, false, false>(D, Q, ne, B, G, Dq, x, gmap, yE, va, s);
}

int pa_apply_L2E(int kind, int layout, int D, int Q, int64_t ne, const double *B, const double *G,
                 const double *Dq, const int *gmap, const double *x, double *yE, hipStream_t s, double *ydir)
{
   if (kind == 2)
   {
      MFA_REQUIRE(!ydir, "pa_apply: the non-symmetric D (kind 2) writes the whole E-vector");
      MFA_REQUIRE(layout == 0, "pa_apply: the non-symmetric D (kind 2) has the MFEM layout only");
      return run_apply_nonsym<true, false>(D, Q, ne, B, G, Dq, x, gmap, yE, s);
   }
   if (layout)
   {
      if (kind == 0) { return run_apply<0, true, false, true>(D, Q, ne, B, G, Dq, x, gmap, yE, s, ydir); }
      return run_apply<1, true, false, true>(D, Q, ne, B, G, Dq, x, gmap, yE, s, ydir);
   }
   if (kind == 0) { return run_apply<0, true, false, false>(D, Q, ne, B, G, Dq, x, gmap, yE, s, ydir); }
   return run_apply<1, true, false, false>(D, Q, ne, B, G, Dq, x, gmap, yE, s, ydir);
}

} // namespace mfa

using namespace mfa;

extern "C" int mfem_amd_has_specialization(int kind, int D1D, int Q1D)
{
   if (kind == 0) { return find_apply<0, false, true, false>(D1D, Q1D) != nullptr; }
   return find_apply<1, false, true, false>(D1D, Q1D) != nullptr;
}

extern "C" int mfem_amd_pa_diffusion_apply_3d(int D1D, int Q1D, int64_t ne, int symmetric,
                                              const double *B, const double *G,
                                              const double *D, const double *xE, double *yE,
                                              void *stream)
{
   if (!symmetric) { return run_apply_nonsym<false, true>(D1D, Q1D, ne, B, G, D, xE, nullptr, yE, as_stream(stream)); }
   return run_apply<1, false, true, false>(D1D, Q1D, ne, B, G, D, xE, nullptr, yE, as_stream(stream));
}

extern "C" int mfem_amd_pa_mass_apply_3d(int D1D, int Q1D, int64_t ne, const double *B,
                                         const double *D, const double *xE, double *yE,
                                         void *stream)
{
   return run_apply<0, false, true, false>(D1D, Q1D, ne, B, nullptr, D, xE, nullptr, yE, as_stream(stream));
}

extern "C" int mfem_amd_pa_apply_L2E(int kind, int pa_layout, int D1D, int Q1D, int64_t ne,
                                     const double *B, const double *G, const double *D,
                                     const int *gather_map, const double *x, double *yE,
                                     void *stream)
{
   MFA_REQUIRE(kind >= 0 && kind <= 2, "pa_apply_L2E: kind must be 0 (mass), 1 (diffusion) or 2 (diffusion, non-symmetric D)");
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_apply_L2E: pa_layout must be 0 (MFEM) or 1 (blocked)");
   return pa_apply_L2E(kind, pa_layout, D1D, Q1D, ne, B, G, D, gather_map, x, yE, as_stream(stream), nullptr);
}


extern "C" int mfem_amd_pa_apply_L2E_packed(int kind, int pa_layout, int D1D, int Q1D, int64_t ne,
                                            const double *B, const double *G, const double *D,
                                            const int *gather_map, const double *x, double *yE_packed,
                                            double *y_interior, void *stream)
{
   MFA_REQUIRE(kind == 0 || kind == 1, "pa_apply_L2E_packed: kind must be 0 (mass) or 1 (diffusion)");
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_apply_L2E_packed: pa_layout must be 0 (MFEM) or 1 (blocked)");
   MFA_REQUIRE(D1D >= 3 && y_interior, "pa_apply_L2E_packed: needs D1D >= 3 and the interior rows of y");
   return pa_apply_L2E(kind, pa_layout, D1D, Q1D, ne, B, G, D, gather_map, x, yE_packed, as_stream(stream), y_interior);
}

extern "C" int mfem_amd_pa_apply_L2E_vdim(int kind, int pa_layout, int D1D, int Q1D, int64_t ne,
                                          int vdim, int byvdim, int64_t ndofs,
                                          const double *B, const double *G, const double *D,
                                          int64_t d_comp_stride, const int *gather_map,
                                          const unsigned char *ess_marker, const double *x,
                                          double *yE, void *stream)
{
   MFA_REQUIRE(kind == 0 || kind == 1, "pa_apply_L2E_vdim: kind must be 0 (mass) or 1 (diffusion)");
   MFA_REQUIRE(pa_layout == 0 || pa_layout == 1, "pa_apply_L2E_vdim: pa_layout must be 0 (MFEM) or 1 (blocked)");
   MFA_REQUIRE(ndofs >= 0 && d_comp_stride >= 0, "pa_apply_L2E_vdim: bad sizes");
   VecArgs va = {vdim, byvdim ? 1 : 0, ndofs, ess_marker, d_comp_stride, 0};
   return pa_apply_L2E_vec(kind, pa_layout, D1D, Q1D, ne, B, G, D, gather_map, x, yE, va, as_stream(stream));
}

extern "C" int mfem_amd_pa_vector_diffusion_apply_3d(int D1D, int Q1D, int64_t ne, int vdim, int coeff_vdim,
                                                     const double *B, const double *G, const double *D,
                                                     const double *xE, double *yE, void *stream)
{
   if (coeff_vdim == 0)
   {
      // extension: D is the SCALAR integrator's (Q,Q,Q,6,NE), shared by all components
      VecArgs vs = {vdim, 0, 0, nullptr, 0, 0};
      return run_apply_vec<1, false, true, false>(D1D, Q1D, ne, B, G, D, xE, nullptr, yE, vs, as_stream(stream));
   }
   MFA_REQUIRE(coeff_vdim == 1 || coeff_vdim == vdim,
               "pa_vector_diffusion_apply_3d: scalar and vector coefficients only (coeff_vdim = %d, vdim = %d); "
               "matrix coefficients are not built", coeff_vdim, vdim);
   // the reference stores one block per component for scalar and vector coefficients alike, and
   // allocates dim*dim = 9 planes per block of which the symmetric 6 are filled and read
   // (bilininteg_vecdiffusion_pa.cpp:133-134, 228-290)
   VecArgs va = {vdim, 0, 0, nullptr, 1, 9*(int64_t)Q1D*Q1D*Q1D};
   return run_apply_vec<1, false, true, false>(D1D, Q1D, ne, B, G, D, xE, nullptr, yE, va, as_stream(stream));
}

extern "C" int mfem_amd_pa_vector_mass_apply_3d(int D1D, int Q1D, int64_t ne, int vdim, int coeff_vdim,
                                                const double *B, const double *D, const double *xE,
                                                double *yE, void *stream)
{
   MFA_REQUIRE(coeff_vdim == 1 || coeff_vdim == vdim,
               "pa_vector_mass_apply_3d: scalar and vector coefficients only (coeff_vdim = %d, vdim = %d); "
               "matrix coefficients are not built", coeff_vdim, vdim);
   VecArgs va = {vdim, 0, 0, nullptr, coeff_vdim == 1 ? 0 : 1, 0};   // (Q^3,coeff_vdim,NE)
   return run_apply_vec<0, false, true, false>(D1D, Q1D, ne, B, nullptr, D, xE, nullptr, yE, va, as_stream(stream));
}

extern "C" int64_t mfem_amd_pa_data_size(int kind, int pa_layout, int Q1D, int64_t ne)
{
   if (kind == 2) { return pa_layout ? -1 : ne*(int64_t)9*Q1D*Q1D*Q1D; }
   if (pa_layout) { return blocked_size(kind, Q1D, ne); }
   return ne*(int64_t)(kind ? 6 : 1)*Q1D*Q1D*Q1D;
}

// bench_ceed_p -- the CEED bake-off problems of bench_ceed on the PARALLEL classes (ParMesh, ParFiniteElementSpace,
// ParBilinearForm, CGSolver(MPI_Comm)): one process per GPU, the mesh partitioned in contiguous chunks of its element order,
// scalar (BP1/3/5) and vector (BP2/4/6, vdim = 3, byVDIM as tests/benchmarks/bench_ceed.cpp:47) spaces.  Prints the same
// known-answer quantities as bench_ceed -- iterations, |X| over the owned dofs -- so that any number of ranks can be
// compared with the serial driver.
//   python -m torch.distributed.run --nproc-per-node 2 --no-python examples/bin/bench_ceed_p -bp 4 -p 2 -n 6
#include "mfem.hpp"
#include <chrono>
#include <cstdio>
using namespace mfem;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char *argv[])
{
   Mpi::Init();
   Hypre::Init();
   int bp = 3, p = 4, N = 12, max_it = 2000, reps = 20;
   double rtol = 1e-12;
   const char *device_config = "hip";
   OptionsParser args(argc, argv);
   args.AddOption(&bp, "-bp", "--bake-off-problem", "1..6 (even: the vector versions).");
   args.AddOption(&p, "-p", "--order", "Polynomial degree.");
   args.AddOption(&N, "-n", "--elements", "Elements per direction of the global mesh.");
   args.AddOption(&reps, "-reps", "--repetitions", "Timed operator applications.");
   args.AddOption(&max_it, "-mi", "--max-iter", "CG iteration limit.");
   args.AddOption(&rtol, "-rtol", "--rel-tol", "CG relative tolerance.");
   args.AddOption(&device_config, "-d", "--device", "Device configuration string.");
   args.Parse();
   if (!args.Good()) { if (Mpi::Root()) { args.PrintUsage(std::cout); } return 1; }
   MFEM_VERIFY(bp >= 1 && bp <= 6, "bp must be 1..6");
   Device device(device_config);
   const int me = Mpi::WorldRank(), nranks = Mpi::WorldSize();
   const bool GLL = bp >= 5, mass = bp <= 2;
   const int VDIM = (bp % 2 == 0) ? 3 : 1;
   // set-up times of this rank: the serial mesh (every rank builds it, as the reference's ex1p reads it), the rank's part,
   // the space with its shared-dof tables (work of the rank's part and its surface, not of the global problem)
   double ts = now();
   Mesh mesh = Mesh::MakeCartesian3D(N, N, N, Element::HEXAHEDRON, 1.0, 1.0, 1.0, true);
   const double t_mesh = now() - ts; ts = now();
   ParMesh pmesh(MPI_COMM_WORLD, mesh);
   const double t_pmesh = now() - ts; ts = now();
   H1_FECollection fec(p, 3, BasisType::GaussLobatto);
   ParFiniteElementSpace fes(&pmesh, &fec, VDIM, VDIM == 3 ? Ordering::byVDIM : Ordering::byNODES);
   const double t_space = now() - ts;
   IntegrationRules irs(0, GLL ? Quadrature1D::GaussLobatto : Quadrature1D::GaussLegendre);
   const IntegrationRule *ir = &irs.Get(Geometry::CUBE, 2*p + (GLL ? -1 : 3));
   ConstantCoefficient one(1.0);
   ParBilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   if (VDIM == 1 && mass) { a.AddDomainIntegrator(new MassIntegrator(one, ir)); }
   else if (VDIM == 1) { a.AddDomainIntegrator(new DiffusionIntegrator(one, ir)); }
   else if (mass) { a.AddDomainIntegrator(new VectorMassIntegrator(one, ir)); }
   else { a.AddDomainIntegrator(new VectorDiffusionIntegrator(one, ir)); }
   a.Assemble();

   Array<int> ess_bdr(pmesh.SerialMesh().bdr_attributes.Max()), ess_tdof_list;
   ess_bdr = 1;
   fes.GetEssentialTrueDofs(ess_bdr, ess_tdof_list);
   ParLinearForm b(&fes);
   Vector uvec(3); uvec = 1.0; uvec /= uvec.Norml2();
   VectorConstantCoefficient unit_vec(uvec);
   if (VDIM == 1) { b.AddDomainIntegrator(new DomainLFIntegrator(one)); }
   else { b.AddDomainIntegrator(new VectorDomainLFIntegrator(unit_vec)); }
   b.UseFastAssembly(true);
   b.Assemble();
   ParGridFunction x(&fes);
   x = 0.0;
   OperatorPtr A;
   Vector B, X;
   a.FormLinearSystem(ess_tdof_list, x, b, A, X, B);
   OperatorJacobiSmoother M(a, ess_tdof_list);
   CGSolver cg(MPI_COMM_WORLD);
   cg.SetRelTol(rtol); cg.SetAbsTol(0.0); cg.SetMaxIter(max_it); cg.SetPrintLevel(-1);
   cg.SetOperator(*A);
   cg.SetPreconditioner(M);
   cg.iterative_mode = false;
   cg.Mult(B, X);
   internal::DeviceSync();
   double t0 = now();
   cg.Mult(B, X);
   internal::DeviceSync();
   const double t_solve = now() - t0;
   // the unconstrained P^T A P (BilinearForm::Mult), timed
   Vector y(X.Size());
   a.Mult(X, y);
   internal::DeviceSync();
   t0 = now();
   for (int i = 0; i < reps; i++) { a.Mult(X, y); }
   internal::DeviceSync();
   const double t_apply = (now() - t0)/reps;
   // norms over the owned dofs, summed over the ranks
   Vector xo(X), yo(y);
   const Array<int> &gh = fes.GhostDofs();
   if (gh.Size()) { xo.SetSubVector(gh, 0.0); yo.SetSubVector(gh, 0.0); }
   double s[2] = { xo*xo, yo*yo };
   mfem_amd_world_allreduce_host(Mpi::World(), s, 2, 0);
   if (me == 0)
   {
      printf("{\"phase\": \"problem\", \"ranks\": %d, \"bp\": %d, \"p\": %d, \"N\": %d, \"dofs\": %lld, \"iterations\": %d, \"converged\": %d, "
             "\"norm_X\": %.15e, \"norm_AX\": %.15e, \"solve_s\": %.6f, \"ms_per_apply\": %.6f, "
             "\"serial_mesh_s\": %.3f, \"par_mesh_s\": %.3f, \"par_space_s\": %.3f, \"local_dofs\": %d}\n", nranks, bp, p, N,
             (long long)fes.GlobalTrueVSize(), cg.GetNumIterations(), (int)cg.GetConverged(), std::sqrt(s[0]), std::sqrt(s[1]),
             t_solve, t_apply*1e3, t_mesh, t_pmesh, t_space, fes.GetVSize());
      fflush(stdout);
   }
   return 0;
}

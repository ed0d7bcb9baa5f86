// ex1p -- the parallel Poisson example (-Delta u = 1, u = 0 on the boundary) through the parallel layer of
// mfem_amd/host/mfem.hpp: one process per MI355X, ParMesh / ParFiniteElementSpace / ParBilinearForm with partial
// assembly, Jacobi-preconditioned CGSolver(MPI_COMM_WORLD) whose dot products and shared-dof exchanges run on RCCL.
// Command line of MFEM's examples/ex1p.cpp (plus -r, -rp, -save); launch one process per GPU, e.g.
//
//   python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --no-python
//          examples/bin/ex1p -m tests/golden/meshes/inline-hex.mesh -o 3 -pa -d hip -no-vis
//
// (any launcher that sets RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT works; without them: one rank).
// Written for this repository; it makes the reference example's API calls in the same order.
#include "mfem.hpp"
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>

using namespace mfem;

namespace
{
struct Options
{
   const char *mesh_path = "../data/star.mesh";
   const char *device = "cpu";
   int degree = 1, serial_levels = -1, parallel_levels = 2;
   bool sc = false, partial = false, full = false, glvis = true, write_files = true;

   bool Read(int argc, char **argv, bool root)
   {
      OptionsParser cli(argc, argv);
      cli.AddOption(&mesh_path, "-m", "--mesh", "Hexahedral mesh file (MFEM mesh v1.0).");
      cli.AddOption(&degree, "-o", "--order", "Polynomial degree of the H1 space; <= 0: the degree of the mesh nodes.");
      cli.AddOption(&sc, "-sc", "--static-condensation", "-no-sc", "--no-static-condensation", "Static condensation (not built).");
      cli.AddOption(&partial, "-pa", "--partial-assembly", "-no-pa", "--no-partial-assembly", "Partial assembly (the path built here).");
      cli.AddOption(&full, "-fa", "--full-assembly", "-no-fa", "--no-full-assembly", "Full assembly (not built).");
      cli.AddOption(&device, "-d", "--device", "Device string: hip.");
      cli.AddOption(&glvis, "-vis", "--visualization", "-no-vis", "--no-visualization", "Send the solution to a GLVis server.");
      cli.AddOption(&serial_levels, "-r", "--refine-serial", "Serial refinement levels (-1: up to 10000 elements).");
      cli.AddOption(&parallel_levels, "-rp", "--refine-parallel", "Parallel refinement levels.");
      cli.AddOption(&write_files, "-save", "--save-files", "-no-save", "--no-save-files", "Write mesh.<rank> and sol.<rank>.");
      cli.Parse();
      if (!cli.Good()) { if (root) { cli.PrintUsage(std::cout); } return false; }
      if (root) { cli.PrintOptions(std::cout); }
      return true;
   }
};
} // namespace

int main(int argc, char *argv[])
{
   Mpi::Init(argc, argv);
   const int nranks = Mpi::WorldSize(), me = Mpi::WorldRank();
   Hypre::Init();

   Options opt;
   if (!opt.Read(argc, argv, me == 0)) { return 1; }
   Device device(opt.device);
   if (me == 0) { device.Print(); }

   // the serial mesh on every rank, refined until it has ~10 000 elements; then partitioned and refined in parallel
   Mesh mesh(opt.mesh_path, 1, 1);
   const int dim = mesh.Dimension();
   int levels = opt.serial_levels;
   if (levels < 0) { levels = (int)floor(log(10000./mesh.GetNE())/log(2.)/dim); }
   while (levels-- > 0) { mesh.UniformRefinement(); }

   ParMesh pmesh(MPI_COMM_WORLD, mesh);
   mesh.Clear();
   for (int l = 0; l < opt.parallel_levels; l++) { pmesh.UniformRefinement(); }

   std::unique_ptr<FiniteElementCollection> owned;
   FiniteElementCollection *coll = nullptr;
   if (opt.degree <= 0 && pmesh.GetNodes())
   {
      coll = pmesh.GetNodes()->OwnFEC();
      if (me == 0) { std::cout << "Using isoparametric FEs: " << coll->Name() << std::endl; }
   }
   else
   {
      owned.reset(new H1_FECollection(opt.degree > 0 ? opt.degree : 1, dim));
      coll = owned.get();
   }
   ParFiniteElementSpace V(&pmesh, coll);
   const HYPRE_BigInt size = V.GlobalTrueVSize();
   if (me == 0) { std::cout << "Number of finite element unknowns: " << size << "  (ranks = " << nranks << ")" << std::endl; }

   // homogeneous Dirichlet conditions on the whole (external) boundary
   Array<int> dirichlet;
   {
      const Mesh &sm = pmesh.SerialMesh();
      if (sm.bdr_attributes.Size())
      {
         Array<int> on_boundary(sm.bdr_attributes.Max());
         on_boundary = 0;
         sm.MarkExternalBoundaries(on_boundary);
         V.GetEssentialTrueDofs(on_boundary, dirichlet);
      }
   }

   ConstantCoefficient unit(1.0);
   ParLinearForm rhs(&V);
   rhs.AddDomainIntegrator(new DomainLFIntegrator(unit));
   rhs.Assemble();

   ParGridFunction u(&V);
   u = 0.0;

   ParBilinearForm stiffness(&V);
   if (opt.partial) { stiffness.SetAssemblyLevel(AssemblyLevel::PARTIAL); }
   if (opt.full) { stiffness.SetAssemblyLevel(AssemblyLevel::FULL); }
   stiffness.AddDomainIntegrator(new DiffusionIntegrator(unit));
   if (opt.sc) { stiffness.EnableStaticCondensation(); }
   stiffness.Assemble();

   OperatorPtr A;
   Vector B, X;
   stiffness.FormLinearSystem(dirichlet, u, rhs, A, X, B);

   // Jacobi-preconditioned CG as in the reference example: relative tolerance 1e-12 on sqrt((B r, r)), 2000 iterations
   OperatorJacobiSmoother jacobi(stiffness, dirichlet);
   CGSolver cg(MPI_COMM_WORLD);
   cg.SetRelTol(1e-12);
   cg.SetMaxIter(2000);
   cg.SetPrintLevel(1);
   cg.SetPreconditioner(jacobi);
   cg.SetOperator(*A);
   cg.Mult(B, X);

   stiffness.RecoverFEMSolution(X, rhs, u);
   // |u| over the owned dofs, summed over the ranks (the solution's known-answer norm)
   {
      Vector owned_u(u);
      const Array<int> &gh = V.GhostDofs();
      if (gh.Size()) { owned_u.SetSubVector(gh, 0.0); }
      double s[1] = { owned_u*owned_u };
      mfem_amd_world_allreduce_host(Mpi::World(), s, 1, 0);
      if (me == 0) { std::cout << std::setprecision(15) << "Solution norm: |x|_2 = " << std::sqrt(s[0]) << std::endl; }
   }
   if (opt.write_files)
   {
      std::ostringstream mname, sname;
      mname << "mesh." << std::setfill('0') << std::setw(6) << me;
      sname << "sol." << std::setfill('0') << std::setw(6) << me;
      std::ofstream fm(mname.str()), fu(sname.str());
      fm.precision(8); fu.precision(8);
      pmesh.Print(fm);
      u.Save(fu);
   }
   if (opt.glvis)
   {
      char host[] = "localhost";
      socketstream glvis(host, 19916);
      glvis << "parallel " << nranks << " " << me << "\n";
      glvis.precision(8);
      glvis << "solution\n" << pmesh << u << std::flush;
   }
   return 0;
}

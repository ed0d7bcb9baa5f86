// ex1 -- Poisson problem -Delta u = 1, u = 0 on the boundary, on a hexahedral mesh, through
// the MFEM API surface of mfem_amd (mfem_amd/host/mfem.hpp): H1 space, partial assembly,
// Jacobi-preconditioned CG on the MI355X.  Same command line as MFEM's examples/ex1.cpp; only
// the partial-assembly path exists here, so run it as
//
//     ex1 -m tests/golden/meshes/inline-hex.mesh -o 3 -pa -d hip -no-vis
//
// Written for this repository (it is not the reference's file): the same sequence of API
// calls as the reference's example, so the two can be read side by side.
#include "mfem.hpp"
#include <fstream>
#include <iostream>

using namespace mfem;

int main(int argc, char *argv[])
{
   const char *mesh_file = "../data/star.mesh";
   int order = 1;
   bool static_cond = false, pa = false, fa = false, visualization = true;
   const char *device_config = "cpu";
   int ref_override = -1;
   bool save = true;

   OptionsParser args(argc, argv);
   args.AddOption(&mesh_file, "-m", "--mesh", "Mesh file to use.");
   args.AddOption(&order, "-o", "--order", "Finite element order (polynomial degree) or -1 for isoparametric space.");
   args.AddOption(&static_cond, "-sc", "--static-condensation", "-no-sc", "--no-static-condensation", "Enable static condensation.");
   args.AddOption(&pa, "-pa", "--partial-assembly", "-no-pa", "--no-partial-assembly", "Enable Partial Assembly.");
   args.AddOption(&fa, "-fa", "--full-assembly", "-no-fa", "--no-full-assembly", "Enable Full Assembly.");
   args.AddOption(&device_config, "-d", "--device", "Device configuration string, see Device::Configure().");
   args.AddOption(&visualization, "-vis", "--visualization", "-no-vis", "--no-visualization", "Enable or disable GLVis visualization.");
   args.AddOption(&ref_override, "-r", "--refine", "Number of uniform refinements (-1: as ex1, up to 50000 elements).");
   args.AddOption(&save, "-save", "--save-files", "-no-save", "--no-save-files", "Write refined.mesh and sol.gf.");
   args.Parse();
   if (!args.Good()) { args.PrintUsage(std::cout); return 1; }
   args.PrintOptions(std::cout);

   Device device(device_config);
   device.Print();

   Mesh mesh(mesh_file, 1, 1);
   const int dim = mesh.Dimension();
   {
      const int ref_levels = ref_override >= 0 ? ref_override : (int)floor(log(50000./mesh.GetNE())/log(2.)/dim);
      for (int l = 0; l < ref_levels; l++) { mesh.UniformRefinement(); }
   }

   FiniteElementCollection *fec;
   bool delete_fec = true;
   if (order > 0) { fec = new H1_FECollection(order, dim); }
   else if (mesh.GetNodes()) { fec = mesh.GetNodes()->OwnFEC(); delete_fec = false; std::cout << "Using isoparametric FEs: " << fec->Name() << std::endl; }
   else { fec = new H1_FECollection(order = 1, dim); }
   FiniteElementSpace fespace(&mesh, fec);
   std::cout << "Number of finite element unknowns: " << fespace.GetTrueVSize() << std::endl;

   Array<int> ess_tdof_list;
   if (mesh.bdr_attributes.Size())
   {
      Array<int> ess_bdr(mesh.bdr_attributes.Max());
      ess_bdr = 0;
      mesh.MarkExternalBoundaries(ess_bdr);
      fespace.GetEssentialTrueDofs(ess_bdr, ess_tdof_list);
   }

   LinearForm b(&fespace);
   ConstantCoefficient one(1.0);
   b.AddDomainIntegrator(new DomainLFIntegrator(one));
   b.Assemble();

   GridFunction x(&fespace);
   x = 0.0;

   BilinearForm a(&fespace);
   if (pa) { a.SetAssemblyLevel(AssemblyLevel::PARTIAL); }
   if (fa) { a.SetAssemblyLevel(AssemblyLevel::FULL); }
   a.AddDomainIntegrator(new DiffusionIntegrator(one));
   if (static_cond) { a.EnableStaticCondensation(); }
   a.Assemble();

   OperatorPtr A;
   Vector B, X;
   a.FormLinearSystem(ess_tdof_list, x, b, A, X, B);
   std::cout << "Size of linear system: " << A->Height() << std::endl;

   OperatorJacobiSmoother M(a, ess_tdof_list);
   PCG(*A, M, B, X, 1, 400, 1e-12, 0.0);

   a.RecoverFEMSolution(X, b, x);
   std::cout.precision(15);
   std::cout << "Solution norms: |x|_2 = " << x.Norml2() << "  |x|_inf = " << x.Normlinf() << std::endl;

   if (save)
   {
      std::ofstream mesh_ofs("refined.mesh");
      mesh_ofs.precision(8);
      mesh.Print(mesh_ofs);
      std::ofstream sol_ofs("sol.gf");
      sol_ofs.precision(8);
      x.Save(sol_ofs);
   }
   if (visualization)
   {
      char vishost[] = "localhost";
      socketstream sol_sock(vishost, 19916);
      sol_sock.precision(8);
      sol_sock << "solution\n" << mesh << x << std::flush;
   }
   if (delete_fec) { delete fec; }
   return 0;
}

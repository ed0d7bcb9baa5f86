// ex1 -- Poisson problem -Delta u = 1, u = 0 on the boundary, on a hexahedral mesh, through
// the MFEM API surface of mfem_amd (mfem_amd/host/mfem.hpp): H1 space, partial assembly,
// Jacobi-preconditioned CG on the MI355X.  It takes the command line of MFEM's examples/ex1.cpp
// (plus -r and -save); only the partial-assembly path exists here, so run it as
//
//     ex1 -m tests/golden/meshes/inline-hex.mesh -o 3 -pa -d hip -no-vis
//
// Written for this repository (it is not the reference's file); it makes the API calls of the
// reference's example in the same order, which is what pins the unknown and iteration counts
// of BASELINE.md 2.1 (tests/test_gpu_cxx.py).
#include "mfem.hpp"
#include <fstream>
#include <iostream>
#include <memory>

using namespace mfem;

namespace
{
struct Options
{
   const char *mesh_path = "../data/star.mesh";
   const char *device = "cpu";
   int degree = 1, levels = -1;
   bool sc = false, partial = false, full = false, glvis = true, write_files = true;

   bool Read(int argc, char **argv)
   {
      OptionsParser cli(argc, argv);
      cli.AddOption(&mesh_path, "-m", "--mesh", "Hexahedral mesh file (MFEM mesh v1.0).");
      cli.AddOption(&degree, "-o", "--order", "Polynomial degree of the H1 space; <= 0: the degree of the mesh nodes.");
      cli.AddOption(&sc, "-sc", "--static-condensation", "-no-sc", "--no-static-condensation", "Static condensation (not built).");
      cli.AddOption(&partial, "-pa", "--partial-assembly", "-no-pa", "--no-partial-assembly", "Partial assembly (the path built here).");
      cli.AddOption(&full, "-fa", "--full-assembly", "-no-fa", "--no-full-assembly", "Full assembly (not built).");
      cli.AddOption(&device, "-d", "--device", "Device string: hip.");
      cli.AddOption(&glvis, "-vis", "--visualization", "-no-vis", "--no-visualization", "Send the solution to a GLVis server.");
      cli.AddOption(&levels, "-r", "--refine", "Uniform refinement levels (-1: refine up to 50000 elements).");
      cli.AddOption(&write_files, "-save", "--save-files", "-no-save", "--no-save-files", "Write refined.mesh and sol.gf.");
      cli.Parse();
      if (!cli.Good()) { cli.PrintUsage(std::cout); return false; }
      cli.PrintOptions(std::cout);
      return true;
   }
};

// refine uniformly while the mesh stays below 50 000 elements (or `levels` times)
void Refine(Mesh &mesh, int levels)
{
   if (levels < 0) { levels = (int)floor(log(50000./mesh.GetNE())/log(2.)/mesh.Dimension()); }
   while (levels-- > 0) { mesh.UniformRefinement(); }
}

void Export(const Options &opt, Mesh &mesh, GridFunction &u)
{
   if (opt.write_files)
   {
      std::ofstream fm("refined.mesh"), fu("sol.gf");
      fm.precision(8); fu.precision(8);
      mesh.Print(fm);
      u.Save(fu);
   }
   if (opt.glvis)
   {
      char host[] = "localhost";
      socketstream glvis(host, 19916);
      glvis.precision(8);
      glvis << "solution\n" << mesh << u << std::flush;
   }
}
} // namespace

int main(int argc, char *argv[])
{
   Options opt;
   if (!opt.Read(argc, argv)) { return 1; }
   Device device(opt.device);
   device.Print();

   Mesh mesh(opt.mesh_path, 1, 1);
   Refine(mesh, opt.levels);

   // H1 space of the requested degree, or the nodal space's own collection (isoparametric)
   std::unique_ptr<FiniteElementCollection> owned;
   FiniteElementCollection *coll = nullptr;
   if (opt.degree <= 0 && mesh.GetNodes())
   {
      coll = mesh.GetNodes()->OwnFEC();
      std::cout << "Using isoparametric FEs: " << coll->Name() << std::endl;
   }
   else
   {
      owned.reset(new H1_FECollection(opt.degree > 0 ? opt.degree : 1, mesh.Dimension()));
      coll = owned.get();
   }
   FiniteElementSpace V(&mesh, coll);
   std::cout << "Number of finite element unknowns: " << V.GetTrueVSize() << std::endl;

   // homogeneous Dirichlet conditions on the whole (external) boundary
   Array<int> dirichlet;
   if (mesh.bdr_attributes.Size())
   {
      Array<int> on_boundary(mesh.bdr_attributes.Max());
      on_boundary = 0;
      mesh.MarkExternalBoundaries(on_boundary);
      V.GetEssentialTrueDofs(on_boundary, dirichlet);
   }

   ConstantCoefficient unit(1.0);
   LinearForm rhs(&V);
   rhs.AddDomainIntegrator(new DomainLFIntegrator(unit));
   rhs.Assemble();

   GridFunction u(&V);
   u = 0.0;

   BilinearForm stiffness(&V);
   if (opt.partial) { stiffness.SetAssemblyLevel(AssemblyLevel::PARTIAL); }
   if (opt.full) { stiffness.SetAssemblyLevel(AssemblyLevel::FULL); }
   stiffness.AddDomainIntegrator(new DiffusionIntegrator(unit));
   if (opt.sc) { stiffness.EnableStaticCondensation(); }
   stiffness.Assemble();

   OperatorPtr A;
   Vector B, X;
   stiffness.FormLinearSystem(dirichlet, u, rhs, A, X, B);
   std::cout << "Size of linear system: " << A->Height() << std::endl;

   // Jacobi-preconditioned CG, relative tolerance sqrt(1e-12), at most 400 iterations
   OperatorJacobiSmoother jacobi(stiffness, dirichlet);
   PCG(*A, jacobi, B, X, 1, 400, 1e-12, 0.0);

   stiffness.RecoverFEMSolution(X, rhs, u);
   std::cout.precision(15);
   std::cout << "Solution norms: |x|_2 = " << u.Norml2() << "  |x|_inf = " << u.Normlinf() << std::endl;
   Export(opt, mesh, u);
   return 0;
}

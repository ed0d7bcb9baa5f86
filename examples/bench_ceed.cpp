// bench_ceed -- the CEED bake-off problems (BP1 mass, BP3 / BP5 diffusion; scalar) and
// kernels through the MFEM API surface of mfem_amd, set up as MFEM's
// tests/benchmarks/bench_ceed.cpp does (Cartesian N^3 hex mesh with the default Hilbert
// element ordering, H1 Gauss-Lobatto basis of order p, Gauss-Legendre 2p+3 or Gauss-Lobatto
// 2p-1 rule, unit coefficient, Dirichlet on the whole boundary, f = 1).
//
//   bench_ceed -bp 3 -p 4 -n 12                 # operator apply ("kernel") and CG solve ("problem")
//
// Prints one line per phase with the known-answer quantities (|A x|, iterations, |X|) that
// BASELINE.md tabulates for the reference, plus MDof/s.
#include "mfem.hpp"
#include <chrono>
#include <cstdio>

using namespace mfem;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int reps = 20, max_it = 2000, seed = 1;
static double rtol = 1e-12;
static bool jacobi = true, lex = false, varcoef = false, check = false;
static const char *mesh_file = "";
static int ref_levels = 0;

static void run_case(int bp, int p, int N);

int main(int argc, char *argv[])
{
   int bp = 3, p = 4, N = 12;
   const char *device_config = "hip";
   OptionsParser args(argc, argv);
   args.AddOption(&bp, "-bp", "--bake-off-problem", "1 (mass), 3 (diffusion, q=p+2), 5 (diffusion, GLL q=p+1); 2, 4, 6: the vector (vdim = 3) versions.");
   args.AddOption(&p, "-p", "--order", "Polynomial degree (0: sweep p = 1..8).");
   args.AddOption(&N, "-n", "--elements", "Elements per direction (0: 48/p, the known-answer table of BASELINE.md).");
   args.AddOption(&reps, "-reps", "--repetitions", "Timed operator applications.");
   args.AddOption(&max_it, "-mi", "--max-iter", "CG iteration limit.");
   args.AddOption(&rtol, "-rtol", "--rel-tol", "CG relative tolerance.");
   args.AddOption(&jacobi, "-jacobi", "--jacobi", "-no-jacobi", "--no-jacobi", "Jacobi preconditioner.");
   args.AddOption(&lex, "-lex", "--lexicographic", "-sfc", "--space-filling-curve", "Element ordering of the mesh.");
   args.AddOption(&seed, "-s", "--seed", "Seed of x.Randomize().");
   args.AddOption(&device_config, "-d", "--device", "Device configuration string.");
   args.AddOption(&mesh_file, "-m", "--mesh", "Hexahedral mesh file instead of the Cartesian box (curved meshes: BASELINE config 4).");
   args.AddOption(&ref_levels, "-r", "--refine", "Uniform refinement levels of the mesh file.");
   args.AddOption(&check, "-check", "--property-checks", "-no-check", "--no-property-checks",
                  "Size-independent properties of the operator in the kernel line: symmetry, action on constants.");
   args.AddOption(&varcoef, "-vc", "--variable-coefficient", "-cc", "--constant-coefficient",
                  "Coefficient 1 + x^2 + y z at the quadrature points instead of 1 (scalar problems).");
   args.Parse();
   if (!args.Good()) { args.PrintUsage(std::cout); return 1; }
   MFEM_VERIFY(bp >= 1 && bp <= 6, "bp must be 1..6");
   Device device(device_config);
   for (int q = (p ? p : 1); q <= (p ? p : 8); q++) { run_case(bp, q, N ? N : 48/q); }
   return 0;
}

static void run_case(int bp, int p, int N)
{
   // bench_ceed.cpp:141-156: BP1/2 mass, BP3/4 diffusion q=p+2, BP5/6 diffusion GLL q=p+1; even = VDIM 3, byVDIM (:47)
   const bool GLL = bp >= 5, mass = bp <= 2;
   const int VDIM = (bp % 2 == 0) ? 3 : 1;
   double t0 = now();
   const bool from_file = mesh_file[0] != 0;
   Mesh mesh = from_file ? Mesh(mesh_file, 1, 1) : Mesh::MakeCartesian3D(N, N, N, Element::HEXAHEDRON, 1.0, 1.0, 1.0, !lex);
   for (int l = 0; from_file && l < ref_levels; l++) { mesh.UniformRefinement(); }
   H1_FECollection fec(p, 3, BasisType::GaussLobatto);
   FiniteElementSpace fes(&mesh, &fec, VDIM, VDIM == 3 ? Ordering::byVDIM : Ordering::byNODES);
   IntegrationRules irs(0, GLL ? Quadrature1D::GaussLobatto : Quadrature1D::GaussLegendre);
   const IntegrationRule *ir = &irs.Get(Geometry::CUBE, 2*p + (GLL ? -1 : 3));
   ConstantCoefficient one(1.0);
   FunctionCoefficient var([](const Vector &x) { return 1.0 + x(0)*x(0) + x(1)*x(2); });
   MFEM_VERIFY(!varcoef || VDIM == 1, "the variable coefficient is wired for the scalar problems");
   Coefficient &coef = varcoef ? (Coefficient&)var : (Coefficient&)one;
   const int dofs = fes.GetTrueVSize();
   GridFunction x(&fes), y(&fes);
   BilinearForm a(&fes);
   a.SetAssemblyLevel(AssemblyLevel::PARTIAL);
   if (VDIM == 1 && mass) { a.AddDomainIntegrator(new MassIntegrator(coef, ir)); }
   else if (VDIM == 1) { a.AddDomainIntegrator(new DiffusionIntegrator(coef, ir)); }
   else if (mass) { a.AddDomainIntegrator(new VectorMassIntegrator(one, ir)); }
   else { a.AddDomainIntegrator(new VectorDiffusionIntegrator(one, ir)); }
   a.Assemble();
   double t_setup = now() - t0;

   // kernel: y = A x (BilinearForm::Mult), x = Randomize(seed)
   x.Randomize(seed);
   a.Mult(x, y);
   const double ynorm = y.Norml2();
   x.Read(); y.Write();
   // the device needs 25-35 ms of continuous work after any idle period to reach its steady rate
   // (profiles/r2/warmup_probe.txt): ~0.1 s of the same applies, untimed, before the timed ones
   {
      const double D1w = p + 1, Q1w = ir->n1d;
      const double bw = (double)mesh.GetNE()*((mass ? 1 : 6)*Q1w*Q1w*Q1w*8 + 2*D1w*D1w*D1w*4) + 20.0*dofs;
      const int spin = (int)std::min(4000.0, std::max(10.0, std::ceil(0.1/(bw/3.0e12))));
      for (int i = 0; i < spin; i++) { a.Mult(x, y); }
   }
   internal::DeviceSync();
   t0 = now();
   for (int i = 0; i < reps; i++) { a.Mult(x, y); }
   internal::DeviceSync();
   const double t_apply = (now() - t0)/reps;
   // algorithmic bytes of one L->L apply: NE*(C*Q^3*8 + 2*D^3*4) + N_L*20
   const double D1 = p + 1, Q1 = ir->n1d, NE = (double)mesh.GetNE();
   if (from_file) { N = 0; }
   // vector spaces: q-data, gather_map and indices once per element; x, y per vdof; offsets per scalar dof
   const double bytes_apply = NE*((mass ? 1 : 6)*Q1*Q1*Q1*8 + 2*D1*D1*D1*4) + 16.0*dofs + 4.0*dofs/VDIM;
   if (check)
   {
      // properties that hold at any size (the known answers stop at ~1e5 dofs): (x, A z) = (z, A x); the diffusion
      // operator annihilates constants, the mass operator integrates them: (1, M 1) = volume of the unit cube
      GridFunction z(&fes), w(&fes);
      z.Randomize(seed + 1);
      a.Mult(z, w);
      const double xw = x*w, zy = z*y;
      z = 1.0;
      a.Mult(z, w);
      const double cst = mass ? std::fabs((z*w)/VDIM - 1.0) : w.Norml2()/ynorm;
      printf("{\"phase\": \"check\", \"bp\": %d, \"p\": %d, \"dofs\": %d, \"sym_err\": %.3e, \"const_err\": %.3e}\n", bp, p, dofs,
             std::fabs(xw - zy)/std::fabs(xw), cst);
   }
   printf("{\"phase\": \"kernel\", \"bp\": %d, \"p\": %d, \"N\": %d, \"NE\": %d, \"dofs\": %d, \"norm_Ax\": %.15e, \"ms_per_apply\": %.6f, "
          "\"MDof/s\": %.3f, \"GB/s\": %.1f, \"hbm_frac\": %.4f, \"setup_s\": %.3f}\n", bp, p, N, mesh.GetNE(), dofs, ynorm, t_apply*1e3,
          1e-6*dofs/t_apply, 1e-9*bytes_apply/t_apply, 1e-9*bytes_apply/t_apply/8000.0, t_setup);

   // problem: f = 1, homogeneous Dirichlet, (Jacobi-)CG to rtol
   Array<int> ess_bdr(mesh.bdr_attributes.Max()), ess_tdof_list;
   ess_bdr = 1;
   fes.GetEssentialTrueDofs(ess_bdr, ess_tdof_list);
   LinearForm b(&fes);
   Vector uvec(3); uvec = 1.0; uvec /= uvec.Norml2();
   VectorConstantCoefficient unit_vec(uvec);
   if (VDIM == 1) { b.AddDomainIntegrator(new DomainLFIntegrator(one)); }
   else { b.AddDomainIntegrator(new VectorDomainLFIntegrator(unit_vec)); }
   b.UseFastAssembly(true);
   b.Assemble();
   x = 0.0;
   OperatorPtr A;
   Vector B, X;
   a.FormLinearSystem(ess_tdof_list, x, b, A, X, B);
   OperatorJacobiSmoother M(a, ess_tdof_list);
   CGSolver cg;
   cg.SetRelTol(rtol); cg.SetAbsTol(0.0); cg.SetMaxIter(max_it); cg.SetPrintLevel(-1);
   cg.SetOperator(*A);
   if (jacobi) { cg.SetPreconditioner(M); }
   cg.iterative_mode = false;
   cg.Mult(B, X);                   // warm-up + the known answer
   internal::DeviceSync();
   t0 = now();
   cg.Mult(B, X);
   internal::DeviceSync();
   const double t_solve = now() - t0;
   const int its = cg.GetNumIterations();
   // the stopping margin: (B r, r) one iteration before the stop, at the stop, and the threshold
   const std::vector<real_t> &h = cg.ResidualHistory();
   const double nom0 = h.empty() ? 0.0 : h[0], nom_last = h.empty() ? 0.0 : h.back();
   const double nom_prev = h.size() >= 2 ? h[h.size() - 2] : nom0;
   printf("{\"phase\": \"problem\", \"bp\": %d, \"p\": %d, \"N\": %d, \"dofs\": %d, \"iterations\": %d, \"converged\": %d, "
          "\"norm_X\": %.15e, \"norm_B\": %.15e, \"initial_norm\": %.15e, \"final_norm\": %.15e, \"solve_s\": %.6f, \"MDof_it/s\": %.3f, "
          "\"hbm_frac\": %.4f, \"nom0\": %.6e, \"nom_prev\": %.6e, \"nom_last\": %.6e, \"threshold\": %.6e}\n",
          bp, p, N, dofs, its, (int)cg.GetConverged(), X.Norml2(), B.Norml2(), cg.GetInitialNorm(), cg.GetFinalNorm(),
          t_solve, 1e-6*dofs*its/t_solve, 1e-9*(bytes_apply + 96.0*dofs)*its/t_solve/8000.0,
          nom0, nom_prev, nom_last, std::max(nom0*rtol*rtol, 0.0));
   fflush(stdout);
}

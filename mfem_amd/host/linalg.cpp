// linalg.cpp -- Vector, Operator, ConstrainedOperator, OperatorJacobiSmoother, CGSolver.
// All arithmetic goes through the C ABI (device kernels); host loops only move or inspect data.
#include "mfem.hpp"
#include <iomanip>

namespace mfem
{

namespace
{
// one reduction workspace + result slot per process (the library never allocates)
struct ReduceWs
{
   void *ws = nullptr; double *res = nullptr;
   void Ensure()
   {
      if (ws) { return; }
      ws = internal::DeviceAlloc((size_t)mfem_amd_reduce_workspace_bytes());
      res = (double*)internal::DeviceAlloc(256);
   }
} reduce_ws;
}

// ---- Vector ---------------------------------------------------------------------------
Vector &Vector::operator=(const Vector &v)
{
   if (this != &v) { data = v.data; }
   return *this;
}
Vector &Vector::operator=(real_t value)
{
   const int64_t n = Size64();
   if (n == 0) { return *this; }
   if (data.OnDevice()) { MFEM_AMD_CALL(mfem_amd_vec_fill(n, value, data.Write(), nullptr)); }
   else { real_t *p = data.HostWrite(); std::fill(p, p + n, value); }   // not yet on the device: plain fill
   return *this;
}
Vector &Vector::operator*=(real_t c)
{
   if (Size()) { real_t *p = data.ReadWrite(); MFEM_AMD_CALL(mfem_amd_vec_axpby(Size(), c, p, 0.0, nullptr, p, nullptr)); }
   return *this;
}
Vector &Vector::operator+=(const Vector &v) { return Add(1.0, v); }
Vector &Vector::operator-=(const Vector &v) { return Add(-1.0, v); }
Vector &Vector::Add(real_t a, const Vector &Va)
{
   MFEM_VERIFY(Size() == Va.Size(), "incompatible Vectors!");
   if (Size()) { real_t *p = data.ReadWrite(); MFEM_AMD_CALL(mfem_amd_vec_axpby(Size(), 1.0, p, a, Va.Read(), p, nullptr)); }
   return *this;
}
Vector &Vector::Set(real_t a, const Vector &x)
{
   MFEM_VERIFY(Size() == x.Size(), "incompatible Vectors!");
   if (Size()) { MFEM_AMD_CALL(mfem_amd_vec_axpby(Size(), a, x.Read(), 0.0, nullptr, data.Write(), nullptr)); }
   return *this;
}
real_t Vector::operator*(const Vector &v) const
{
   MFEM_VERIFY(Size() == v.Size(), "incompatible Vectors!");
   if (Size() == 0) { return 0.0; }
   reduce_ws.Ensure();
   MFEM_AMD_CALL(mfem_amd_vec_dot(Size(), Read(), v.Read(), reduce_ws.res, reduce_ws.ws, nullptr));
   double r;
   internal::CopyToHost(&r, reduce_ws.res, sizeof(double));
   return r;
}
real_t Vector::Normlinf() const { real_t m = 0; const real_t *p = HostRead(); for (int i = 0; i < Size(); i++) { m = std::max(m, std::fabs(p[i])); } return m; }
real_t Vector::Sum() const { real_t s = 0; const real_t *p = HostRead(); for (int i = 0; i < Size(); i++) { s += p[i]; } return s; }
real_t Vector::Max() const { const real_t *p = HostRead(); MFEM_VERIFY(Size() > 0, "empty Vector"); return *std::max_element(p, p + Size()); }
real_t Vector::Min() const { const real_t *p = HostRead(); MFEM_VERIFY(Size() > 0, "empty Vector"); return *std::min_element(p, p + Size()); }
void Vector::SetSubVector(const Array<int> &dofs, real_t value)
{
   if (dofs.Size()) { MFEM_AMD_CALL(mfem_amd_set_subvector(dofs.Size(), dofs.Read(), nullptr, value, data.ReadWrite(), nullptr)); }
}
void Vector::SetSubVector(const Array<int> &dofs, const Vector &elemvect)
{
   real_t *p = HostReadWrite(); const real_t *e = elemvect.HostRead();
   for (int i = 0; i < dofs.Size(); i++) { const int j = dofs[i]; if (j >= 0) { p[j] = e[i]; } else { p[-1-j] = -e[i]; } }
}
void Vector::GetSubVector(const Array<int> &dofs, Vector &elemvect) const
{
   elemvect.SetSize(dofs.Size());
   const real_t *p = HostRead(); real_t *e = elemvect.HostWrite();
   for (int i = 0; i < dofs.Size(); i++) { const int j = dofs[i]; e[i] = (j >= 0) ? p[j] : -p[-1-j]; }
}
void Vector::SetSubVectorComplement(const Array<int> &dofs, real_t val)
{
   // linalg/vector.cpp:794-801: save the listed entries, fill, restore them
   Vector keep(*this);
   *this = val;
   if (dofs.Size()) { MFEM_AMD_CALL(mfem_amd_set_subvector(dofs.Size(), dofs.Read(), keep.Read(), 0.0, data.ReadWrite(), nullptr)); }
}
void Vector::Randomize(int seed)
{
   // linalg/vector.cpp:955-966: glibc rand() scaled to [0,1)
   const real_t max = (real_t)(RAND_MAX) + 1.0;
   if (seed == 0) { seed = (int)time(nullptr); }
   srand((unsigned)seed);
   real_t *p = HostWrite();
   for (int i = 0; i < Size(); i++) { p[i] = std::abs(rand()/max); }
}
void Vector::Print(std::ostream &os, int width) const
{
   const real_t *p = HostRead();
   for (int i = 0; i < Size(); i++) { os << p[i] << (((i + 1) % width == 0 || i + 1 == Size()) ? "\n" : " "); }
}
void add(const Vector &x, real_t a, const Vector &y, Vector &z) { add(1.0, x, a, y, z); }
void add(real_t a, const Vector &x, real_t b, const Vector &y, Vector &z)
{
   MFEM_VERIFY(x.Size() == y.Size() && x.Size() == z.Size(), "incompatible Vectors!");
   if (!z.Size()) { return; }
   const real_t *xp = x.Read(), *yp = y.Read();
   real_t *zp = (&z == &x || &z == &y) ? z.ReadWrite() : z.Write();
   MFEM_AMD_CALL(mfem_amd_vec_axpby(z.Size(), a, xp, b, yp, zp, nullptr));
}
void subtract(const Vector &x, const Vector &y, Vector &z) { add(1.0, x, -1.0, y, z); }

// ---- Operator --------------------------------------------------------------------------
void Operator::AddMult(const Vector &x, Vector &y, const real_t a) const
{
   Vector t(y.Size()); Mult(x, t); y.Add(a, t);
}
// linalg/operator.cpp:114-129 with no prolongation/restriction (conforming serial space)
void Operator::FormLinearSystem(const Array<int> &ess_tdof_list, Vector &x, Vector &b,
                                Operator *&Aout, Vector &X, Vector &B, int copy_interior)
{
   X = x; B = b;
   if (!copy_interior) { X.SetSubVectorComplement(ess_tdof_list, 0.0); }
   ConstrainedOperator *constrained = new ConstrainedOperator(this, ess_tdof_list, false, DIAG_ONE);
   constrained->EliminateRHS(X, B);
   Aout = constrained;
}
void Operator::FormSystemOperator(const Array<int> &ess_tdof_list, Operator *&Aout)
{
   Aout = new ConstrainedOperator(this, ess_tdof_list, false, DIAG_ONE);
}
void Operator::RecoverFEMSolution(const Vector &X, const Vector &, Vector &x) { x = X; }

// ---- OperatorJacobiSmoother (linalg/solvers.cpp:331-453) -------------------------------
OperatorJacobiSmoother::OperatorJacobiSmoother(const Vector &d, const Array<int> &ess, const real_t dmp)
   : Solver(d.Size()), damping(dmp), ess_tdof_list(&ess) { Setup(d); }
void OperatorJacobiSmoother::Setup(const Vector &diag)
{
   height = width = diag.Size();
   dinv.SetSize(diag.Size());
   const int ness = ess_tdof_list ? ess_tdof_list->Size() : 0;
   MFEM_AMD_CALL(mfem_amd_jacobi_setup(diag.Size(), diag.Read(), damping, ness,
                                       ness ? ess_tdof_list->Read() : nullptr, dinv.Write(), nullptr));
}
void OperatorJacobiSmoother::SetOperator(const Operator &op)
{
   // solvers.cpp:371-399: the diagonal comes from the operator itself
   if (dinv.Size() == op.Height()) { return; }        // already set up from the form's diagonal
   Vector diag(op.Height()); op.AssembleDiagonal(diag); Setup(diag);
}
void OperatorJacobiSmoother::Mult(const Vector &x, Vector &y) const
{
   MFEM_VERIFY(x.Size() == Width() && y.Size() == Height(), "incompatible sizes");
   MFEM_VERIFY(!iterative_mode, "iterative_mode of OperatorJacobiSmoother is not built");
   MFEM_AMD_CALL(mfem_amd_jacobi_apply(x.Size(), dinv.Read(), x.Read(), y.Write(), nullptr));
}

// ---- CGSolver (linalg/solvers.cpp:869-1050) --------------------------------------------
// Generic operator / preconditioner (anything the fused device PCG of the C ABI does not take).  Organised like
// the device solver (csrc/vector_cg.hip pcg_impl): a state record, one routine per stage of an iteration, and a
// stop code; the decisions, their order and the messages are the reference's.
namespace
{
enum class PcgStop { None, Converged, NotSPDPreconditioner, ZeroCurvature, MaxIterations };

struct PcgRun
{
   const Operator &A; Solver *M; Vector &x, &r, &d, &z;
   const bool warn;
   real_t rr = 0.0, rr_prev = 0.0, rr0 = 0.0, dAd = 0.0, threshold = 0.0;   // (B r, r) now / before / initial, (A d, d)
   int it = 0;
   std::vector<real_t> &log;

   static void RequireFinite(const char *what, real_t v) { MFEM_VERIFY(std::isfinite(v), what << " = " << v); }
   // z = B r (or r itself without a preconditioner); returns (B r, r)
   real_t Precondition()
   {
      if (M) { M->Mult(r, z); return r*z; }
      return r*r;
   }
   const Vector &Preconditioned() const { return M ? z : r; }
   // (A d, d) with the reference's diagnostics; false: the direction has zero curvature, stop
   bool Curvature()
   {
      A.Mult(d, z);
      dAd = d*z;
      RequireFinite("den", dAd);
      if (dAd > 0.0) { return true; }
      if (warn && d*d > 0.0) { mfem::out << "PCG: The operator is not positive definite. (Ad, d) = " << dAd << '\n'; }
      return dAd != 0.0;
   }
   void WarnPreconditioner(real_t v) const
   {
      if (warn) { mfem::out << "PCG: The preconditioner is not positive definite. (Br, r) = " << v << '\n'; }
   }
   // x and r along d, new (B r, r), stopping test; then the next direction and its curvature
   PcgStop Advance()
   {
      const real_t alpha = rr_prev/dAd;
      add(x,  alpha, d, x);
      add(r, -alpha, z, r);
      rr = Precondition();
      RequireFinite("betanom", rr);
      if (rr < 0.0) { WarnPreconditioner(rr); return PcgStop::NotSPDPreconditioner; }
      log.push_back(rr);
      return PcgStop::None;
   }
   PcgStop NextDirection(int max_iter)
   {
      if (rr <= threshold) { return PcgStop::Converged; }
      if (++it > max_iter) { return PcgStop::MaxIterations; }
      add(Preconditioned(), rr/rr_prev, d, d);
      if (!Curvature()) { return PcgStop::ZeroCurvature; }
      rr_prev = rr;
      return PcgStop::None;
   }
};
}

void CGSolver::Mult(const Vector &b, Vector &x) const
{
   MFEM_VERIFY(oper, "CGSolver: operator is not set");
   history.clear();
   if (FusedMult(b, x)) { return; }

   const bool every = print_level == 1, summary = print_level == 2, ends = print_level == 3;
   auto line = [](int k, real_t v, const char *tail) { mfem::out << "   Iteration : " << std::setw(3) << k << "  (B r, r) = " << v << tail; };
   r.SetSize(width); d.SetSize(width); z.SetSize(width);
   PcgRun run{*oper, prec, x, r, d, z, print_level >= 0, 0, 0, 0, 0, 0, 0, history};

   // iteration 0: residual, first direction, (B r, r)_0
   if (iterative_mode) { oper->Mult(x, r); subtract(b, r, r); }
   else { r = b; x = 0.0; }
   run.rr0 = run.rr = run.rr_prev = run.Precondition();
   d = run.Preconditioned();
   if (run.rr0 >= 0.0) { initial_norm = std::sqrt(run.rr0); }
   PcgRun::RequireFinite("nom", run.rr0);
   history.push_back(run.rr0);
   if (every || ends) { line(0, run.rr0, ends ? " ...\n" : "\n"); }
   converged = false; final_iter = 0;
   if (run.rr0 < 0.0)
   {
      run.WarnPreconditioner(run.rr0);
      initial_norm = final_norm = run.rr0;
      return;
   }
   run.threshold = std::max(run.rr0*rel_tol*rel_tol, abs_tol*abs_tol);
   final_norm = std::sqrt(run.rr0);
   if (run.rr0 <= run.threshold) { converged = true; return; }
   if (!run.Curvature()) { return; }

   run.it = 1;
   PcgStop stop = PcgStop::None;
   while (stop == PcgStop::None)
   {
      stop = run.Advance();
      if (stop != PcgStop::None) { break; }
      if (every) { line(run.it, run.rr, ""); mfem::out << std::endl; }
      stop = run.NextDirection(max_iter);
   }
   converged = stop == PcgStop::Converged;
   final_iter = stop == PcgStop::MaxIterations ? max_iter : run.it;
   if (ends && !every) { line(final_iter, run.rr, "\n"); }
   if (summary || (run.warn && !converged)) { mfem::out << "PCG: Number of iterations: " << final_iter << '\n'; }
   if (summary || every || ends) { mfem::out << "Average reduction factor = " << std::pow(run.rr/run.rr0, 0.5/final_iter) << '\n'; }
   if (run.warn && !converged) { mfem::out << "PCG: No convergence!" << '\n'; }
   final_norm = std::sqrt(run.rr);
}

void CG(const Operator &A, const Vector &b, Vector &x, int print_iter, int max_num_iter,
        real_t RTOLERANCE, real_t ATOLERANCE)
{
   CGSolver cg;
   cg.SetPrintLevel(print_iter); cg.SetMaxIter(max_num_iter);
   cg.SetRelTol(std::sqrt(RTOLERANCE)); cg.SetAbsTol(std::sqrt(ATOLERANCE));
   cg.SetOperator(A);
   cg.Mult(b, x);
}
void PCG(const Operator &A, Solver &B, const Vector &b, Vector &x, int print_iter, int max_num_iter,
         real_t RTOLERANCE, real_t ATOLERANCE)
{
   CGSolver pcg;
   pcg.SetPrintLevel(print_iter); pcg.SetMaxIter(max_num_iter);
   pcg.SetRelTol(std::sqrt(RTOLERANCE)); pcg.SetAbsTol(std::sqrt(ATOLERANCE));
   pcg.SetOperator(A); pcg.SetPreconditioner(B);
   pcg.Mult(b, x);
}

} // namespace mfem

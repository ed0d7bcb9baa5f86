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
   const int n = Size();
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
static bool IsFinite(real_t v) { return std::isfinite(v); }

void CGSolver::Mult(const Vector &b, Vector &x) const
{
   MFEM_VERIFY(oper, "CGSolver: operator is not set");
   history.clear();
   if (FusedMult(b, x)) { return; }

   // generic operator / preconditioner: the reference's sequence on device vectors
   const bool it_print = print_level == 1, summary = print_level == 2, first_last = print_level == 3;
   const bool warnings = print_level >= 0;
   int i;
   real_t r0, den, nom, nom0, betanom = 0, alpha, beta;
   r.SetSize(width); d.SetSize(width); z.SetSize(width);
   if (iterative_mode) { oper->Mult(x, r); subtract(b, r, r); }
   else { r = b; x = 0.0; }
   if (prec) { prec->Mult(r, z); d = z; }
   else { d = r; }
   nom0 = nom = Dot(d, r);
   if (nom0 >= 0.0) { initial_norm = std::sqrt(nom0); }
   MFEM_VERIFY(IsFinite(nom), "nom = " << nom);
   history.push_back(nom);
   if (it_print || first_last)
   { mfem::out << "   Iteration : " << std::setw(3) << 0 << "  (B r, r) = " << nom << (first_last ? " ...\n" : "\n"); }
   if (nom < 0.0)
   {
      if (warnings) { mfem::out << "PCG: The preconditioner is not positive definite. (Br, r) = " << nom << '\n'; }
      converged = false; final_iter = 0; initial_norm = nom; final_norm = nom;
      return;
   }
   r0 = std::max(nom*rel_tol*rel_tol, abs_tol*abs_tol);
   if (nom <= r0) { converged = true; final_iter = 0; final_norm = std::sqrt(nom); return; }
   oper->Mult(d, z);
   den = Dot(z, d);
   MFEM_VERIFY(IsFinite(den), "den = " << den);
   if (den <= 0.0)
   {
      if (Dot(d, d) > 0.0 && warnings) { mfem::out << "PCG: The operator is not positive definite. (Ad, d) = " << den << '\n'; }
      if (den == 0.0) { converged = false; final_iter = 0; final_norm = std::sqrt(nom); return; }
   }
   converged = false; final_iter = max_iter;
   for (i = 1; true; )
   {
      alpha = nom/den;
      add(x,  alpha, d, x);
      add(r, -alpha, z, r);
      if (prec) { prec->Mult(r, z); betanom = Dot(r, z); }
      else { betanom = Dot(r, r); }
      MFEM_VERIFY(IsFinite(betanom), "betanom = " << betanom);
      if (betanom < 0.0)
      {
         if (warnings) { mfem::out << "PCG: The preconditioner is not positive definite. (Br, r) = " << betanom << '\n'; }
         converged = false; final_iter = i; break;
      }
      history.push_back(betanom);
      if (it_print) { mfem::out << "   Iteration : " << std::setw(3) << i << "  (B r, r) = " << betanom << std::endl; }
      if (betanom <= r0) { converged = true; final_iter = i; break; }
      if (++i > max_iter) { break; }
      beta = betanom/nom;
      if (prec) { add(z, beta, d, d); }
      else { add(r, beta, d, d); }
      oper->Mult(d, z);
      den = Dot(d, z);
      MFEM_VERIFY(IsFinite(den), "den = " << den);
      if (den <= 0.0)
      {
         if (Dot(d, d) > 0.0 && warnings) { mfem::out << "PCG: The operator is not positive definite. (Ad, d) = " << den << '\n'; }
         if (den == 0.0) { final_iter = i; break; }
      }
      nom = betanom;
   }
   if (first_last && !it_print) { mfem::out << "   Iteration : " << std::setw(3) << final_iter << "  (B r, r) = " << betanom << '\n'; }
   if (summary || (warnings && !converged)) { mfem::out << "PCG: Number of iterations: " << final_iter << '\n'; }
   if (summary || it_print || first_last) { mfem::out << "Average reduction factor = " << std::pow(betanom/nom0, 0.5/final_iter) << '\n'; }
   if (warnings && !converged) { mfem::out << "PCG: No convergence!" << '\n'; }
   final_norm = std::sqrt(betanom);
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

// mfem.hpp -- C++ host layer of mfem_amd: the MFEM API surface of the partial-assembly
// (AssemblyLevel::PARTIAL) operator-apply + Krylov path for H1 hexahedral meshes, written
// from scratch above the C ABI of include/mfem_amd.h (hand-written HIP for gfx950).
//
// Same class names, method names, argument meaning and error behaviour (abort with a
// message) as the reference, so drivers written for MFEM's ex1 / CEED bake-off problems
// compile and run against it.  Only what that path needs exists; everything that computes
// runs on the GPU through the C ABI -- there is no CPU device and no fallback.
// Reference interfaces are cited as file:line relative to the MFEM source root.
#ifndef MFEM_AMD_HOST_MFEM_HPP
#define MFEM_AMD_HOST_MFEM_HPP

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/mfem_amd.h"
#include "../../include/mfem_amd_comm.h"

namespace mfem
{

typedef double real_t;

// ---- errors (general/error.hpp:  MFEM_ABORT / MFEM_VERIFY print and abort) ------------
[[noreturn]] void mfem_error(const char *msg = nullptr);
[[noreturn]] void mfem_abort_at(const std::string &msg, const char *file, int line);
#define MFEM_ABORT(msg) \
   do { std::ostringstream mfem_msg_; mfem_msg_ << msg; \
        ::mfem::mfem_abort_at(mfem_msg_.str(), __FILE__, __LINE__); } while (0)
#define MFEM_VERIFY(cond, msg) do { if (!(cond)) { MFEM_ABORT("Verification failed: (" #cond ") is false:\n --> " << msg); } } while (0)
#ifdef MFEM_DEBUG
#define MFEM_ASSERT(cond, msg) MFEM_VERIFY(cond, msg)
#else
#define MFEM_ASSERT(cond, msg)
#endif
#define MFEM_WARNING(msg) do { std::cerr << "\n\nMFEM Warning: " << msg << std::endl; } while (0)
/// C-ABI status -> MFEM's abort-with-message convention.
#define MFEM_AMD_CALL(expr) do { if ((expr) != 0) { MFEM_ABORT("mfem_amd: " << mfem_amd_last_error()); } } while (0)

extern std::ostream &out;   // general/globals.hpp:66
extern std::ostream &err;

// ---- device memory (general/mem_manager.hpp, reduced to one host + one HIP buffer) ----
namespace internal
{
void *DeviceAlloc(size_t bytes);
void DeviceFree(void *p);
void CopyToDevice(void *dst, const void *src, size_t bytes);
void CopyToHost(void *dst, const void *src, size_t bytes);
void CopyOnDevice(void *dst, const void *src, size_t bytes);
void CopyRowsOnDevice(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows);
void DeviceSync();

/// Host buffer + device mirror, both created on first use, with validity flags: the semantics of
/// Memory<T>::Read/Write/ReadWrite (general/device.hpp:352-400).  Sizes are 64-bit -- the reference's containers
/// index with int (linalg/dtensor.hpp:32,121; general/array.hpp), which cannot hold BP3's pa_data at 1e8 dofs -- and a
/// buffer that is only ever written and read on the device (pa_data, E-vectors) never gets a host allocation.
template <class T> class Dual
{
   mutable T *h = nullptr, *d = nullptr;
   int64_t n = 0, cap = 0;
   mutable bool hv = true, dv = false;
   void EnsureHost() const
   {
      if (!h && cap > 0)
      {
         h = (T*)std::malloc(sizeof(T)*(size_t)cap);
         MFEM_VERIFY(h, "out of host memory (" << sizeof(T)*(size_t)cap << " bytes)");
      }
   }
public:
   Dual() {}
   ~Dual() { Destroy(); }
   Dual(const Dual &o) { CopyFrom(o); }
   Dual &operator=(const Dual &o) { if (this != &o) { CopyFrom(o); } return *this; }
   Dual(Dual &&o) noexcept { Steal(o); }
   Dual &operator=(Dual &&o) noexcept { if (this != &o) { Destroy(); Steal(o); } return *this; }
   /// the reference's int-valued size; aborts (instead of wrapping) when the buffer is larger: use Size64()
   int Size() const
   {
      MFEM_VERIFY(n <= (int64_t)2147483647, "mfem_amd: a buffer of " << n << " entries has no int size: use Size64()");
      return (int)n;
   }
   int64_t Size64() const { return n; }
   int64_t Capacity() const { return cap; }
   void Destroy()
   {
      std::free(h); if (d) { DeviceFree(d); }
      h = d = nullptr; n = cap = 0; hv = true; dv = false;
   }
   /// Resize; contents are kept only if @a keep (and only the valid side).
   void SetSize(int64_t nn, bool keep = false)
   {
      MFEM_VERIFY(nn >= 0, "negative size");
      if (nn <= cap) { n = nn; return; }
      const int64_t ncap = keep ? std::max(nn, 2*cap) : nn;
      T *nh = nullptr;
      if (keep && n > 0)
      {
         nh = (T*)std::malloc(sizeof(T)*(size_t)ncap);
         MFEM_VERIFY(nh, "out of host memory");
         std::memcpy(nh, HostRead(), sizeof(T)*(size_t)n);
      }
      std::free(h); if (d) { DeviceFree(d); d = nullptr; }
      h = nh; cap = ncap; n = nn; hv = true; dv = false;
   }
   const T *HostRead() const
   {
      EnsureHost();
      if (!hv) { CopyToHost(h, d, sizeof(T)*(size_t)n); hv = true; }
      return h;
   }
   T *HostWrite() { EnsureHost(); hv = true; dv = false; return h; }
   T *HostReadWrite() { HostRead(); dv = false; return h; }
   const T *Read() const
   {
      if (!d && cap > 0) { d = (T*)DeviceAlloc(sizeof(T)*(size_t)cap); }
      if (!dv && n > 0) { EnsureHost(); CopyToDevice(d, h, sizeof(T)*(size_t)n); }
      dv = true;
      return d;
   }
   T *Write()
   {
      if (!d && cap > 0) { d = (T*)DeviceAlloc(sizeof(T)*(size_t)cap); }
      dv = true; hv = false; return d;
   }
   T *ReadWrite() { Read(); hv = false; return d; }
   bool OnDevice() const { return dv; }
private:
   void CopyFrom(const Dual &o)
   {
      SetSize(o.n);
      if (o.dv && !o.hv) { if (n > 0) { CopyOnDevice(Write(), o.d, sizeof(T)*(size_t)n); } }
      else if (n > 0 && o.h) { std::memcpy(HostWrite(), o.h, sizeof(T)*(size_t)n); }
   }
   void Steal(Dual &o)
   {
      h = o.h; d = o.d; n = o.n; cap = o.cap; hv = o.hv; dv = o.dv;
      o.h = o.d = nullptr; o.n = o.cap = 0; o.hv = true; o.dv = false;
   }
};
} // namespace internal

// ---- Array<T> (general/array.hpp) -----------------------------------------------------
template <class T> class Array
{
   internal::Dual<T> mem;
public:
   Array() {}
   explicit Array(int n) { mem.SetSize(n); }
   Array(const T *src, int n) { mem.SetSize(n); std::copy(src, src + n, mem.HostWrite()); }
   int Size() const { return mem.Size(); }
   void SetSize(int64_t n) { mem.SetSize(n, true); }
   int64_t Size64() const { return mem.Size64(); }
   void SetSize(int n, const T &init)
   { int o = Size(); mem.SetSize(n, true); T *p = mem.HostReadWrite(); for (int i = o; i < n; i++) { p[i] = init; } }
   void Reserve(int n) { int s = Size(); if (n > mem.Capacity()) { mem.SetSize(n, true); mem.SetSize(s, true); } }
   void DeleteAll() { mem.Destroy(); }
   T &operator[](int i) { return mem.HostReadWrite()[i]; }
   const T &operator[](int i) const { return mem.HostRead()[i]; }
   int Append(const T &v) { int s = Size(); mem.SetSize(s + 1, true); mem.HostReadWrite()[s] = v; return s + 1; }
   T &Last() { return (*this)[Size() - 1]; }
   Array &operator=(const T &v) { T *p = mem.HostWrite(); std::fill(p, p + Size(), v); return *this; }
   T Max() const { const T *p = mem.HostRead(); MFEM_VERIFY(Size() > 0, "Array is empty"); return *std::max_element(p, p + Size()); }
   T Min() const { const T *p = mem.HostRead(); MFEM_VERIFY(Size() > 0, "Array is empty"); return *std::min_element(p, p + Size()); }
   void Sort() { T *p = mem.HostReadWrite(); std::sort(p, p + Size()); }
   void Unique() { T *p = mem.HostReadWrite(); mem.SetSize((int)(std::unique(p, p + Size()) - p), true); }
   int Find(const T &v) const { const T *p = mem.HostRead(); for (int i = 0; i < Size(); i++) { if (p[i] == v) { return i; } } return -1; }
   void Copy(Array &o) const { o.mem = mem; }
   T *GetData() { return mem.HostReadWrite(); }
   const T *GetData() const { return mem.HostRead(); }
   T *begin() { return GetData(); }
   T *end() { return GetData() + Size(); }
   const T *begin() const { return GetData(); }
   const T *end() const { return GetData() + Size(); }
   const T *Read(bool = true) const { return mem.Read(); }
   T *Write(bool = true) { return mem.Write(); }
   T *ReadWrite(bool = true) { return mem.ReadWrite(); }
   const T *HostRead() const { return mem.HostRead(); }
   T *HostWrite() { return mem.HostWrite(); }
   T *HostReadWrite() { return mem.HostReadWrite(); }
   void Print(std::ostream &os = mfem::out, int width = 4) const
   { for (int i = 0; i < Size(); i++) { os << (*this)[i] << (((i + 1) % width == 0 || i + 1 == Size()) ? "\n" : " "); } }
};

// ---- Device (general/device.hpp:120-330): only the HIP backend exists here -----------
struct Backend { enum Id { CPU = 1, HIP = 1 << 6 }; };
class Device
{
   static bool configured;
   static int dev_id;
public:
   Device() {}
   /// "hip", "hip:2", "gpu"; "cpu" aborts: this library computes on the MI355X only.
   Device(const std::string &config, int dev = 0) { Configure(config, dev); }
   void Configure(const std::string &config, int dev = 0);
   void Print(std::ostream &os = mfem::out);
   static bool IsEnabled() { return configured; }
   static bool IsConfigured() { return configured; }
   static bool Allows(unsigned long) { return configured; }
   static int GetId() { return dev_id; }
   /// Called by every object that touches the GPU: configures device 0 on first use.
   static void Require();
};

// ---- OptionsParser (general/optparser.hpp) ---------------------------------------------
class OptionsParser
{
   struct Option
   {
      int type; void *var; std::string s, l, ns, nl, descr; bool required;
   };
   int argc; char **argv;
   std::vector<Option> options;
   int error_type = 0, error_idx = 0;
   void WriteValue(const Option &opt, std::ostream &os) const;
public:
   OptionsParser(int argc_, char *argv_[]) : argc(argc_), argv(argv_) {}
   void AddOption(bool *var, const char *enable_short, const char *enable_long,
                  const char *disable_short, const char *disable_long, const char *descr, bool req = false)
   { options.push_back({0, var, enable_short, enable_long, disable_short, disable_long, descr, req}); }
   void AddOption(int *var, const char *s, const char *l, const char *descr, bool req = false)
   { options.push_back({1, var, s, l, "", "", descr, req}); }
   void AddOption(real_t *var, const char *s, const char *l, const char *descr, bool req = false)
   { options.push_back({2, var, s, l, "", "", descr, req}); }
   void AddOption(const char **var, const char *s, const char *l, const char *descr, bool req = false)
   { options.push_back({3, var, s, l, "", "", descr, req}); }
   void Parse();
   void ParseCheck(std::ostream &os = mfem::out);
   bool Good() const { return error_type == 0; }
   bool Help() const { return error_type == 1; }
   void PrintOptions(std::ostream &os) const;
   void PrintError(std::ostream &os) const;
   void PrintHelp(std::ostream &os) const;
   void PrintUsage(std::ostream &os) const;
};

/// GLVis socket (general/socketstream.hpp): visualisation is outside this path; the stream
/// swallows what is written to it and is never "open".
class socketstream : public std::ostream
{
   struct NullBuf : std::streambuf { int overflow(int c) override { return c; } } nb;
public:
   socketstream() : std::ostream(&nb) {}
   socketstream(const char *, int) : std::ostream(&nb) { setstate(std::ios::failbit); }
   bool is_open() const { return false; }
};

// ---- Vector (linalg/vector.hpp) -------------------------------------------------------
class Vector
{
protected:
   internal::Dual<real_t> data;
public:
   Vector() {}
   explicit Vector(int s) { data.SetSize(s); }
   explicit Vector(int64_t s) { data.SetSize(s); }
   Vector(const Vector &) = default;
   Vector(Vector &&) = default;
   Vector &operator=(Vector &&) = default;
   virtual ~Vector() {}
   int Size() const { return data.Size(); }
   /// 64-bit size (device-resident quadrature data can pass 2^31 entries; Size() aborts there)
   int64_t Size64() const { return data.Size64(); }
   void SetSize(int64_t s) { data.SetSize(s, true); }
   void UseDevice(bool) {}
   bool UseDevice() const { return true; }
   void Destroy() { data.Destroy(); }
   // element access is host access (HostReadWrite semantics)
   real_t &operator()(int i) { return data.HostReadWrite()[i]; }
   const real_t &operator()(int i) const { return data.HostRead()[i]; }
   real_t &operator[](int i) { return (*this)(i); }
   const real_t &operator[](int i) const { return (*this)(i); }
   real_t *GetData() { return data.HostReadWrite(); }
   const real_t *GetData() const { return data.HostRead(); }
   const real_t *Read(bool = true) const { return data.Read(); }
   real_t *Write(bool = true) { return data.Write(); }
   real_t *ReadWrite(bool = true) { return data.ReadWrite(); }
   const real_t *HostRead() const { return data.HostRead(); }
   real_t *HostWrite() { return data.HostWrite(); }
   real_t *HostReadWrite() { return data.HostReadWrite(); }
   // arithmetic: all on the device through the C ABI (linalg/vector.cpp)
   Vector &operator=(const Vector &v);
   Vector &operator=(real_t value);                 // vector.cpp:232
   Vector &operator*=(real_t c);
   Vector &operator/=(real_t c) { return (*this) *= (1.0/c); }
   Vector &operator+=(const Vector &v);
   Vector &operator-=(const Vector &v);
   Vector &Add(real_t a, const Vector &Va);         // this += a*Va
   Vector &Set(real_t a, const Vector &x);          // this = a*x
   void Neg() { (*this) *= -1.0; }
   real_t operator*(const Vector &v) const;         // vector.cpp:1079-1152
   real_t Norml2() const { return std::sqrt((*this)*(*this)); }
   real_t Normlinf() const;
   real_t Sum() const;
   real_t Max() const;
   real_t Min() const;
   void SetSubVector(const Array<int> &dofs, real_t value);
   void SetSubVector(const Array<int> &dofs, const Vector &elemvect);   // elemvect indexed like dofs
   void GetSubVector(const Array<int> &dofs, Vector &elemvect) const;
   void SetSubVectorComplement(const Array<int> &dofs, real_t val);
   /// uniform [0,1) from glibc rand(), as linalg/vector.cpp:955-966.
   void Randomize(int seed = 0);
   void Print(std::ostream &os = mfem::out, int width = 8) const;
   void Swap(Vector &o) { std::swap(data, o.data); }
};
/// z = x + a*y, z = a*x + b*y, z = x - y (linalg/vector.cpp:414-473,570)
void add(const Vector &x, real_t a, const Vector &y, Vector &z);
void add(real_t a, const Vector &x, real_t b, const Vector &y, Vector &z);
void subtract(const Vector &x, const Vector &y, Vector &z);
inline real_t InnerProduct(const Vector &x, const Vector &y) { return x*y; }

// ---- Operator / Solver (linalg/operator.hpp) ------------------------------------------
class Operator
{
protected:
   int height, width;
public:
   enum DiagonalPolicy { DIAG_ZERO, DIAG_ONE, DIAG_KEEP };
   enum Type { ANY_TYPE, MFEM_SPARSEMAT };
   explicit Operator(int s = 0) : height(s), width(s) {}
   Operator(int h, int w) : height(h), width(w) {}
   virtual ~Operator() {}
   int Height() const { return height; }
   int NumRows() const { return height; }
   int Width() const { return width; }
   int NumCols() const { return width; }
   virtual void Mult(const Vector &x, Vector &y) const = 0;                 // operator.hpp:89
   virtual void MultTranspose(const Vector &, Vector &) const
   { mfem_error("Operator::MultTranspose() is not overridden!"); }
   virtual void AddMult(const Vector &x, Vector &y, const real_t a = 1.0) const;
   virtual void AssembleDiagonal(Vector &) const
   { MFEM_ABORT("Not relevant or not implemented for this Operator."); }     // operator.hpp:143
   virtual const Operator *GetProlongation() const { return nullptr; }
   virtual const Operator *GetRestriction() const { return nullptr; }
   /// operator.cpp:114-129
   void FormLinearSystem(const Array<int> &ess_tdof_list, Vector &x, Vector &b,
                         Operator *&A, Vector &X, Vector &B, int copy_interior = 0);
   void FormSystemOperator(const Array<int> &ess_tdof_list, Operator *&A);
   virtual void RecoverFEMSolution(const Vector &X, const Vector &b, Vector &x);
};

class Solver : public Operator
{
public:
   bool iterative_mode;
   explicit Solver(int s = 0, bool iter_mode = false) : Operator(s), iterative_mode(iter_mode) {}
   virtual void SetOperator(const Operator &op) = 0;
};

/// OperatorHandle / OperatorPtr (linalg/handle.hpp): owning or borrowing pointer.
class OperatorHandle
{
   Operator *oper = nullptr; bool own = false;
public:
   OperatorHandle() {}
   explicit OperatorHandle(Operator::Type) {}
   ~OperatorHandle() { Clear(); }
   OperatorHandle(const OperatorHandle &) = delete;
   void Clear() { if (own) { delete oper; } oper = nullptr; own = false; }
   void Reset(Operator *A, bool own_A = true) { Clear(); oper = A; own = own_A; }
   Operator *Ptr() const { return oper; }
   Operator *operator->() const { return oper; }
   Operator &operator*() const { return *oper; }
   template <class OpType> OpType *As() const { return static_cast<OpType*>(oper); }
   template <class OpType> OpType *Is() const { return dynamic_cast<OpType*>(oper); }
   void SetOperatorOwner(bool o = true) { own = o; }
   bool OwnsOperator() const { return own; }
};
typedef OperatorHandle OperatorPtr;

class PABilinearFormExtension;

/// linalg/operator.hpp:895-, operator.cpp:511-646.
class ConstrainedOperator : public Operator
{
protected:
   Array<int> constraint_list;
   Operator *A; bool own_A;
   mutable Vector z, w;
   DiagonalPolicy diag_policy;
   const PABilinearFormExtension *pa = nullptr;     // fast path: fused masked apply
   mutable mfem_amd_pa_operator cop;
   Array<unsigned char> marker; Array<int> masked_gather;
   void RefreshOperator() const;
   friend class CGSolver;
public:
   ConstrainedOperator(Operator *A, const Array<int> &list, bool own_A = false,
                       DiagonalPolicy diag_policy = DIAG_ONE);
   ~ConstrainedOperator() override { if (own_A) { delete A; } }
   const Array<int> &GetEssentialTrueDofs() const { return constraint_list; }
   void AssembleDiagonal(Vector &diag) const override;
   void EliminateRHS(const Vector &x, Vector &b) const;      // operator.cpp:559-584
   void Mult(const Vector &x, Vector &y) const override;     // operator.cpp:586-646,710-720
};

class IterativeSolver : public Solver
{
protected:
   const Operator *oper = nullptr; Solver *prec = nullptr;
   int max_iter = 10, print_level = -1;
   real_t rel_tol = 0.0, abs_tol = 0.0;
   mutable int final_iter = -1; mutable bool converged = false;
   mutable real_t initial_norm = -1.0, final_norm = -1.0;
   real_t Dot(const Vector &x, const Vector &y) const { return x*y; }
public:
   IterativeSolver() : Solver(0, true) {}
   void SetRelTol(real_t rtol) { rel_tol = rtol; }
   void SetAbsTol(real_t atol) { abs_tol = atol; }
   void SetMaxIter(int max_it) { max_iter = max_it; }
   virtual void SetPrintLevel(int print_lvl) { print_level = print_lvl; }
   int GetNumIterations() const { return final_iter; }
   bool GetConverged() const { return converged; }
   real_t GetInitialNorm() const { return initial_norm; }
   real_t GetFinalNorm() const { return final_norm; }
   real_t GetFinalRelNorm() const { return final_norm/initial_norm; }
   virtual void SetPreconditioner(Solver &pr) { prec = &pr; prec->iterative_mode = false; }
   void SetOperator(const Operator &op) override
   { oper = &op; height = op.Height(); width = op.Width(); if (prec) { prec->SetOperator(*oper); } }
};

class BilinearForm;

/// linalg/solvers.hpp:331-, solvers.cpp:331-453.
class OperatorJacobiSmoother : public Solver
{
   Vector dinv; real_t damping; const Array<int> *ess_tdof_list = nullptr;
   friend class CGSolver;
public:
   OperatorJacobiSmoother(const real_t damping = 1.0) : damping(damping) {}
   OperatorJacobiSmoother(const BilinearForm &a, const Array<int> &ess_tdof_list, const real_t damping = 1.0);
   OperatorJacobiSmoother(const Vector &d, const Array<int> &ess_tdof_list, const real_t damping = 1.0);
   void Setup(const Vector &diag);
   void SetOperator(const Operator &op) override;
   void Mult(const Vector &x, Vector &y) const override;
   void MultTranspose(const Vector &x, Vector &y) const override { Mult(x, y); }
};

/// linalg/solvers.hpp:626-, solvers.cpp:855-1050.
typedef int MPI_Comm;                      // no MPI on the target image: a tag for the process group of Mpi::World()
constexpr MPI_Comm MPI_COMM_WORLD = 0x44000000;
typedef long long HYPRE_BigInt;

class CGSolver : public IterativeSolver
{
   mutable Vector r, d, z, ws;
   mutable std::vector<real_t> history;
   bool parallel = false;                  // CGSolver(MPI_Comm): dot products are all-reduced (solvers.cpp:44-54)
   bool FusedMult(const Vector &b, Vector &x) const;
public:
   CGSolver() {}
   explicit CGSolver(MPI_Comm) : parallel(true) {}
   void Mult(const Vector &b, Vector &x) const override;
   /// the "(B r, r)" sequence of the last solve (what print level 1 prints)
   const std::vector<real_t> &ResidualHistory() const { return history; }
};
/// solvers.cpp:1052-1081: tolerances are squared, as in the reference.
void CG(const Operator &A, const Vector &b, Vector &x, int print_iter = 0, int max_num_iter = 1000,
        real_t RTOLERANCE = 1e-12, real_t ATOLERANCE = 1e-24);
void PCG(const Operator &A, Solver &B, const Vector &b, Vector &x, int print_iter = 0,
         int max_num_iter = 1000, real_t RTOLERANCE = 1e-12, real_t ATOLERANCE = 1e-24);

// ---- quadrature (fem/intrules.hpp) ----------------------------------------------------
struct Geometry { enum Type { INVALID = -1, POINT = 0, SEGMENT, TRIANGLE, SQUARE, TETRAHEDRON, CUBE, PRISM, PYRAMID, NumGeom }; };
struct Quadrature1D { enum { Invalid = -1, GaussLegendre = 0, GaussLobatto = 1 }; };
struct BasisType { enum { Invalid = -1, GaussLegendre = 0, GaussLobatto = 1 }; };
struct Element { enum Type { POINT, SEGMENT, TRIANGLE, QUADRILATERAL, TETRAHEDRON, HEXAHEDRON, WEDGE, PYRAMID }; };

struct IntegrationPoint { real_t x = 0, y = 0, z = 0, weight = 0; };
class IntegrationRule
{
   std::vector<IntegrationPoint> pts;
   mutable Array<real_t> weights;
   int order = 0;
public:
   int n1d = 0, quad_type = 0;            // tensor rules remember their 1D factor
   std::vector<real_t> x1d, w1d;
   IntegrationRule() {}
   explicit IntegrationRule(int np) : pts(np) {}
   int GetNPoints() const { return (int)pts.size(); }
   int Size() const { return GetNPoints(); }
   IntegrationPoint &IntPoint(int i) { return pts[i]; }
   const IntegrationPoint &IntPoint(int i) const { return pts[i]; }
   int GetOrder() const { return order; }
   void SetOrder(int o) { order = o; }
   const Array<real_t> &GetWeights() const;          // intrules.cpp:56-84 layout (x fastest)
};
/// fem/intrules.cpp:1043-1251 for SEGMENT and CUBE (tensor Gauss-Legendre / Gauss-Lobatto).
class IntegrationRules
{
   int quad_type;
   mutable std::vector<IntegrationRule*> seg, cube;
public:
   explicit IntegrationRules(int ref = 0, int type = Quadrature1D::GaussLegendre) : quad_type(type) { (void)ref; }
   ~IntegrationRules();
   const IntegrationRule &Get(int GeomType, int Order) const;
};
extern IntegrationRules IntRules;
/// Points on [0,1] (fem/intrules.cpp:424-642).
void GaussLegendrePoints(int np, real_t *x, real_t *w);
void GaussLobattoPoints(int np, real_t *x, real_t *w);
/// Barycentric Lagrange basis on `nodes` tabulated at `pts`: B[q + nq*d], G[q + nq*d]
/// (fem/fe/fe_base.cpp:1754-1772,1861-1928,2622-2673).
void LagrangeBasisTables(int p, const real_t *nodes, int nq, const real_t *pts, real_t *B, real_t *G);

// ---- coefficients (fem/coefficient.hpp) -----------------------------------------------
class QuadratureFunction;
class Coefficient
{
public:
   virtual ~Coefficient() {}
   /// value at a physical point
   virtual real_t EvalAt(const Vector &x) const = 0;
   virtual bool IsConstant(real_t &) const { return false; }
   /// fem/coefficient.cpp:52-70: the values at every quadrature point of qf's space, (Q^3, NE)
   virtual void Project(QuadratureFunction &qf);
};
class ConstantCoefficient : public Coefficient
{
public:
   real_t constant;
   explicit ConstantCoefficient(real_t c = 1.0) : constant(c) {}
   real_t EvalAt(const Vector &) const override { return constant; }
   bool IsConstant(real_t &c) const override { c = constant; return true; }
};
/// fem/coefficient.hpp VectorCoefficient / VectorConstantCoefficient (the constant one is what the
/// bake-off right-hand sides use, tests/benchmarks/bench_ceed.cpp:54)
class VectorCoefficient
{
protected:
   int vdim;
public:
   explicit VectorCoefficient(int vd) : vdim(vd) {}
   virtual ~VectorCoefficient() {}
   int GetVDim() const { return vdim; }
   virtual bool IsConstant(Vector &) const { return false; }
   /// value at a physical point
   virtual void EvalAt(const Vector &x, Vector &v) const { Vector c; MFEM_VERIFY(IsConstant(c), "VectorCoefficient::EvalAt is not implemented"); v = c; (void)x; }
};
class VectorConstantCoefficient : public VectorCoefficient
{
   Vector vec;
public:
   explicit VectorConstantCoefficient(const Vector &v) : VectorCoefficient(v.Size()), vec(v) {}
   const Vector &GetVec() const { return vec; }
   bool IsConstant(Vector &v) const override { v = vec; return true; }
};
class FunctionCoefficient : public Coefficient
{
   std::function<real_t(const Vector &)> F;
public:
   explicit FunctionCoefficient(std::function<real_t(const Vector &)> f) : F(std::move(f)) {}
   real_t EvalAt(const Vector &x) const override { return F(x); }
};
class VectorFunctionCoefficient : public VectorCoefficient
{
   std::function<void(const Vector &, Vector &)> F;
public:
   VectorFunctionCoefficient(int dim, std::function<void(const Vector &, Vector &)> f) : VectorCoefficient(dim), F(std::move(f)) {}
   void EvalAt(const Vector &x, Vector &v) const override { v.SetSize(vdim); F(x, v); }
};
/// linalg/densemat.hpp, the little of it the matrix coefficients need: small column-major matrix on the host
class DenseMatrix
{
   int h = 0, w = 0; std::vector<real_t> a;
public:
   DenseMatrix() {}
   explicit DenseMatrix(int n) { SetSize(n); }
   DenseMatrix(int m, int n) { SetSize(m, n); }
   void SetSize(int n) { SetSize(n, n); }
   void SetSize(int m, int n) { h = m; w = n; a.assign((size_t)m*n, 0.0); }
   int Height() const { return h; }
   int Width() const { return w; }
   real_t &operator()(int i, int j) { return a[i + (size_t)h*j]; }
   const real_t &operator()(int i, int j) const { return a[i + (size_t)h*j]; }
   DenseMatrix &operator=(real_t v) { for (auto &t : a) { t = v; } return *this; }
};
/// linalg/symmat.hpp: upper triangle stored row by row, (0,0),(0,1),...,(1,1),...
class DenseSymmetricMatrix
{
   int n = 0; std::vector<real_t> a;
   int idx(int i, int j) const { if (i > j) { std::swap(i, j); } return i*n - i*(i - 1)/2 + (j - i); }
public:
   DenseSymmetricMatrix() {}
   explicit DenseSymmetricMatrix(int s) { SetSize(s); }
   void SetSize(int s) { n = s; a.assign((size_t)s*(s + 1)/2, 0.0); }
   int Height() const { return n; }
   int Width() const { return n; }
   real_t &operator()(int i, int j) { return a[idx(i, j)]; }
   const real_t &operator()(int i, int j) const { return a[idx(i, j)]; }
   const real_t *GetData() const { return a.data(); }
};
/// fem/coefficient.hpp MatrixCoefficient family.  Packed(): the entries the way
/// CoefficientVector::ProjectTranspose leaves them for the PA setup (fem/coefficient.cpp:2026-2056):
/// 9 row-major entries of K, or the 6 of the upper triangle for the symmetric classes.
class MatrixCoefficient
{
protected:
   int height, width;
public:
   explicit MatrixCoefficient(int dim) : height(dim), width(dim) {}
   virtual ~MatrixCoefficient() {}
   int GetHeight() const { return height; }
   int GetWidth() const { return width; }
   virtual bool IsSymmetric() const { return false; }
   virtual bool IsConstant() const { return false; }
   /// K at a physical point
   virtual void EvalAt(const Vector &x, DenseMatrix &K) const = 0;
   /// packed entries at a physical point: PackedSize() numbers
   int PackedSize() const { return IsSymmetric() ? height*(height + 1)/2 : height*width; }
   virtual void PackedAt(const Vector &x, real_t *c) const
   {
      DenseMatrix K; EvalAt(x, K);
      for (int i = 0; i < height; i++) { for (int j = 0; j < width; j++) { c[i*width + j] = K(i, j); } }
   }
};
class MatrixConstantCoefficient : public MatrixCoefficient
{
   DenseMatrix mat;
public:
   explicit MatrixConstantCoefficient(const DenseMatrix &m) : MatrixCoefficient(m.Height()), mat(m) {}
   const DenseMatrix &GetMatrix() const { return mat; }
   bool IsConstant() const override { return true; }
   void EvalAt(const Vector &, DenseMatrix &K) const override { K = mat; }
};
class MatrixFunctionCoefficient : public MatrixCoefficient
{
   std::function<void(const Vector &, DenseMatrix &)> F;
public:
   MatrixFunctionCoefficient(int dim, std::function<void(const Vector &, DenseMatrix &)> f) : MatrixCoefficient(dim), F(std::move(f)) {}
   void EvalAt(const Vector &x, DenseMatrix &K) const override { K.SetSize(height, width); F(x, K); }
};
class SymmetricMatrixCoefficient : public MatrixCoefficient
{
public:
   explicit SymmetricMatrixCoefficient(int dim) : MatrixCoefficient(dim) {}
   bool IsSymmetric() const override { return true; }
   virtual void EvalAt(const Vector &x, DenseSymmetricMatrix &K) const = 0;
   void EvalAt(const Vector &x, DenseMatrix &K) const override
   {
      DenseSymmetricMatrix S; EvalAt(x, S);
      K.SetSize(height, width);
      for (int i = 0; i < height; i++) { for (int j = 0; j < width; j++) { K(i, j) = S(i, j); } }
   }
   void PackedAt(const Vector &x, real_t *c) const override
   {
      DenseSymmetricMatrix S; EvalAt(x, S);
      for (int k = 0; k < PackedSize(); k++) { c[k] = S.GetData()[k]; }
   }
};
class SymmetricMatrixConstantCoefficient : public SymmetricMatrixCoefficient
{
   DenseSymmetricMatrix mat;
public:
   explicit SymmetricMatrixConstantCoefficient(const DenseSymmetricMatrix &m) : SymmetricMatrixCoefficient(m.Height()), mat(m) {}
   const DenseSymmetricMatrix &GetMatrix() const { return mat; }
   bool IsConstant() const override { return true; }
   using SymmetricMatrixCoefficient::EvalAt;
   void EvalAt(const Vector &, DenseSymmetricMatrix &K) const override { K = mat; }
};
class SymmetricMatrixFunctionCoefficient : public SymmetricMatrixCoefficient
{
   std::function<void(const Vector &, DenseSymmetricMatrix &)> F;
public:
   SymmetricMatrixFunctionCoefficient(int dim, std::function<void(const Vector &, DenseSymmetricMatrix &)> f)
      : SymmetricMatrixCoefficient(dim), F(std::move(f)) {}
   using SymmetricMatrixCoefficient::EvalAt;
   void EvalAt(const Vector &x, DenseSymmetricMatrix &K) const override { K.SetSize(height); F(x, K); }
};

// ---- quadrature-point data (fem/qspace.hpp, fem/qfunction.hpp, fem/coefficient.hpp:2432-2460) ----
class Mesh;
/// One tensor rule on every (hexahedral) element of a mesh.
class QuadratureSpace
{
   Mesh *mesh;
   const IntegrationRule *ir;
   int order;
public:
   QuadratureSpace(Mesh *m, int order_);
   QuadratureSpace(Mesh &m, const IntegrationRule &ir_);
   Mesh *GetMesh() const { return mesh; }
   int GetNE() const;
   int GetOrder() const { return order; }
   const IntegrationRule &GetIntRule(int = 0) const { return *ir; }
   const IntegrationRule &GetElementIntRule(int = 0) const { return *ir; }
   int GetSize() const { return GetNE()*ir->GetNPoints(); }
};
/// vdim values per quadrature point, (vdim, Q^3, NE); a Vector, so it lives on the device like any other
class QuadratureFunction : public Vector
{
   QuadratureSpace *qspace;
   int vdim;
public:
   explicit QuadratureFunction(QuadratureSpace *qs, int vdim_ = 1) : Vector((int64_t)vdim_*qs->GetSize()), qspace(qs), vdim(vdim_) {}
   QuadratureSpace *GetSpace() const { return qspace; }
   int GetVDim() const { return vdim; }
   QuadratureFunction &operator=(real_t v) { Vector::operator=(v); return *this; }
   QuadratureFunction &operator=(const Vector &v) { MFEM_VERIFY(v.Size() == Size(), "QuadratureFunction: size mismatch"); Vector::operator=(v); return *this; }
};
/// A coefficient given by its values at the quadrature points.  The PA setup kernels read those values where they
/// are (CoefficientVector::Project makes a reference, fem/coefficient.cpp:1992-1995): nothing is evaluated on the host.
class QuadratureFunctionCoefficient : public Coefficient
{
   const QuadratureFunction &QuadF;
public:
   explicit QuadratureFunctionCoefficient(const QuadratureFunction &qf) : QuadF(qf)
   { MFEM_VERIFY(qf.GetVDim() == 1, "QuadratureFunction's vdim must be 1"); }
   const QuadratureFunction &GetQuadFunction() const { return QuadF; }
   real_t EvalAt(const Vector &) const override
   { MFEM_ABORT("mfem_amd: a QuadratureFunctionCoefficient has values at its quadrature points only"); return 0.0; }
   void Project(QuadratureFunction &qf) override { qf = QuadF; }
};

// ---- finite elements: H1 on hexahedra (fem/fe_coll.hpp, fem/fe/fe_h1.cpp) --------------
struct DofToQuad
{
   enum Mode { FULL, TENSOR };
   int ndof, nqpt;
   Array<real_t> B, G, Bt, Gt;        // B[q + nqpt*d], Bt[d + ndof*q] (fe_base.cpp:2663-2667)
};
/// The one element type of this path: tensor-product Lagrange on Gauss-Lobatto points.
class FiniteElement
{
   int order, dim;
   Array<int> dof_map;                 // lexicographic -> native (fe_base.cpp:2482-2605)
   std::vector<real_t> nodes1d;
   mutable std::vector<std::pair<const IntegrationRule*, DofToQuad*>> d2q;
public:
   FiniteElement(int p, int dim);
   ~FiniteElement();
   int GetOrder() const { return order; }
   int GetDim() const { return dim; }
   int GetDof() const { int n = order + 1; return dim == 3 ? n*n*n : (dim == 2 ? n*n : n); }
   const Array<int> &GetDofMap() const { return dof_map; }
   const std::vector<real_t> &Nodes1D() const { return nodes1d; }
   const DofToQuad &GetDofToQuad(const IntegrationRule &ir, DofToQuad::Mode mode = DofToQuad::TENSOR) const;
};
class FiniteElementCollection
{
public:
   virtual ~FiniteElementCollection() {}
   virtual const char *Name() const = 0;
   virtual int GetOrder() const = 0;
   virtual const FiniteElement *FiniteElementForGeometry(Geometry::Type g) const = 0;
   static FiniteElementCollection *New(const char *name);
};
class H1_FECollection : public FiniteElementCollection
{
   int p, dim; char h1_name[32];
   FiniteElement *hex;
   std::vector<int> seg_ord[2], quad_ord[8];      // fe_coll.cpp:1788-1906
public:
   H1_FECollection(const int p, const int dim = 3, const int btype = BasisType::GaussLobatto);
   ~H1_FECollection() override { delete hex; }
   const char *Name() const override { return h1_name; }
   int GetOrder() const override { return p; }
   const FiniteElement *FiniteElementForGeometry(Geometry::Type g) const override
   { MFEM_VERIFY(g == Geometry::CUBE, "mfem_amd: only hexahedra"); return hex; }
   const int *SegDofOrder(int ori) const { return seg_ord[ori > 0 ? 0 : 1].data(); }
   const int *QuadDofOrder(int ori) const { return quad_ord[ori].data(); }
};

class GridFunction;
class FiniteElementSpace;

// ---- Mesh: conforming hexahedral meshes (mesh/mesh.hpp) --------------------------------
class Mesh
{
   int Dim = 3, spaceDim = 3;
   std::vector<real_t> vertices;          // 3 per vertex
   std::vector<int> elements, elem_attr;  // 8 vertices per hex
   std::vector<int> boundary, bdr_attr;   // 4 vertices per boundary quad
   // topology (FinalizeTopology, mesh/mesh.cpp:3503-3607)
   int NumOfEdges = 0, NumOfFaces = 0;
   std::vector<int> el_to_edge, el_to_face, bel_to_edge, be_to_face;
   std::vector<int> face_vertices;        // 4 per face, as seen from its first element
   std::vector<int> face_elem1, face_elem2, face_inf1, face_inf2;
   std::vector<int> edge_vertices;        // 2 per edge (v0 < v1)
   GridFunction *Nodes = nullptr; bool own_nodes = false;
   long sequence = 0;
   void Make3D(int nx, int ny, int nz, real_t sx, real_t sy, real_t sz, bool sfc_ordering);
   void FinalizeTopology();
   void CheckElementOrientation();
   void Load(std::istream &input, int refine);
   void SetNodesFromElementValues(int order, int ordering, const std::vector<real_t> &old_nodes, const std::vector<real_t> &xe_old);
   void SetNodesFromLegacyCubic(const std::vector<real_t> &raw, int ordering);
   void Destroy();
   void Swap(Mesh &o);
   friend class FiniteElementSpace;
   friend class ParMesh;
public:
   Array<int> attributes, bdr_attributes;
   /// recompute `attributes` / `bdr_attributes` after SetAttribute / SetBdrAttribute (mesh/mesh.cpp SetAttributes)
   void SetAttributes();
   Mesh() {}
   /// mesh/mesh.cpp:4944-: "MFEM mesh v1.0" (hexahedra) and "MFEM INLINE mesh v1.0" files.
   explicit Mesh(const std::string &filename, int generate_edges = 0, int refine = 1, bool fix_orientation = true);
   Mesh(const Mesh &) = delete;
   Mesh(Mesh &&o) noexcept { Swap(o); }
   Mesh &operator=(Mesh &&o) noexcept { Swap(o); return *this; }
   virtual ~Mesh() { Destroy(); }
   /// mesh/mesh.hpp Clear(): an empty mesh
   void Clear() { Destroy(); Mesh empty; Swap(empty); }
   /// mesh/mesh.cpp:4628-4636 (vertex/element/boundary conventions of Make3D :3684-3964)
   static Mesh MakeCartesian3D(int nx, int ny, int nz, Element::Type type, real_t sx = 1.0,
                               real_t sy = 1.0, real_t sz = 1.0, bool sfc_ordering = true);
   int Dimension() const { return Dim; }
   int SpaceDimension() const { return spaceDim; }
   int GetNV() const { return (int)vertices.size()/3; }
   int GetNE() const { return (int)elements.size()/8; }
   int GetNBE() const { return (int)boundary.size()/4; }
   int GetNEdges() const { return NumOfEdges; }
   int GetNFaces() const { return NumOfFaces; }
   long GetSequence() const { return sequence; }
   const real_t *GetVertex(int i) const { return &vertices[3*i]; }
   const int *GetElementVertices(int e) const { return &elements[8*e]; }
   const int *GetBdrElementVertices(int b) const { return &boundary[4*b]; }
   int GetAttribute(int e) const { return elem_attr[e]; }
   int GetBdrAttribute(int b) const { return bdr_attr[b]; }
   /// follow with SetAttributes() (mesh/mesh.hpp)
   void SetAttribute(int e, int attr) { elem_attr[e] = attr; }
   void SetBdrAttribute(int b, int attr) { bdr_attr[b] = attr; }
   const int *GetElementEdges(int e) const { return &el_to_edge[12*e]; }
   const int *GetElementFaces(int e) const { return &el_to_face[6*e]; }
   int GetBdrElementFaceIndex(int b) const { return be_to_face[b]; }
   int GetFaceElement1(int f) const { return face_elem1[f]; }     // the element a face was first seen from (-1: none)
   const int *GetBdrElementEdges(int b) const { return &bel_to_edge[4*b]; }
   const int *GetEdgeVertices(int i) const { return &edge_vertices[2*i]; }
   const int *GetFaceVertices(int f) const { return &face_vertices[4*f]; }
   /// orientation of local face lf of element e relative to the face's base (mesh.cpp:8082-8103)
   int GetElementFaceOrientation(int e, int lf) const;
   void MarkExternalBoundaries(Array<int> &bdr_marker, bool excl = true) const;   // mesh.cpp:1742
   void UniformRefinement(int ref_algo = 0);                                      // mesh.cpp:10141-
   GridFunction *GetNodes() { return Nodes; }
   const GridFunction *GetNodes() const { return Nodes; }
   const FiniteElementSpace *GetNodalFESpace() const;
   void EnsureNodes();                                                            // mesh.cpp:6994-7033
   void SetCurvature(int order, bool discont = false, int space_dim = -1, int ordering = 1);
   /// Replace the nodes by x -> f(x) applied to every node (Mesh::Transform, mesh.cpp:13625-)
   void Transform(std::function<void(const Vector &, Vector &)> f);
   void Print(std::ostream &os = mfem::out) const;                                // mesh.cpp:12225-
   void PrintInfo(std::ostream &os = mfem::out) const;
   static int GetQuadOrientation(const int *base, const int *test);               // mesh.cpp:7377
};
std::ostream &operator<<(std::ostream &os, const Mesh &mesh);
namespace internal
{
/// out(jx,jy,jz,c) = sum_i Px[jx][ix] Py[jy][iy] Pz[jz][iz] in(ix,iy,iz,c), c < 3; P* are (nout x nin) row-major
void TensorInterp3(int nin, int nout, const real_t *Px, const real_t *Py, const real_t *Pz, const real_t *in, real_t *out);
/// P[j][i] = L_i(pts[j]) for the Lagrange basis on `nodes`
std::vector<real_t> InterpMatrix(const std::vector<real_t> &nodes, const std::vector<real_t> &pts);
/// lexicographic nodal coordinates of every element, xe[(ix + D*(iy + D*iz)) + D^3*c + 3*D^3*e]; returns D
int ElementNodesLex(const Mesh &mesh, std::vector<real_t> &xe);
}

// ---- finite element space (fem/fespace.hpp) -------------------------------------------
struct Ordering { enum Type { byNODES, byVDIM }; };
enum class ElementDofOrdering { NATIVE, LEXICOGRAPHIC };
enum class AssemblyLevel { LEGACY = 0, LEGACYFULL = 0, FULL, ELEMENT, PARTIAL, NONE };

/// fem/restriction.hpp:40-, restriction.cpp:26-221.
class ElementRestriction : public Operator
{
   const FiniteElementSpace &fes;
   int ne, vdim, ndofs, dof; bool byvdim;
   Array<int> offsets, indices, gather_map;
   mutable Array<int> packed_indices;       // built on first use (a device kernel: host-only tools never ask for it)
   long long direct_rows = 0;
public:
   ElementRestriction(const FiniteElementSpace &, ElementDofOrdering);
   /// NE*(D1D-2)^3 when the last L-dofs are the element-interior dofs, element by element in lexicographic order (the
   /// native H1 numbering puts them there), else 0: rows the PA element kernel writes itself (mfem_amd_interior_rows)
   long long DirectRows() const { return direct_rows; }
   /// `indices` for the packed work E-vector that goes with the direct rows (mfem_amd_pack_indices); empty without them
   const Array<int> &PackedIndices() const;
   void Mult(const Vector &x, Vector &y) const override;
   void MultTranspose(const Vector &x, Vector &y) const override;
   void AddMultTranspose(const Vector &x, Vector &y, const real_t a = 1.0) const;
   void AbsMultTranspose(const Vector &x, Vector &y) const;
   const Array<int> &GatherMap() const { return gather_map; }
   const Array<int> &Indices() const { return indices; }
   const Array<int> &Offsets() const { return offsets; }
};

class FiniteElementSpace
{
   Mesh *mesh; const FiniteElementCollection *fec; const H1_FECollection *h1;
   int vdim; Ordering::Type ordering;
   int ndofs = 0, nvdofs = 0, nedofs = 0, nfdofs = 0, nbdofs = 0;
   std::vector<int> elem_dof;            // NE x dof, native ordering (fespace.cpp:3392-3500)
   mutable ElementRestriction *R_lex = nullptr;
   long sequence = 0;
   void Construct();
public:
   FiniteElementSpace(Mesh *mesh, const FiniteElementCollection *fec, int vdim = 1, int ordering = Ordering::byNODES);
   virtual ~FiniteElementSpace() { delete R_lex; }
   /// the mesh object moved (Mesh::Swap): follow it
   void SetMesh(Mesh *m) { mesh = m; }
   FiniteElementSpace(const FiniteElementSpace &) = delete;
   Mesh *GetMesh() const { return mesh; }
   const FiniteElementCollection *FEColl() const { return fec; }
   int GetVDim() const { return vdim; }
   Ordering::Type GetOrdering() const { return ordering; }
   int GetNDofs() const { return ndofs; }
   int GetVSize() const { return vdim*ndofs; }
   int GetTrueVSize() const { return GetVSize(); }
   int GetNE() const { return mesh->GetNE(); }
   int GetNBE() const { return mesh->GetNBE(); }
   int GetOrder(int) const { return fec->GetOrder(); }
   int GetMaxElementOrder() const { return fec->GetOrder(); }
   const FiniteElement *GetFE(int) const { return fec->FiniteElementForGeometry(Geometry::CUBE); }
   const FiniteElement *GetTypicalFE() const { return GetFE(0); }
   int DofToVDof(int dof, int vd) const { return ordering == Ordering::byNODES ? dof + vd*ndofs : dof*vdim + vd; }
   /// Native-order element dofs (vertices, edges, faces, interior), fespace.cpp:3392-3500.
   void GetElementDofs(int e, Array<int> &dofs) const;
   const int *ElementDofs(int e) const { return &elem_dof[(size_t)e*GetTypicalFE()->GetDof()]; }
   void GetBdrElementDofs(int b, Array<int> &dofs) const;
   /// The numbering itself, for one (boundary) element of any mesh: the (p+1)^3 (or (p+1)^2) dofs in the native order,
   /// 64-bit -- what a space on `m` would hold in its table, without building the space (the parallel space numbers the
   /// elements it needs of the serial mesh this way).  Class offsets follow from the mesh's own entity counts.
   static void ElementDofsOf(const Mesh &m, const H1_FECollection &h1, int e, int64_t *dofs);
   static void BdrElementDofsOf(const Mesh &m, const H1_FECollection &h1, int b, int64_t *dofs);
   static int64_t NDofsOf(const Mesh &m, int order);
   void GetEssentialVDofs(const Array<int> &bdr_attr_is_ess, Array<int> &ess_vdofs, int component = -1) const;
   void GetEssentialTrueDofs(const Array<int> &bdr_attr_is_ess, Array<int> &ess_tdof_list, int component = -1) const;
   void GetBoundaryTrueDofs(Array<int> &boundary_dofs, int component = -1) const;
   static void MarkerToList(const Array<int> &marker, Array<int> &list);
   const ElementRestriction *GetElementRestriction(ElementDofOrdering e_ordering) const;
   const Operator *GetProlongationMatrix() const { return nullptr; }
   const Operator *GetRestrictionOperator() const { return nullptr; }
   /// Rebuild after the mesh changed (fespace.cpp Update); grid functions call this.
   void Update();
   long GetSequence() const { return sequence; }
};
inline bool UsesTensorBasis(const FiniteElementSpace &) { return true; }

/// fem/gridfunc.hpp
class GridFunction : public Vector
{
   FiniteElementSpace *fes; FiniteElementCollection *fec_owned = nullptr; bool own_fes = false;
   long fes_sequence = 0;
public:
   explicit GridFunction(FiniteElementSpace *f) : Vector(f->GetVSize()), fes(f), fes_sequence(f->GetSequence()) {}
   ~GridFunction() override;
   GridFunction &operator=(real_t v) { Vector::operator=(v); return *this; }
   GridFunction &operator=(const Vector &v) { Vector::operator=(v); return *this; }
   FiniteElementSpace *FESpace() { return fes; }
   const FiniteElementSpace *FESpace() const { return fes; }
   FiniteElementCollection *OwnFEC() { return fec_owned; }
   /// the grid function takes ownership of the collection and of its space
   void MakeOwner(FiniteElementCollection *fec) { fec_owned = fec; own_fes = true; }
   void GetTrueDofs(Vector &tv) const { tv = *this; }
   void SetFromTrueDofs(const Vector &tv) { Vector::operator=(tv); }
   /// nodal interpolation of a function of the physical coordinates
   void ProjectCoefficient(Coefficient &coeff);
   /// physical coordinates of this (scalar) space's nodes, (ndofs x 3), from the mesh nodes
   void GetNodeCoordinates(std::vector<real_t> &xyz) const;
   void Save(std::ostream &os) const;            // gridfunc.cpp Save
   /// max-norm / L2 error helpers used by the drivers' checks
   real_t ComputeMaxError(Coefficient &exsol) const;
};
std::ostream &operator<<(std::ostream &os, const GridFunction &gf);

// ---- forms and integrators (fem/bilinearform.hpp, fem/bilininteg.hpp, fem/linearform.hpp)
class LinearFormIntegrator { public: virtual ~LinearFormIntegrator() {} };
/// (f, v) with rule order oa*p + ob (fem/lininteg.hpp; device path lininteg_domain.cpp:22-59)
class DomainLFIntegrator : public LinearFormIntegrator
{
public:
   Coefficient &Q; int oa, ob; const IntegrationRule *IntRule = nullptr;
   DomainLFIntegrator(Coefficient &QF, int a = 2, int b = 0) : Q(QF), oa(a), ob(b) {}
   DomainLFIntegrator(Coefficient &QF, const IntegrationRule *ir) : Q(QF), oa(1), ob(1), IntRule(ir) {}
};
/// (f, v) for a vector space (fem/lininteg.hpp VectorDomainLFIntegrator); constant f only.
class VectorDomainLFIntegrator : public LinearFormIntegrator
{
public:
   VectorCoefficient &Q; const IntegrationRule *IntRule = nullptr;
   explicit VectorDomainLFIntegrator(VectorCoefficient &QF) : Q(QF) {}
   VectorDomainLFIntegrator(VectorCoefficient &QF, const IntegrationRule *ir) : Q(QF), IntRule(ir) {}
};
class LinearForm : public Vector
{
   FiniteElementSpace *fes; std::vector<LinearFormIntegrator*> domain_integs;
public:
   explicit LinearForm(FiniteElementSpace *f) : Vector(f->GetVSize()), fes(f) {}
   ~LinearForm() override { for (auto *i : domain_integs) { delete i; } }
   FiniteElementSpace *FESpace() { return fes; }
   void AddDomainIntegrator(LinearFormIntegrator *lfi) { domain_integs.push_back(lfi); }
   void UseFastAssembly(bool) {}
   void Assemble();
   LinearForm &operator=(real_t v) { Vector::operator=(v); return *this; }
};

/// fem/bilininteg.hpp:52-97: the partial-assembly interface of an integrator.
class BilinearFormIntegrator
{
protected:
   const IntegrationRule *IntRule;
   // PA state (device)
   int kind = -1, dofs1D = 0, quad1D = 0, ne = 0, pa_layout = 1;
   int vdim = 1;                         // components of the space (vector integrators)
   bool vector_integrator = false;
   bool symmetric = true;                // false: 9-plane D of a full matrix coefficient (fem/bilininteg.hpp:2208)
   /// operator kind at the C ABI: 0 mass, 1 diffusion, 2 diffusion with the non-symmetric D
   int OperatorKind() const { return (kind == 1 && !symmetric) ? 2 : kind; }
   const DofToQuad *maps = nullptr;
   Vector pa_data;
   mutable Vector pa_data_mfem;          // MFEM-layout copy, made on the first E-vector AddMultPA
   const Vector &MFEMLayoutData() const;
   explicit BilinearFormIntegrator(const IntegrationRule *ir = nullptr) : IntRule(ir) {}
   void SetupPA(const FiniteElementSpace &fes, int kind, Coefficient *Q, const IntegrationRule &ir,
                VectorCoefficient *VQ = nullptr, MatrixCoefficient *MQ = nullptr);
   /// vector integrators with a VectorCoefficient: one scalar pa_data per component (coeff_vdim = vdim,
   /// fem/integ/bilininteg_vecdiffusion_pa.cpp:109-134, bilininteg_vecmass_pa.cpp:57-75), pa_comp_stride apart
   void SetupPAComponents(const FiniteElementSpace &fes, int kind, VectorCoefficient &VQ, const IntegrationRule &ir);
   int64_t pa_comp_stride = 0;
   friend class PABilinearFormExtension;
   friend class BilinearForm;
public:
   virtual ~BilinearFormIntegrator() {}
   void SetIntRule(const IntegrationRule *ir) { IntRule = ir; }
   void SetIntegrationRule(const IntegrationRule &ir) { IntRule = &ir; }
   const IntegrationRule *GetIntegrationRule() const { return IntRule; }
   virtual void AssemblePA(const FiniteElementSpace &fes) = 0;
   /// y += A_E x on E-vectors (MFEM layout), the dispatch-table level of the C ABI
   virtual void AddMultPA(const Vector &x, Vector &y) const = 0;
   virtual void AddMultTransposePA(const Vector &x, Vector &y) const { AddMultPA(x, y); }
   virtual void AssembleDiagonalPA(Vector &diag) = 0;
   const Vector &GetPAData() const { return pa_data; }
};
/// fem/bilininteg.hpp:2169-2360; PA in fem/integ/bilininteg_diffusion_pa.cpp.
class DiffusionIntegrator : public BilinearFormIntegrator
{
   Coefficient *Q; VectorCoefficient *VQ = nullptr; MatrixCoefficient *MQ = nullptr;
public:
   DiffusionIntegrator(const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr) {}
   DiffusionIntegrator(Coefficient &q, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(&q) {}
   /// diag(q) as the diffusion tensor (fem/bilininteg.hpp:2282-2291)
   DiffusionIntegrator(VectorCoefficient &q, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr), VQ(&q) {}
   /// full or symmetric matrix coefficient (fem/bilininteg.hpp:2293-2302)
   DiffusionIntegrator(MatrixCoefficient &q, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr), MQ(&q) {}
   /// the operator is symmetric only when D is; otherwise aborts as the reference does
   /// (fem/integ/bilininteg_diffusion_pa.cpp:76-87)
   void AddMultTransposePA(const Vector &x, Vector &y) const override;
   static const IntegrationRule &GetRule(const FiniteElement &trial_fe, const FiniteElement &test_fe);
   void AssemblePA(const FiniteElementSpace &fes) override;
   void AddMultPA(const Vector &x, Vector &y) const override;
   void AssembleDiagonalPA(Vector &diag) override;
};
/// fem/bilininteg.hpp:2376-2460; PA in fem/integ/bilininteg_mass_pa.cpp.
class MassIntegrator : public BilinearFormIntegrator
{
   Coefficient *Q;
public:
   MassIntegrator(const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr) {}
   MassIntegrator(Coefficient &q, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(&q) {}
   /// order 2p + OrderW of the nodal transformation (fem/bilininteg.cpp:1450-1462)
   static const IntegrationRule &GetRule(const FiniteElement &trial_fe, const FiniteElement &test_fe, int nodes_order);
   void AssemblePA(const FiniteElementSpace &fes) override;
   void AddMultPA(const Vector &x, Vector &y) const override;
   void AssembleDiagonalPA(Vector &diag) override;
};

/// fem/bilininteg.hpp VectorDiffusionIntegrator; PA in fem/integ/bilininteg_vecdiffusion_pa.cpp: the scalar
/// operator on every component of a vector H1 space (scalar coefficient; CEED BP4/BP6).
class VectorDiffusionIntegrator : public BilinearFormIntegrator
{
   Coefficient *Q; VectorCoefficient *VQ = nullptr;
public:
   VectorDiffusionIntegrator(const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr) { vector_integrator = true; }
   VectorDiffusionIntegrator(Coefficient &q, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(&q) { vector_integrator = true; }
   /// one diffusion coefficient per component (fem/integ/bilininteg_vecdiffusion_pa.cpp:62-67)
   VectorDiffusionIntegrator(VectorCoefficient &vq, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr), VQ(&vq) { vector_integrator = true; }
   void AssemblePA(const FiniteElementSpace &fes) override;
   void AddMultPA(const Vector &x, Vector &y) const override;
   void AssembleDiagonalPA(Vector &diag) override;
};
/// fem/bilininteg.hpp VectorMassIntegrator; PA in fem/integ/bilininteg_vecmass_pa.cpp (CEED BP2).
class VectorMassIntegrator : public BilinearFormIntegrator
{
   Coefficient *Q; VectorCoefficient *VQ = nullptr;
public:
   VectorMassIntegrator(const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr) { vector_integrator = true; }
   VectorMassIntegrator(Coefficient &q, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(&q) { vector_integrator = true; }
   /// diagonal (per-component) coefficient (fem/bilininteg.hpp VectorMassIntegrator(VectorCoefficient&))
   VectorMassIntegrator(VectorCoefficient &vq, const IntegrationRule *ir = nullptr) : BilinearFormIntegrator(ir), Q(nullptr), VQ(&vq) { vector_integrator = true; }
   void AssemblePA(const FiniteElementSpace &fes) override;
   void AddMultPA(const Vector &x, Vector &y) const override;
   void AssembleDiagonalPA(Vector &diag) override;
};

/// fem/bilinearform_ext.hpp:67-144, bilinearform_ext.cpp:256-564.
class PABilinearFormExtension : public Operator
{
   BilinearForm *a; const FiniteElementSpace *fes;
   const ElementRestriction *elem_restrict = nullptr;
   mutable Vector localX, localY;
   friend class ConstrainedOperator;
   friend class CGSolver;
public:
   explicit PABilinearFormExtension(BilinearForm *form);
   void Assemble();
   void AssembleDiagonal(Vector &diag) const override;
   void Mult(const Vector &x, Vector &y) const override;
   void MultTranspose(const Vector &x, Vector &y) const override { Mult(x, y); }
   /// single-integrator fast path description for the fused kernels (nullptr otherwise)
   Array<int> elem_attributes;            // bilinearform_ext.cpp:290-300
   mutable Vector tmp_evec;
   bool HasMarkers() const;
   const BilinearFormIntegrator *SingleIntegrator() const;
   const ElementRestriction *GetElementRestriction() const { return elem_restrict; }
   /// description of the operator for the fused kernels; false unless one assembled integrator
   bool FillOperator(mfem_amd_pa_operator &op) const;
};

class BilinearForm : public Operator
{
   FiniteElementSpace *fes; AssemblyLevel assembly = AssemblyLevel::LEGACY;
   std::vector<BilinearFormIntegrator*> domain_integs;
   std::vector<Array<int>*> domain_markers;       // per integrator: NULL or the caller's attribute marker (not owned)
   PABilinearFormExtension *ext = nullptr;
   friend class PABilinearFormExtension;
public:
   explicit BilinearForm(FiniteElementSpace *f) : Operator(f->GetVSize()), fes(f) {}
   ~BilinearForm() override;
   FiniteElementSpace *FESpace() { return fes; }
   const FiniteElementSpace *FESpace() const { return fes; }
   void SetAssemblyLevel(AssemblyLevel assembly_level);                       // bilinearform.cpp:109-136
   AssemblyLevel GetAssemblyLevel() const { return assembly; }
   const PABilinearFormExtension *GetPAExtension() const { return ext; }
   void EnableStaticCondensation() { MFEM_ABORT("mfem_amd: static condensation is outside the partial-assembly path"); }
   void EnableSparseMatrixSorting(bool) {}
   void AddDomainIntegrator(BilinearFormIntegrator *bfi) { domain_integs.push_back(bfi); domain_markers.push_back(nullptr); }
   /// restricted to the elements whose attribute is marked (fem/bilinearform.cpp AddDomainIntegrator(bfi, elem_marker));
   /// the marker array is referenced, not copied, as in the reference
   void AddDomainIntegrator(BilinearFormIntegrator *bfi, Array<int> &elem_marker) { domain_integs.push_back(bfi); domain_markers.push_back(&elem_marker); }
   void Assemble(int skip_zeros = 1);
   void Mult(const Vector &x, Vector &y) const override;
   void MultTranspose(const Vector &x, Vector &y) const override;
   void AssembleDiagonal(Vector &diag) const override;
   void FormLinearSystem(const Array<int> &ess_tdof_list, Vector &x, Vector &b, OperatorHandle &A,
                         Vector &X, Vector &B, int copy_interior = 0);
   void FormSystemMatrix(const Array<int> &ess_tdof_list, OperatorHandle &A);
   void RecoverFEMSolution(const Vector &X, const Vector &b, Vector &x) override;
   void Update() { delete ext; ext = nullptr; height = width = fes->GetVSize(); if (assembly == AssemblyLevel::PARTIAL) { ext = new PABilinearFormExtension(this); } }
};

// ---- types named by ex1 on branches this path never takes (they abort when used) --------
class SparseMatrix : public Operator
{ public: void Mult(const Vector &, Vector &) const override { MFEM_ABORT("mfem_amd: assembled matrices are outside the partial-assembly path"); } };
class GSSmoother : public Solver
{
public:
   explicit GSSmoother(const SparseMatrix &, int = 0, int = 1) { MFEM_ABORT("mfem_amd: GSSmoother needs an assembled matrix; use -pa"); }
   void SetOperator(const Operator &) override {}
   void Mult(const Vector &, Vector &) const override {}
};
namespace ceed
{
class AlgebraicSolver : public Solver
{
public:
   AlgebraicSolver(BilinearForm &, const Array<int> &) { MFEM_ABORT("mfem_amd: libCEED algebraic solver is outside this path"); }
   void SetOperator(const Operator &) override {}
   void Mult(const Vector &, Vector &) const override {}
};
}


// ---- parallel layer: one process per GPU (mesh/pmesh.hpp, fem/pfespace.hpp, fem/pbilinearform.hpp, ...) --------
// Vectors of a parallel space live on the rank's local dofs; entries of dofs another rank owns (ghosts) are kept at
// zero between operator applications, so "true-dof vectors" and local vectors have one size and no T<->L copies
// exist (the reference copies through ltdof_ldof, fem/pfespace.cpp:5385-5391).  The transport is RCCL
// (include/mfem_amd_comm.h).

/// general/communication.hpp `Mpi`: the process group comes from the launcher's environment (RANK, WORLD_SIZE,
/// LOCAL_RANK, MASTER_ADDR, MASTER_PORT -- torch.distributed.run or any launcher that sets them).
class Mpi
{
public:
   static void Init() {}
   static void Init(int &, char **&) {}
   static int WorldRank();
   static int WorldSize();
   static bool Root() { return WorldRank() == 0; }
   static void Finalize();
   /// the communicator, created on first use (after the Device is configured: RCCL binds to the current GPU)
   static mfem_amd_world *World();
};
class Hypre { public: static void Init() {} };

/// mesh/pmesh.hpp: the rank's part of an element partition of a serial mesh.  The local part is a Mesh of its own
/// (this object); the serial mesh and the partition stay available for the global dof numbering.
class ParMesh : public Mesh
{
   MPI_Comm comm; int rank, nranks;
   Mesh serial; std::vector<int> partition, local_elems;
   int n_interior = 0;
   void BuildLocal();
public:
   /// partitioning: rank of every element, or NULL = contiguous chunks of the element order (METIS, the
   /// reference's default mesh/mesh.cpp:9106-9195, is not on the target image).  `mesh` is emptied (moved from).
   ParMesh(MPI_Comm comm, Mesh &mesh, const int *partitioning = nullptr);
   MPI_Comm GetComm() const { return comm; }
   int GetMyRank() const { return rank; }
   int GetNRanks() const { return nranks; }
   long long GetGlobalNE() const { return serial.GetNE(); }
   /// mesh/pmesh.cpp UniformRefinement: children inherit their parent's rank
   void UniformRefinement(int ref_algo = 0);
   const Mesh &SerialMesh() const { return serial; }
   const std::vector<int> &LocalElements() const { return local_elems; }
   /// the local elements are listed interior first: the first InteriorElements() of them touch no vertex of another rank's
   /// elements, so nothing they read comes through the shared-dof exchange (it can run beside them)
   int InteriorElements() const { return n_interior; }
   const std::vector<int> &Partitioning() const { return partition; }
};

/// fem/pfespace.hpp: local H1 space + the shared-dof tables of DeviceConformingProlongationOperator
/// (fem/pfespace.cpp:5266-5345) + the exchange.
class ParFiniteElementSpace : public FiniteElementSpace
{
   ParMesh *pmesh;
   int64_t global_ndofs = 0;                    // dofs of the serial space (never built: its numbering is evaluated
                                                // element by element for the rank's elements and their neighbours)
   std::vector<int64_t> l2g64;                  // local dof -> global dof
   std::vector<std::pair<int64_t, int>> g2l;    // (global dof, local dof) of the rank's dofs, ascending
   int LocalDof(int64_t g) const;               // -1: not a dof of this rank
   Array<int> shr_ltdof, ext_ldof, unq_ltdof, unq_shr_i, unq_shr_j;
   std::vector<int> up_ranks, up_sizes, lo_ranks, lo_sizes;
   mfem_amd_exchange *exchange = nullptr;
   mfem_amd_comm cb;
   int n_owned = 0;
   void BuildSharedTables();
public:
   ParFiniteElementSpace(ParMesh *pm, const FiniteElementCollection *fec, int vdim = 1, int ordering = Ordering::byNODES);
   ~ParFiniteElementSpace();
   ParMesh *GetParMesh() const { return pmesh; }
   MPI_Comm GetComm() const { return pmesh->GetComm(); }
   HYPRE_BigInt GlobalTrueVSize() const { return global_ndofs*GetVDim(); }
   HYPRE_BigInt GlobalVSize() const { return global_ndofs*GetVDim(); }
   int GetTrueVSize() const { return GetVSize(); }
   int GetNOwned() const { return n_owned; }
   const std::vector<int64_t> &LocalToGlobal() const { return l2g64; }
   const Array<int> &GhostDofs() const { return ext_ldof; }
   /// fem/pfespace.cpp GetEssentialTrueDofs: the marker is synchronised over the sharing ranks there; here every rank
   /// walks the serial mesh's boundary elements that touch its vertices.  Ghost copies of essential dofs are in the list too.
   void GetEssentialTrueDofs(const Array<int> &bdr_attr_is_ess, Array<int> &ess_tdof_list, int component = -1) const;
   const mfem_amd_comm *Comm() const { return &cb; }
   void Forward(Vector &v) const;               // P: ghosts := owner values
   void Reverse(Vector &v) const;               // P^T: owners += ghost contributions; ghosts := 0
};

class ParGridFunction : public GridFunction
{
   ParFiniteElementSpace *pfes;
public:
   explicit ParGridFunction(ParFiniteElementSpace *f) : GridFunction(f), pfes(f) {}
   ParGridFunction &operator=(real_t v) { Vector::operator=(v); return *this; }
   ParGridFunction &operator=(const Vector &v) { Vector::operator=(v); return *this; }
   ParFiniteElementSpace *ParFESpace() const { return pfes; }
};

class ParLinearForm : public LinearForm
{
   ParFiniteElementSpace *pfes;
public:
   explicit ParLinearForm(ParFiniteElementSpace *f) : LinearForm(f), pfes(f) {}
   ParFiniteElementSpace *ParFESpace() const { return pfes; }
   /// fem/plinearform.cpp ParallelAssemble: tv = P^T b
   void ParallelAssemble(Vector &tv) const { tv = *this; pfes->Reverse(tv); }
};

/// RAPOperator(P, ConstrainedOperator(A), P) of the PA form (linalg/operator.cpp:433, 586-646) on the local vectors
class ParPAOperator : public Operator
{
   const ParFiniteElementSpace *pfes;
   const PABilinearFormExtension *pa;
   mutable mfem_amd_pa_operator cop;
   Array<int> ess; Array<unsigned char> marker; Array<int> masked_gather;
   void RefreshOperator() const;
   void SetSplit() const;
   friend class CGSolver;
public:
   ParPAOperator(const PABilinearFormExtension &ext, const ParFiniteElementSpace &pfes, const Array<int> &ess_tdof_list);
   /// the ghost entries of x are overwritten with their owners' values (P is applied in place)
   void Mult(const Vector &x, Vector &y) const override;
   const Array<int> &GetEssentialTrueDofs() const { return ess; }
};

/// fem/pbilinearform.hpp, PARTIAL assembly level only
class ParBilinearForm : public BilinearForm
{
   ParFiniteElementSpace *pfes;
public:
   explicit ParBilinearForm(ParFiniteElementSpace *f) : BilinearForm(f), pfes(f) {}
   ParFiniteElementSpace *ParFESpace() const { return pfes; }
   /// y = P^T A P x
   void Mult(const Vector &x, Vector &y) const override;
   /// fem/pbilinearform.cpp:293-316: local diagonal, one P^T exchange; ghost entries read +infinity (1/diag = 0)
   void AssembleDiagonal(Vector &diag) const override;
   void FormLinearSystem(const Array<int> &ess_tdof_list, Vector &x, Vector &b, OperatorHandle &A,
                         Vector &X, Vector &B, int copy_interior = 0);
   /// x = P X
   void RecoverFEMSolution(const Vector &X, const Vector &b, Vector &x) override;
};

} // namespace mfem
#endif

// general.cpp -- errors, device memory, Device, OptionsParser, 1D rules and basis tables.
#include "mfem.hpp"
#include <hip/hip_runtime_api.h>
#include <cstdio>

namespace mfem
{

std::ostream &out = std::cout;
std::ostream &err = std::cerr;

// general/error.cpp: print the message and abort.
void mfem_error(const char *msg)
{
   std::cerr << "\n\n" << (msg ? msg : "MFEM abort") << "\n\nMFEM abort: aborting!\n" << std::flush;
   std::abort();
}
void mfem_abort_at(const std::string &msg, const char *file, int line)
{
   std::cerr << "\n\n" << msg << "\n ... in file: " << file << ':' << line
             << "\n\nMFEM abort: aborting!\n" << std::flush;
   std::abort();
}

namespace internal
{
// general/hip.hpp:32-39 (MFEM_GPU_CHECK): a failing HIP call aborts with its message.
#define HOST_HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { MFEM_ABORT("HIP error: " << hipGetErrorString(e_) << " in " #x); } } while (0)

void *DeviceAlloc(size_t bytes)
{
   Device::Require();
   void *p = nullptr;
   HOST_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
   return p;
}
void DeviceFree(void *p) { if (p) { (void)hipFree(p); } }
void CopyToDevice(void *dst, const void *src, size_t bytes)
{ if (bytes) { HOST_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); } }
void CopyToHost(void *dst, const void *src, size_t bytes)
{ if (bytes) { HOST_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); } }
void CopyOnDevice(void *dst, const void *src, size_t bytes)
{ if (bytes) { HOST_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); } }
// rows of `width` bytes, `spitch` apart at the source, to rows `dpitch` apart at the destination (device to device)
void CopyRowsOnDevice(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows)
{ if (width && rows) { HOST_HIP_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToDevice, 0)); } }
void DeviceSync() { HOST_HIP_CHECK(hipStreamSynchronize(0)); }
}

// ---- Device ----------------------------------------------------------------------------
bool Device::configured = false;
int Device::dev_id = 0;

void Device::Configure(const std::string &config, int dev)
{
   std::string name = config;
   size_t colon = name.find(':');
   if (colon != std::string::npos) { dev = std::atoi(name.c_str() + colon + 1); name = name.substr(0, colon); }
   else if (dev == 0 && getenv("LOCAL_RANK"))
   {
      // one process per GPU: without an explicit index a rank takes the device of its LOCAL_RANK
      int nd = 0;
      if (hipGetDeviceCount(&nd) == hipSuccess && nd > 0) { dev = std::atoi(getenv("LOCAL_RANK")) % nd; }
   }
   if (name == "cpu" || name == "omp" || name == "debug")
   {
      MFEM_ABORT("mfem_amd: device '" << config << "' requested, but this library computes on the "
                 "MI355X (HIP) only and has no CPU path: run with '-d hip'");
   }
   MFEM_VERIFY(name == "hip" || name == "gpu" || name == "cuda", "mfem_amd: unknown device '" << config << "'");
   MFEM_VERIFY(!configured || dev == dev_id, "Device is already configured");
   if (configured) { return; }
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= dev)
   { MFEM_ABORT("mfem_amd: no HIP device " << dev << " (" << ndev << " visible)"); }
   MFEM_AMD_CALL(mfem_amd_init(dev));
   dev_id = dev; configured = true;
}
void Device::Require() { if (!configured) { Device d; d.Configure("hip", 0); } }
void Device::Print(std::ostream &os)
{
   os << "Device configuration: hip";
   if (configured)
   {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev_id) == hipSuccess)
      { os << " (device " << dev_id << ": " << prop.name << ", " << prop.gcnArchName << ")"; }
   }
   os << "\nMemory configuration: host-std,hip\n";
}

// ---- OptionsParser (general/optparser.cpp) -------------------------------------------
void OptionsParser::Parse()
{
   std::vector<bool> seen(options.size(), false);
   for (int i = 1; i < argc; )
   {
      const std::string a = argv[i];
      if (a == "-h" || a == "--help") { error_type = 1; return; }
      bool found = false;
      for (size_t j = 0; j < options.size() && !found; j++)
      {
         Option &o = options[j];
         if (o.type == 0)
         {
            if (a == o.s || a == o.l) { *(bool*)o.var = true; }
            else if (a == o.ns || a == o.nl) { *(bool*)o.var = false; }
            else { continue; }
            found = true; seen[j] = true; i++;
         }
         else if (a == o.s || a == o.l)
         {
            if (i + 1 >= argc) { error_type = 4; error_idx = i; return; }
            const char *v = argv[i + 1]; char *end = nullptr;
            if (o.type == 1) { *(int*)o.var = (int)std::strtol(v, &end, 10); }
            else if (o.type == 2) { *(real_t*)o.var = std::strtod(v, &end); }
            else { *(const char**)o.var = v; end = (char*)v + std::strlen(v); }
            if (end == v || *end) { error_type = 5; error_idx = i + 1; return; }
            found = true; seen[j] = true; i += 2;
         }
      }
      if (!found) { error_type = 2; error_idx = i; return; }
   }
   for (size_t j = 0; j < options.size(); j++)
   { if (options[j].required && !seen[j]) { error_type = 6; error_idx = (int)j; return; } }
}
void OptionsParser::ParseCheck(std::ostream &os)
{
   Parse();
   if (!Good()) { PrintUsage(os); std::exit(Help() ? 0 : 1); }
   PrintOptions(os);
}
void OptionsParser::WriteValue(const Option &o, std::ostream &os) const
{
   if (o.type == 1) { os << *(int*)o.var; }
   else if (o.type == 2) { os << *(real_t*)o.var; }
   else if (o.type == 3) { os << *(const char**)o.var; }
}
void OptionsParser::PrintOptions(std::ostream &os) const
{
   os << "Options used:\n";
   for (const Option &o : options)
   {
      if (o.type == 0) { os << "   " << (*(bool*)o.var ? o.l : o.nl) << '\n'; }
      else { os << "   " << o.l << ' '; WriteValue(o, os); os << '\n'; }
   }
}
void OptionsParser::PrintError(std::ostream &os) const
{
   static const char *what[] = {"", "", "Unrecognized option: ", "", "Missing argument for the last option: ",
                                "Invalid argument: ", "Missing required option: "
                               };
   if (error_type >= 2)
   { os << what[error_type] << (error_type == 6 ? options[error_idx].l.c_str() : argv[error_idx]) << "\n\n"; }
}
void OptionsParser::PrintHelp(std::ostream &os) const
{
   os << "Usage: " << argv[0] << " [options] ...\nOptions:\n   -h, --help\n\tPrint this help message and exit.\n";
   for (const Option &o : options)
   {
      os << "   " << o.s << (o.type ? " <value>" : "") << ", " << o.l << (o.type ? " <value>" : "");
      if (o.type == 0) { os << ", " << o.ns << ", " << o.nl; }
      os << ", current value: ";
      if (o.type == 0) { os << (*(bool*)o.var ? o.l : o.nl); } else { WriteValue(o, os); }
      os << "\n\t" << o.descr << '\n';
   }
}
void OptionsParser::PrintUsage(std::ostream &os) const { PrintError(os); PrintHelp(os); }

// ---- 1D quadrature points on [0,1] (fem/intrules.cpp:424-510 and :512-642) ------------
// Newton iteration on the Legendre recurrence, started from the Chebyshev guess; the last
// pass recomputes the polynomial at the converged point so the weight uses it.
void GaussLegendrePoints(int np, real_t *x, real_t *w)
{
   switch (np)
   {
      case 1: x[0] = 0.5; w[0] = 1.0; return;
      case 2:
         x[0] = 0.21132486540518711775; x[1] = 0.78867513459481288225;
         w[0] = w[1] = 0.5; return;
      case 3:
         x[0] = 0.11270166537925831148; x[1] = 0.5; x[2] = 0.88729833462074168852;
         w[0] = w[2] = 5./18.; w[1] = 4./9.; return;
   }
   const int half = (np + 1)/2;
   for (int i = 1; i <= half; i++)
   {
      real_t z = std::cos(M_PI*(i - 0.25)/(np + 0.5)), dp = 0, xi = 0;
      for (bool last = false; ; )
      {
         real_t pm = 1, pk = z;
         for (int j = 2; j <= np; j++) { const real_t pmm = pm; pm = pk; pk = ((2*j - 1)*z*pm - (j - 1)*pmm)/j; }
         dp = np*(z*pk - pm)/(z*z - 1);
         if (last) { break; }
         const real_t dz = pk/dp;
         if (std::fabs(dz) < 1e-16) { last = true; xi = ((1 - z) + dz)/2; }
         z -= dz;
      }
      x[i-1] = xi; x[np-i] = 1 - xi;
      w[i-1] = w[np-i] = 1./(4*xi*(1 - xi)*dp*dp);
   }
}
void GaussLobattoPoints(int np, real_t *x, real_t *w)
{
   if (np == 1) { x[0] = 0.5; w[0] = 1.0; return; }
   x[0] = 0.0; x[np-1] = 1.0;
   w[0] = w[np-1] = 1.0/(np*(np - 1));
   for (int i = 1; i <= (np - 1)/2; ++i)
   {
      real_t xi = std::sin(M_PI*((real_t)i/(np - 1) - 0.5)), zi = 0, pl = 0;
      bool last = false;
      for (int iter = 0; ; ++iter)
      {
         real_t plm = 1.0; pl = xi;
         for (int l = 1; l < np - 1; ++l) { const real_t nx = ((2*l + 1)*xi*pl - l*plm)/(l + 1); plm = pl; pl = nx; }
         if (last) { break; }
         const real_t dx = (xi*pl - plm)/(np*pl);
         if (std::fabs(dx) < 1e-16) { last = true; zi = ((1.0 + xi) - dx)/2; }
         if (iter >= 8) { break; }
         xi -= dx;
      }
      x[i] = zi; w[i] = 1.0/(np*(np - 1)*pl*pl);
      x[np-1-i] = 1.0 - zi; w[np-1-i] = w[i];
   }
}

// Barycentric Lagrange evaluation with derivative (fem/fe/fe_base.cpp:1754-1772 weights,
// :1861-1928 evaluation), tabulated into the DofToQuad TENSOR layout.
void LagrangeBasisTables(int p, const real_t *x, int nq, const real_t *pts, real_t *B, real_t *G)
{
   std::vector<real_t> w(p + 1, 1.0), u(p + 1), d(p + 1);
   for (int i = 0; i <= p; i++)
      for (int j = 0; j < i; j++) { const real_t xij = x[i] - x[j]; w[i] *= xij; w[j] *= -xij; }
   for (int i = 0; i <= p; i++) { w[i] = 1.0/w[i]; }
   for (int q = 0; q < nq; q++)
   {
      const real_t y = pts[q];
      if (p == 0) { u[0] = 1.0; d[0] = 0.0; }
      else
      {
         // k = index of the node nearest to y; lk = prod_{i != k} (y - x_i)
         int k = 0; real_t lk = 1.0;
         for (; k < p; k++)
         {
            if (y >= (x[k] + x[k+1])/2) { lk *= y - x[k]; }
            else { for (int i = k + 1; i <= p; i++) { lk *= y - x[i]; } break; }
         }
         const real_t l = lk*(y - x[k]);
         real_t sk = 0.0;
         for (int i = 0; i <= p; i++)
         {
            if (i == k) { u[k] = lk*w[k]; continue; }
            const real_t si = 1.0/(y - x[i]);
            sk += si; u[i] = l*si*w[i];
         }
         const real_t lp = l*sk + lk;
         for (int i = 0; i <= p; i++)
         { d[i] = (i == k) ? sk*u[k] : (lp*w[i] - u[i])/(y - x[i]); }
      }
      for (int i = 0; i <= p; i++) { B[q + nq*i] = u[i]; G[q + nq*i] = d[i]; }
   }
}

// ---- IntegrationRule(s) ----------------------------------------------------------------
const Array<real_t> &IntegrationRule::GetWeights() const
{
   if (weights.Size() != GetNPoints())
   {
      weights.SetSize(GetNPoints());
      for (int i = 0; i < GetNPoints(); i++) { weights[i] = pts[i].weight; }
   }
   return weights;
}
IntegrationRules IntRules(0, Quadrature1D::GaussLegendre);
IntegrationRules::~IntegrationRules()
{
   for (auto *r : seg) { delete r; }
   for (auto *r : cube) { delete r; }
}
// fem/intrules.cpp:1181-1251 (segment: RealOrder and point count per family) and
// :1600- (cube = tensor product, x fastest, :56-84).
const IntegrationRule &IntegrationRules::Get(int GeomType, int Order) const
{
   MFEM_VERIFY(GeomType == Geometry::SEGMENT || GeomType == Geometry::CUBE,
               "mfem_amd: integration rules exist for SEGMENT and CUBE only");
   if (Order < 0) { Order = 0; }
   const int RealOrder = Order | 1;   // rules are stored per odd order
   int n = 0;
   if (quad_type == Quadrature1D::GaussLegendre) { n = Order/2 + 1; }
   else if (quad_type == Quadrature1D::GaussLobatto) { n = Order/2 + 2; }
   else { MFEM_ABORT("unknown quadrature type"); }
   std::vector<IntegrationRule*> &cache = (GeomType == Geometry::SEGMENT) ? seg : cube;
   if ((int)cache.size() <= RealOrder) { cache.resize(RealOrder + 1, nullptr); }
   if (cache[RealOrder]) { return *cache[RealOrder]; }
   std::vector<real_t> x(n), w(n);
   if (quad_type == Quadrature1D::GaussLegendre) { GaussLegendrePoints(n, x.data(), w.data()); }
   else { GaussLobattoPoints(n, x.data(), w.data()); }
   IntegrationRule *ir;
   if (GeomType == Geometry::SEGMENT)
   {
      ir = new IntegrationRule(n);
      for (int i = 0; i < n; i++) { ir->IntPoint(i).x = x[i]; ir->IntPoint(i).weight = w[i]; }
   }
   else
   {
      ir = new IntegrationRule(n*n*n);
      for (int iz = 0; iz < n; iz++)
         for (int iy = 0; iy < n; iy++)
            for (int ix = 0; ix < n; ix++)
            {
               IntegrationPoint &ip = ir->IntPoint((iz*n + iy)*n + ix);
               ip.x = x[ix]; ip.y = x[iy]; ip.z = x[iz];
               ip.weight = w[ix]*w[iy]*w[iz];
            }
   }
   ir->n1d = n; ir->quad_type = quad_type; ir->x1d = x; ir->w1d = w;
   ir->SetOrder(quad_type == Quadrature1D::GaussLegendre ? 2*n - 1 : 2*n - 3);
   cache[RealOrder] = ir;
   return *ir;
}

} // namespace mfem

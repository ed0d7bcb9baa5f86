// Runtime glue: error reporting, device selection, fallback-kernel reporter.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <set>
#include <utility>

namespace mfa
{

static thread_local char g_err[512] = "";
static int g_report = -1;   // -1: read the environment on first use

void set_error(const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof g_err, fmt, ap);
   va_end(ap);
}

int dot_scratch_setup(int dev);      // vector_cg.hip: per-device address of the no-alloc reduction scratch

int check_hip(hipError_t e, const char *what)
{
   if (e == hipSuccess) { return 0; }
   set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
   return (int)e ? (int)e : 1;
}

bool report_kernels_enabled()
{
   if (g_report < 0)
   {
      const char *v = getenv("MFEM_REPORT_KERNELS");
      g_report = (v && *v && strcmp(v, "0") != 0 && strcasecmp(v, "NO") != 0) ? 1 : 0;
   }
   return g_report > 0;
}

// First use of a non-specialised kernel, printed once per (name, D1D, Q1D), in the spirit
// of KernelReporter (fem/kernel_reporter.hpp:54-98).
void report_fallback(const char *name, int D1D, int Q1D)
{
   if (!report_kernels_enabled()) { return; }
   static std::mutex mtx;
   static std::set<std::pair<std::string, int>> seen;
   std::lock_guard<std::mutex> lk(mtx);
   if (seen.insert({name, D1D*100 + Q1D}).second)
   {
      fprintf(stderr, "mfem_amd: fallback kernel %s used for (DIM,D1D,Q1D) = (3,%d,%d)\n",
              name, D1D, Q1D);
   }
}

// ---- roctx ranges, resolved at run time (common.h TraceRange) ----
static int g_trace = -1;                       // -1: not looked at yet; 0: off; 1: on
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)(void) = nullptr;

static void trace_setup()
{
   static std::mutex mtx;
   std::lock_guard<std::mutex> lk(mtx);
   if (g_trace >= 0) { return; }
   const char *v = getenv("MFEM_AMD_TRACE");
   if (!(v && *v && strcmp(v, "0") != 0)) { g_trace = 0; return; }
   const char *libs[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
   for (const char *name : libs)
   {
      void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) { continue; }
      g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      g_roctx_pop = reinterpret_cast<int (*)(void)>(dlsym(h, "roctxRangePop"));
      if (g_roctx_push && g_roctx_pop) { g_trace = 1; return; }
   }
   fprintf(stderr, "mfem_amd: MFEM_AMD_TRACE is set but no roctx library could be loaded; ranges are off\n");
   g_trace = 0;
}

void trace_push(const char *name)
{
   if (g_trace < 0) { trace_setup(); }
   if (g_trace > 0) { g_roctx_push(name); }
}

void trace_pop()
{
   if (g_trace > 0) { g_roctx_pop(); }
}

} // namespace mfa

extern "C" const char *mfem_amd_last_error(void) { return mfa::g_err; }

extern "C" void mfem_amd_report_kernels(int on) { mfa::g_report = on ? 1 : 0; }

extern "C" int mfem_amd_init(int device)
{
   int n = 0;
   if (int rc = mfa::check_hip(hipGetDeviceCount(&n), "hipGetDeviceCount")) { return rc; }
   if (device < 0 || device >= n)
   {
      mfa::set_error("mfem_amd_init: device %d out of range (found %d HIP devices)", device, n);
      return 1;
   }
   if (int rc = mfa::check_hip(hipSetDevice(device), "hipSetDevice")) { return rc; }
   hipDeviceProp_t prop;
   if (int rc = mfa::check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) { return rc; }
   if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
   {
      mfa::set_error("mfem_amd_init: device %d is %s; this library is built for gfx950 (MI355X) only",
                     device, prop.gcnArchName);
      return 1;
   }
   if (device < 64) { if (int rc = mfa::dot_scratch_setup(device)) { return rc; } }
   return 0;
}

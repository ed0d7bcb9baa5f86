// libmfem_amd_comm.so -- the multi-GPU transport behind struct mfem_amd_comm (include/mfem_amd_comm.h):
// process group bootstrap over TCP, shared-dof exchange and all-reduce on RCCL (ncclSend / ncclRecv /
// ncclAllReduce, one process per GPU, every call on the caller's HIP stream), and a host-staged TCP transport
// for tests that put several ranks on one GPU.
//
// Replaces, for the partial-assembly path: GroupCommunicator's MPI_Isend / MPI_Irecv / MPI_Waitall exchange
// (general/communication.cpp:724-1042) as DeviceConformingProlongationOperator drives it
// (fem/pfespace.cpp:5401-5447, 5503-5539) and InnerProduct's MPI_Allreduce (linalg/vector.hpp:773-779).
// Message shapes are the reference's: one message per neighbour and direction, one double per dot product.
// On a 2x2x2 box grid every rank has <= 7 neighbours, each reached over its own xGMI link; the messages are
// small (a face of a 54^3, p = 4 box: 377 KB), so the exchange is latency-bound and RCCL's grouped
// point-to-point calls -- not a ring collective -- are the right primitive.
#include "../../include/mfem_amd_comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace
{

thread_local char g_err[512] = "";

int fail(const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof g_err, fmt, ap);
   va_end(ap);
   return 1;
}

#define COMM_HIP(call) \
   do { hipError_t e_ = (call); if (e_ != hipSuccess) { return fail("%s: HIP error %d (%s)", #call, (int)e_, hipGetErrorString(e_)); } } while (0)
#define COMM_NCCL(call) \
   do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { return fail("%s: RCCL error %d (%s)", #call, (int)r_, ncclGetErrorString(r_)); } } while (0)
#define COMM_ABI(call) \
   do { if ((call) != 0) { return fail("%s: %s", #call, mfem_amd_last_error()); } } while (0)

// ---- blocking socket helpers ----------------------------------------------------------------
int send_all(int fd, const void *buf, size_t n)
{
   const char *p = (const char*)buf;
   while (n > 0)
   {
      const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
      if (k < 0) { if (errno == EINTR) { continue; } return fail("send: %s", strerror(errno)); }
      p += k; n -= (size_t)k;
   }
   return 0;
}

int recv_all(int fd, void *buf, size_t n)
{
   char *p = (char*)buf;
   while (n > 0)
   {
      const ssize_t k = ::recv(fd, p, n, 0);
      if (k < 0) { if (errno == EINTR) { continue; } return fail("recv: %s", strerror(errno)); }
      if (k == 0) { return fail("recv: the peer closed the connection"); }
      p += k; n -= (size_t)k;
   }
   return 0;
}

// Sockets of the rendezvous never wait forever: a peer that died before it connected, or that stopped talking,
// turns into an error on every other rank (the caller falls back or aborts) instead of a hang.
constexpr int PEER_TIMEOUT_S = 300;

void tune(int fd)
{
   int one = 1;
   setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
   timeval tv; tv.tv_sec = PEER_TIMEOUT_S; tv.tv_usec = 0;
   setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
   setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}


// accept() with a deadline
int accept_peer(int listen_fd, sockaddr_in *pa, int seconds)
{
   pollfd pf; pf.fd = listen_fd; pf.events = POLLIN; pf.revents = 0;
   const int rc = ::poll(&pf, 1, seconds*1000);
   if (rc == 0) { errno = ETIMEDOUT; return -1; }
   if (rc < 0) { return -1; }
   socklen_t len = sizeof(sockaddr_in);
   return pa ? ::accept(listen_fd, (sockaddr*)pa, &len) : ::accept(listen_fd, nullptr, nullptr);
}

int listen_on(int port, int *fd_out, int *port_out)
{
   const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
   if (fd < 0) { return fail("socket: %s", strerror(errno)); }
   int one = 1;
   setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
   sockaddr_in a;
   memset(&a, 0, sizeof a);
   a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)port);
   if (::bind(fd, (sockaddr*)&a, sizeof a) != 0) { ::close(fd); return fail("bind(port %d): %s", port, strerror(errno)); }
   if (::listen(fd, 64) != 0) { ::close(fd); return fail("listen: %s", strerror(errno)); }
   socklen_t len = sizeof a;
   getsockname(fd, (sockaddr*)&a, &len);
   *fd_out = fd; *port_out = ntohs(a.sin_port);
   return 0;
}

// connect with retries: the listener may not be up yet (ranks start in any order)
int connect_to(uint32_t addr_be, int port, int *fd_out)
{
   for (int attempt = 0; attempt < 600; attempt++)
   {
      const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) { return fail("socket: %s", strerror(errno)); }
      sockaddr_in a;
      memset(&a, 0, sizeof a);
      a.sin_family = AF_INET; a.sin_addr.s_addr = addr_be; a.sin_port = htons((uint16_t)port);
      if (::connect(fd, (sockaddr*)&a, sizeof a) == 0) { tune(fd); *fd_out = fd; return 0; }
      ::close(fd);
      usleep(100*1000);
   }
   return fail("connect(port %d): no listener after 60 s (%s)", port, strerror(errno));
}

int resolve(const char *host, uint32_t *addr_be)
{
   in_addr a;
   if (inet_pton(AF_INET, host, &a) == 1) { *addr_be = a.s_addr; return 0; }
   addrinfo hints, *res = nullptr;
   memset(&hints, 0, sizeof hints);
   hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
   if (getaddrinfo(host, nullptr, &hints, &res) != 0 || !res) { return fail("cannot resolve host '%s'", host); }
   *addr_be = ((sockaddr_in*)res->ai_addr)->sin_addr.s_addr;
   freeaddrinfo(res);
   return 0;
}

} // namespace

struct mfem_amd_world
{
   int rank = 0, size = 1, transport = MFEM_AMD_TRANSPORT_RCCL;
   ncclComm_t nccl = nullptr;
   int root_fd = -1;                 // ranks > 0: the connection to rank 0
   std::vector<int> fds;             // rank 0: connections to ranks 1..size-1; TCP transport: every peer
   int listen_fd = -1;
};

struct mfem_amd_exchange
{
   mfem_amd_world *world = nullptr;
   std::vector<int> up_ranks, up_sizes, lo_ranks, lo_sizes;
   int64_t nshr = 0, next = 0, nuniq = 0;
   const int *shr_ltdof = nullptr, *ext_ldof = nullptr, *unq_ltdof = nullptr, *unq_shr_i = nullptr, *unq_shr_j = nullptr;
   double *shr_buf = nullptr, *ext_buf = nullptr;            // device
   double *shr_host = nullptr, *ext_host = nullptr;          // pinned host (TCP transport)
   double *red_host = nullptr;                               // pinned host, all-reduce staging (TCP transport)
};

extern "C" const char *mfem_amd_comm_last_error(void) { return g_err; }

extern "C" int mfem_amd_world_create(int rank, int size, const char *addr, int port, int port_offset, int transport,
                                     mfem_amd_world **world)
{
   if (!world) { return fail("world_create: world is NULL"); }
   *world = nullptr;
   if (rank < 0)
   {
      const char *r = getenv("RANK"), *s = getenv("WORLD_SIZE"), *a = getenv("MASTER_ADDR"), *p = getenv("MASTER_PORT");
      rank = r ? atoi(r) : 0; size = s ? atoi(s) : 1;
      addr = a ? a : "127.0.0.1"; port = p ? atoi(p) : 29400;
   }
   if (transport < 0)
   {
      const char *t = getenv("MFEM_AMD_TRANSPORT");
      transport = (t && strcmp(t, "tcp") == 0) ? MFEM_AMD_TRANSPORT_TCP : MFEM_AMD_TRANSPORT_RCCL;
   }
   if (size < 1 || rank < 0 || rank >= size) { return fail("world_create: bad rank %d / size %d", rank, size); }
   if (size > 1 && (port + port_offset < 1 || port + port_offset > 65535)) { return fail("world_create: port %d + offset %d is not a TCP port", port, port_offset); }
   mfem_amd_world *w = new mfem_amd_world;
   w->rank = rank; w->size = size; w->transport = transport;
   w->fds.assign((size_t)size, -1);
   ncclUniqueId id;
   memset(&id, 0, sizeof id);
   if (size > 1)
   {
      const int p0 = port + port_offset;
      if (rank == 0)
      {
         int lport = 0;
         if (listen_on(p0, &w->listen_fd, &lport)) { mfem_amd_world_destroy(w); return 1; }
         // every other rank connects and says who it is and where IT listens (TCP transport: pairwise sockets)
         std::vector<uint32_t> peer_addr((size_t)size, 0), peer_port((size_t)size, 0);
         for (int k = 1; k < size; k++)
         {
            sockaddr_in pa;
            const int fd = accept_peer(w->listen_fd, &pa, 120);
            if (fd < 0) { mfem_amd_world_destroy(w); return fail("accept (%d of %d ranks have not connected within 120 s): %s", size - k, size, strerror(errno)); }
            tune(fd);
            int32_t hello[2];
            if (recv_all(fd, hello, sizeof hello)) { mfem_amd_world_destroy(w); return 1; }
            if (hello[0] < 1 || hello[0] >= size || w->fds[(size_t)hello[0]] >= 0) { mfem_amd_world_destroy(w); return fail("bootstrap: bad or duplicate rank %d", hello[0]); }
            w->fds[(size_t)hello[0]] = fd;
            peer_addr[(size_t)hello[0]] = pa.sin_addr.s_addr; peer_port[(size_t)hello[0]] = (uint32_t)hello[1];
         }
         if (transport == MFEM_AMD_TRANSPORT_RCCL)
         {
            ncclResult_t r = ncclGetUniqueId(&id);
            if (r != ncclSuccess) { mfem_amd_world_destroy(w); return fail("ncclGetUniqueId: %s", ncclGetErrorString(r)); }
         }
         for (int k = 1; k < size; k++)
         {
            if (send_all(w->fds[(size_t)k], &id, sizeof id) || send_all(w->fds[(size_t)k], peer_addr.data(), sizeof(uint32_t)*(size_t)size) ||
                send_all(w->fds[(size_t)k], peer_port.data(), sizeof(uint32_t)*(size_t)size)) { mfem_amd_world_destroy(w); return 1; }
         }
      }
      else
      {
         uint32_t root_addr = 0;
         if (resolve(addr, &root_addr)) { mfem_amd_world_destroy(w); return 1; }
         int lport = 0;
         if (listen_on(0, &w->listen_fd, &lport)) { mfem_amd_world_destroy(w); return 1; }
         if (connect_to(root_addr, p0, &w->root_fd)) { mfem_amd_world_destroy(w); return 1; }
         int32_t hello[2] = {rank, lport};
         std::vector<uint32_t> peer_addr((size_t)size), peer_port((size_t)size);
         if (send_all(w->root_fd, hello, sizeof hello) || recv_all(w->root_fd, &id, sizeof id) ||
             recv_all(w->root_fd, peer_addr.data(), sizeof(uint32_t)*(size_t)size) ||
             recv_all(w->root_fd, peer_port.data(), sizeof(uint32_t)*(size_t)size)) { mfem_amd_world_destroy(w); return 1; }
         w->fds[0] = w->root_fd;
         if (transport == MFEM_AMD_TRANSPORT_TCP)
         {
            // pairwise sockets among ranks >= 1: rank i connects to every j in [1, i), then accepts from the ranks above
            for (int j = 1; j < rank; j++)
            {
               int fd = -1;
               if (connect_to(peer_addr[(size_t)j], (int)peer_port[(size_t)j], &fd)) { mfem_amd_world_destroy(w); return 1; }
               int32_t me = rank;
               if (send_all(fd, &me, sizeof me)) { mfem_amd_world_destroy(w); return 1; }
               w->fds[(size_t)j] = fd;
            }
            for (int k = rank + 1; k < size; k++)
            {
               const int fd = accept_peer(w->listen_fd, nullptr, 120);
               if (fd < 0) { mfem_amd_world_destroy(w); return fail("accept (peer ranks above %d): %s", rank, strerror(errno)); }
               tune(fd);
               int32_t who = -1;
               if (recv_all(fd, &who, sizeof who)) { mfem_amd_world_destroy(w); return 1; }
               if (who <= rank || who >= size || w->fds[(size_t)who] >= 0) { mfem_amd_world_destroy(w); return fail("bootstrap: bad peer rank %d", who); }
               w->fds[(size_t)who] = fd;
            }
         }
      }
   }
   else if (transport == MFEM_AMD_TRANSPORT_RCCL)
   {
      ncclResult_t r = ncclGetUniqueId(&id);
      if (r != ncclSuccess) { mfem_amd_world_destroy(w); return fail("ncclGetUniqueId: %s", ncclGetErrorString(r)); }
   }
   if (transport == MFEM_AMD_TRANSPORT_RCCL)
   {
      ncclResult_t r = ncclCommInitRank(&w->nccl, size, id, rank);
      if (r != ncclSuccess) { mfem_amd_world_destroy(w); return fail("ncclCommInitRank(rank %d of %d): %s", rank, size, ncclGetErrorString(r)); }
   }
   *world = w;
   return 0;
}

extern "C" int mfem_amd_world_rank(const mfem_amd_world *w) { return w ? w->rank : 0; }
extern "C" int mfem_amd_world_size(const mfem_amd_world *w) { return w ? w->size : 1; }
extern "C" int mfem_amd_world_transport(const mfem_amd_world *w) { return w ? w->transport : MFEM_AMD_TRANSPORT_RCCL; }

extern "C" int mfem_amd_world_allreduce_host(mfem_amd_world *w, double *buf, int n, int op)
{
   if (!w || !buf || n < 0) { return fail("world_allreduce_host: bad arguments"); }
   if (w->size == 1 || n == 0) { return 0; }
   const size_t bytes = sizeof(double)*(size_t)n;
   if (w->rank == 0)
   {
      std::vector<double> in((size_t)n);
      for (int k = 1; k < w->size; k++)           // rank order: deterministic
      {
         if (recv_all(w->fds[(size_t)k], in.data(), bytes)) { return 1; }
         for (int i = 0; i < n; i++) { buf[i] = op == 1 ? (in[(size_t)i] > buf[i] ? in[(size_t)i] : buf[i]) : buf[i] + in[(size_t)i]; }
      }
      for (int k = 1; k < w->size; k++) { if (send_all(w->fds[(size_t)k], buf, bytes)) { return 1; } }
      return 0;
   }
   if (send_all(w->root_fd, buf, bytes) || recv_all(w->root_fd, buf, bytes)) { return 1; }
   return 0;
}

extern "C" int mfem_amd_world_barrier(mfem_amd_world *w)
{
   double z = 0.0;
   return mfem_amd_world_allreduce_host(w, &z, 1, 0);
}

extern "C" int mfem_amd_world_destroy(mfem_amd_world *w)
{
   if (!w) { return 0; }
   if (w->nccl) { ncclCommDestroy(w->nccl); }
   for (int fd : w->fds) { if (fd >= 0) { ::close(fd); } }
   if (w->root_fd >= 0 && (w->fds.empty() || w->fds[0] != w->root_fd)) { ::close(w->root_fd); }   // a bootstrap that failed half way
   if (w->listen_fd >= 0) { ::close(w->listen_fd); }
   delete w;
   return 0;
}

// ---- the exchange ------------------------------------------------------------------------------
namespace
{

int64_t total(const std::vector<int> &v) { int64_t s = 0; for (int x : v) { s += x; } return s; }

// One message per neighbour and direction.  RCCL: a grouped ncclSend/ncclRecv on `s`.  TCP: staged through
// pinned host buffers; data flows from the low ranks to the high ones (forward) or back (reverse), and every
// rank receives before it sends, so the blocking sockets cannot form a cycle.
int transfer(mfem_amd_exchange *ex, const double *send_dev, double *send_host, const std::vector<int> &send_ranks,
             const std::vector<int> &send_sizes, double *recv_dev, double *recv_host, const std::vector<int> &recv_ranks,
             const std::vector<int> &recv_sizes, hipStream_t s)
{
   mfem_amd_world *w = ex->world;
   if (send_ranks.empty() && recv_ranks.empty()) { return 0; }
   if (w->transport == MFEM_AMD_TRANSPORT_RCCL)
   {
      COMM_NCCL(ncclGroupStart());
      int64_t o = 0;
      for (size_t k = 0; k < recv_ranks.size(); k++)
      {
         COMM_NCCL(ncclRecv(recv_dev + o, (size_t)recv_sizes[k], ncclDouble, recv_ranks[k], w->nccl, s));
         o += recv_sizes[k];
      }
      o = 0;
      for (size_t k = 0; k < send_ranks.size(); k++)
      {
         COMM_NCCL(ncclSend(send_dev + o, (size_t)send_sizes[k], ncclDouble, send_ranks[k], w->nccl, s));
         o += send_sizes[k];
      }
      COMM_NCCL(ncclGroupEnd());
      return 0;
   }
   const int64_t ns = total(send_sizes), nr = total(recv_sizes);
   if (ns) { COMM_HIP(hipMemcpyAsync(send_host, send_dev, sizeof(double)*(size_t)ns, hipMemcpyDeviceToHost, s)); }
   COMM_HIP(hipStreamSynchronize(s));
   int64_t o = 0;
   for (size_t k = 0; k < recv_ranks.size(); k++)
   {
      if (recv_all(w->fds[(size_t)recv_ranks[k]], recv_host + o, sizeof(double)*(size_t)recv_sizes[k])) { return 1; }
      o += recv_sizes[k];
   }
   o = 0;
   for (size_t k = 0; k < send_ranks.size(); k++)
   {
      if (send_all(w->fds[(size_t)send_ranks[k]], send_host + o, sizeof(double)*(size_t)send_sizes[k])) { return 1; }
      o += send_sizes[k];
   }
   if (nr) { COMM_HIP(hipMemcpyAsync(recv_dev, recv_host, sizeof(double)*(size_t)nr, hipMemcpyHostToDevice, s)); }
   return 0;
}

// P: ghost entries of v := the owner's value (fem/pfespace.cpp:5401-5447)
int cb_forward(void *ctx, double *v, void *stream)
{
   mfem_amd_exchange *ex = (mfem_amd_exchange*)ctx;
   hipStream_t s = (hipStream_t)stream;
   COMM_ABI(mfem_amd_shared_pack(ex->nshr, ex->shr_ltdof, v, ex->shr_buf, stream));
   if (int rc = transfer(ex, ex->shr_buf, ex->shr_host, ex->up_ranks, ex->up_sizes, ex->ext_buf, ex->ext_host,
                         ex->lo_ranks, ex->lo_sizes, s)) { return rc; }
   COMM_ABI(mfem_amd_shared_unpack(ex->next, ex->ext_ldof, ex->ext_buf, v, stream));
   return 0;
}

// P^T in two halves (struct mfem_amd_comm: reverse_begin / reverse_end).  begin: pack the ghost entries of v and move
// them into the owners' receive buffers; v is only read, so the caller may run it on a second stream while the
// transpose-gather is still summing the rows it does not send.
int cb_reverse_begin(void *ctx, const double *v, void *stream)
{
   mfem_amd_exchange *ex = (mfem_amd_exchange*)ctx;
   hipStream_t s = (hipStream_t)stream;
   COMM_ABI(mfem_amd_shared_pack(ex->next, ex->ext_ldof, v, ex->ext_buf, stream));
   // the receive side walks the upper neighbours from the highest rank down in the TCP transport (data flows
   // high -> low); the buffer layout stays ascending, so the accumulate order does not depend on the transport
   if (ex->world->transport == MFEM_AMD_TRANSPORT_TCP)
   {
      mfem_amd_world *w = ex->world;
      const int64_t ns = ex->next, nr = ex->nshr;
      if (ns) { COMM_HIP(hipMemcpyAsync(ex->ext_host, ex->ext_buf, sizeof(double)*(size_t)ns, hipMemcpyDeviceToHost, s)); }
      COMM_HIP(hipStreamSynchronize(s));
      std::vector<int64_t> uo(ex->up_ranks.size() + 1, 0), lo(ex->lo_ranks.size() + 1, 0);
      for (size_t k = 0; k < ex->up_ranks.size(); k++) { uo[k + 1] = uo[k] + ex->up_sizes[k]; }
      for (size_t k = 0; k < ex->lo_ranks.size(); k++) { lo[k + 1] = lo[k] + ex->lo_sizes[k]; }
      for (size_t k = ex->up_ranks.size(); k-- > 0; )
      {
         if (recv_all(w->fds[(size_t)ex->up_ranks[k]], ex->shr_host + uo[k], sizeof(double)*(size_t)ex->up_sizes[k])) { return 1; }
      }
      for (size_t k = ex->lo_ranks.size(); k-- > 0; )
      {
         if (send_all(w->fds[(size_t)ex->lo_ranks[k]], ex->ext_host + lo[k], sizeof(double)*(size_t)ex->lo_sizes[k])) { return 1; }
      }
      if (nr) { COMM_HIP(hipMemcpyAsync(ex->shr_buf, ex->shr_host, sizeof(double)*(size_t)nr, hipMemcpyHostToDevice, s)); }
      return 0;
   }
   return transfer(ex, ex->ext_buf, ex->ext_host, ex->lo_ranks, ex->lo_sizes, ex->shr_buf, ex->shr_host,
                   ex->up_ranks, ex->up_sizes, s);
}

// end: owners += what arrived, in fixed neighbour order; ghosts := 0 (fem/pfespace.cpp:5475-5494, :5503-5539)
int cb_reverse_end(void *ctx, double *v, void *stream)
{
   mfem_amd_exchange *ex = (mfem_amd_exchange*)ctx;
   COMM_ABI(mfem_amd_shared_accumulate(ex->nuniq, ex->unq_ltdof, ex->unq_shr_i, ex->unq_shr_j, ex->shr_buf, v, stream));
   COMM_ABI(mfem_amd_set_subvector(ex->next, ex->ext_ldof, nullptr, 0.0, v, stream));
   return 0;
}

// P^T: owners += the ghost entries of the sharing ranks, in fixed neighbour order; ghosts := 0 (:5503-5539)
int cb_reverse(void *ctx, double *v, void *stream)
{
   if (int rc = cb_reverse_begin(ctx, v, stream)) { return rc; }
   return cb_reverse_end(ctx, v, stream);
}

// in-place sum over ranks of n device doubles (InnerProduct: n = 1)
int cb_allreduce(void *ctx, double *dev_buf, int n, void *stream)
{
   mfem_amd_exchange *ex = (mfem_amd_exchange*)ctx;
   mfem_amd_world *w = ex->world;
   hipStream_t s = (hipStream_t)stream;
   if (w->transport == MFEM_AMD_TRANSPORT_RCCL)
   {
      COMM_NCCL(ncclAllReduce(dev_buf, dev_buf, (size_t)n, ncclDouble, ncclSum, w->nccl, s));
      return 0;
   }
   if (n > 64) { return fail("all-reduce of %d doubles: the TCP transport stages at most 64", n); }
   COMM_HIP(hipMemcpyAsync(ex->red_host, dev_buf, sizeof(double)*(size_t)n, hipMemcpyDeviceToHost, s));
   COMM_HIP(hipStreamSynchronize(s));
   if (mfem_amd_world_allreduce_host(w, ex->red_host, n, 0)) { return 1; }
   COMM_HIP(hipMemcpyAsync(dev_buf, ex->red_host, sizeof(double)*(size_t)n, hipMemcpyHostToDevice, s));
   COMM_HIP(hipStreamSynchronize(s));     // red_host is re-used by the next call
   return 0;
}

} // namespace

extern "C" int mfem_amd_exchange_create(mfem_amd_world *world, int nup, const int *up_ranks, const int *up_sizes,
                                        int nlo, const int *lo_ranks, const int *lo_sizes, const int *shr_ltdof,
                                        const int *ext_ldof, int64_t nuniq, const int *unq_ltdof, const int *unq_shr_i,
                                        const int *unq_shr_j, mfem_amd_exchange **exchange, mfem_amd_comm *comm)
{
   if (!world || !exchange || !comm || nup < 0 || nlo < 0 || nuniq < 0) { return fail("exchange_create: bad arguments"); }
   mfem_amd_exchange *ex = new mfem_amd_exchange;
   ex->world = world;
   ex->up_ranks.assign(up_ranks, up_ranks + nup); ex->up_sizes.assign(up_sizes, up_sizes + nup);
   ex->lo_ranks.assign(lo_ranks, lo_ranks + nlo); ex->lo_sizes.assign(lo_sizes, lo_sizes + nlo);
   for (int k = 0; k < nup; k++)
   {
      if (up_ranks[k] < 0 || up_ranks[k] >= world->size || up_ranks[k] == world->rank || up_sizes[k] < 0 ||
          (k > 0 && up_ranks[k] <= up_ranks[k - 1])) { delete ex; return fail("exchange_create: upper neighbour list must be ascending ranks of the world"); }
   }
   for (int k = 0; k < nlo; k++)
   {
      if (lo_ranks[k] < 0 || lo_ranks[k] >= world->size || lo_ranks[k] == world->rank || lo_sizes[k] < 0 ||
          (k > 0 && lo_ranks[k] <= lo_ranks[k - 1])) { delete ex; return fail("exchange_create: lower neighbour list must be ascending ranks of the world"); }
   }
   ex->nshr = total(ex->up_sizes); ex->next = total(ex->lo_sizes); ex->nuniq = nuniq;
   if ((ex->nshr && !shr_ltdof) || (ex->next && !ext_ldof) || (nuniq && (!unq_ltdof || !unq_shr_i || !unq_shr_j)))
   { delete ex; return fail("exchange_create: null index table"); }
   ex->shr_ltdof = shr_ltdof; ex->ext_ldof = ext_ldof;
   ex->unq_ltdof = unq_ltdof; ex->unq_shr_i = unq_shr_i; ex->unq_shr_j = unq_shr_j;
   const size_t bs = sizeof(double)*(size_t)(ex->nshr > 0 ? ex->nshr : 1), be = sizeof(double)*(size_t)(ex->next > 0 ? ex->next : 1);
   if (hipMalloc((void**)&ex->shr_buf, bs) != hipSuccess || hipMalloc((void**)&ex->ext_buf, be) != hipSuccess)
   { mfem_amd_exchange_destroy(ex); return fail("exchange_create: hipMalloc of the message buffers failed"); }
   if (world->transport == MFEM_AMD_TRANSPORT_TCP)
   {
      if (hipHostMalloc((void**)&ex->shr_host, bs, hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc((void**)&ex->ext_host, be, hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc((void**)&ex->red_host, sizeof(double)*64, hipHostMallocDefault) != hipSuccess)
      { mfem_amd_exchange_destroy(ex); return fail("exchange_create: hipHostMalloc of the staging buffers failed"); }
   }
   comm->ctx = ex;
   comm->allreduce_sum = cb_allreduce;
   comm->forward = cb_forward;
   comm->reverse = cb_reverse;
   comm->reverse_begin = cb_reverse_begin;
   comm->reverse_end = cb_reverse_end;
   *exchange = ex;
   return 0;
}

extern "C" int mfem_amd_exchange_destroy(mfem_amd_exchange *ex)
{
   if (!ex) { return 0; }
   if (ex->shr_buf) { (void)hipFree(ex->shr_buf); }
   if (ex->ext_buf) { (void)hipFree(ex->ext_buf); }
   if (ex->shr_host) { (void)hipHostFree(ex->shr_host); }
   if (ex->ext_host) { (void)hipHostFree(ex->ext_host); }
   if (ex->red_host) { (void)hipHostFree(ex->red_host); }
   delete ex;
   return 0;
}

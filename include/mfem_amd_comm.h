/* mfem_amd_comm.h -- C ABI of the multi-GPU transport of the partial-assembly path: the process group
 * ("world"), and the shared-dof exchange + all-reduce behind struct mfem_amd_comm (mfem_amd.h), written on
 * RCCL (rccl.h: ncclSend / ncclRecv / ncclAllReduce over xGMI), one process per GPU.
 *
 * Reference counterparts (file:line relative to the MFEM source root):
 *   MPI_Init / MPI_Comm_rank / MPI_Comm_size            general/communication.hpp (class Mpi)
 *   GroupCommunicator Bcast / Reduce<Sum>               general/communication.cpp:724-1042
 *   DeviceConformingProlongationOperator::Mult          fem/pfespace.cpp:5401-5447  (forward, P)
 *   DeviceConformingProlongationOperator::MultTranspose fem/pfespace.cpp:5503-5539  (reverse, P^T)
 *   InnerProduct(MPI_Comm, x, y)                        linalg/vector.hpp:773-779   (all-reduce of one double)
 * The library libmfem_amd_comm.so is separate from libmfem_amd.so so that single-GPU users do not load RCCL.
 *
 * Bootstrap: there is no MPI on the target image; ranks find each other through a TCP rendezvous on
 * MASTER_ADDR:MASTER_PORT(+offset) (the environment torch.distributed.run / any launcher sets), which carries
 * the RCCL unique id and the few host-side integers of the set-up.  Transport MFEM_AMD_TRANSPORT_TCP moves the
 * messages themselves over those sockets, staged through the host: for tests with several ranks on ONE GPU
 * (RCCL refuses two ranks on one device) and for debugging -- never for measurements.  The rendezvous listens on
 * every interface, unauthenticated, for the 120 s of the bootstrap: it is meant for the ranks of ONE host (or one
 * trusted cluster network), as torch.distributed's own store is; a bootstrap that fails releases every socket it
 * opened and says which step failed (mfem_amd_comm_last_error).
 */
#ifndef MFEM_AMD_COMM_H
#define MFEM_AMD_COMM_H

#include <stdint.h>
#include "mfem_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfem_amd_world mfem_amd_world;       /* the process group: MPI_COMM_WORLD's stand-in */
typedef struct mfem_amd_exchange mfem_amd_exchange; /* one space's shared-dof exchange */

enum { MFEM_AMD_TRANSPORT_RCCL = 0, MFEM_AMD_TRANSPORT_TCP = 1 };

/* Last error message of the calling thread for the functions of this header (never NULL). */
const char *mfem_amd_comm_last_error(void);

/* Mpi::Init.  rank < 0: take rank / size / address / port from the environment (RANK, WORLD_SIZE, MASTER_ADDR,
 * MASTER_PORT; absent = a world of one).  port_offset is added to MASTER_PORT (torch.distributed's own store
 * listens on MASTER_PORT itself).  The HIP device must be current (mfem_amd_init) before the call: the RCCL
 * communicator is created on it.  transport < 0: MFEM_AMD_TRANSPORT from the environment ("tcp"), else RCCL. */
int mfem_amd_world_create(int rank, int size, const char *addr, int port, int port_offset, int transport,
                          mfem_amd_world **world);
int mfem_amd_world_rank(const mfem_amd_world *world);
int mfem_amd_world_size(const mfem_amd_world *world);
int mfem_amd_world_transport(const mfem_amd_world *world);
/* MPI_Allreduce(MPI_SUM) / MPI_Barrier on HOST values (set-up: global sizes, timing maxima: op 0 sum, 1 max);
 * summed in rank order on rank 0, deterministic. */
int mfem_amd_world_allreduce_host(mfem_amd_world *world, double *buf, int n, int op);
int mfem_amd_world_barrier(mfem_amd_world *world);
/* MPI_Finalize: destroys the communicator and closes the sockets. */
int mfem_amd_world_destroy(mfem_amd_world *world);

/* The exchange of one space.  Tables as DeviceConformingProlongationOperator's (fem/pfespace.cpp:5266-5345),
 * in the convention of mfem_amd.h (vectors live on the local dofs, ghost entries zero between applications):
 *   upper: the neighbours that hold ghost copies of dofs this rank owns -- nup ranks (ascending), up_sizes[k]
 *          entries each; shr_ltdof (device, sum of up_sizes) lists the owned dofs per neighbour, in an order both
 *          sides agree on (ascending global id);
 *   lower: the owners of this rank's ghost dofs -- nlo ranks (ascending), lo_sizes[k]; ext_ldof (device) the
 *          ghost dofs per owner, same order;
 *   unq_ltdof / unq_shr_i / unq_shr_j (device): CSR of the deterministic accumulate (:5475-5494): for every
 *          distinct owned-shared dof the positions of its copies in the receive buffer, ascending.
 * up_ranks, up_sizes, lo_ranks, lo_sizes: HOST.  The exchange allocates its two message buffers (device, and pinned
 * host copies for the TCP transport) here; none of the calls below allocates.  *comm receives the three callbacks of
 * struct mfem_amd_comm (ctx = the exchange): every kernel and every RCCL call they issue is ordered on the
 * `stream` they are handed. */
int mfem_amd_exchange_create(mfem_amd_world *world, int nup, const int *up_ranks, const int *up_sizes,
                             int nlo, const int *lo_ranks, const int *lo_sizes, const int *shr_ltdof,
                             const int *ext_ldof, int64_t nuniq, const int *unq_ltdof, const int *unq_shr_i,
                             const int *unq_shr_j, mfem_amd_exchange **exchange, mfem_amd_comm *comm);
int mfem_amd_exchange_destroy(mfem_amd_exchange *exchange);

#ifdef __cplusplus
}
#endif
#endif /* MFEM_AMD_COMM_H */

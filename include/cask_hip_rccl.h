/*
 * cask_hip_rccl.h -- the collectives of a row-sharded solve, issued by the engine itself.
 *
 * cask_hip_solve_device (include/cask_hip.h) takes the all-reduce of its dot products and, for blocks with global
 * column indices, the all-gather of its operand as callbacks.  A host program may implement them with whatever it
 * has (cask_amd/dist.py: torch.distributed); these are the native ones: RCCL calls on the solver's stream, no
 * interpreter and no second stream on the path (a pass of a sharded solve needs two all-reduces of 8-16 bytes --
 * the host cost of issuing them is the budget).  The reference has no counterpart (single device);
 * the quantities reduced are the cblas_ddot results of pcg (src/runtime/SparseLinearSolvers.hpp:198,208,218).
 *
 * RCCL is opened at run time (librccl.so.1, the copy the process already has if any): libcask_hip.so does not
 * link against it, and a build without RCCL still loads.
 *
 * Bootstrap: rank 0 calls cask_hip_rccl_unique_id and ships the 128 bytes to the other ranks by any transport;
 * every rank then calls cask_hip_rccl_comm_create (collective).
 */
#ifndef CASK_HIP_RCCL_H
#define CASK_HIP_RCCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASK_HIP_RCCL_ID_BYTES 128

typedef struct cask_hip_comm cask_hip_comm;

int cask_hip_rccl_unique_id(unsigned char *id_out /* CASK_HIP_RCCL_ID_BYTES */);

/* bounds: world+1 row boundaries of the partition (bounds[0] = 0; rank g owns [bounds[g], bounds[g+1])), needed by
 * the operand all-gather; NULL if only all-reduces are used.  Uses the current HIP device. */
int cask_hip_rccl_comm_create(const unsigned char *id, int32_t rank, int32_t world, const int64_t *bounds,
                              cask_hip_comm **out);
int cask_hip_rccl_comm_destroy(cask_hip_comm *comm);
/* Padded layout of the gathered vector: rank g's slice at [g*stride, g*stride + its rows), stride >= the longest slice.
 * cask_hip_rccl_allgather is then ONE ncclAllGather of `stride` doubles per rank whatever the row partition --
 * nnz-balanced blocks are always uneven, and the unpadded form is one broadcast per rank -- and d_local must hold
 * `stride` doubles (cask_hip_solve_device lays its vector slots out with cask_hip_solver_config.stride).  The
 * block's column indices are remapped to the layout by the caller at plan time.  0 restores the contiguous layout. */
int cask_hip_rccl_comm_set_stride(cask_hip_comm *comm, int64_t stride);

/* What RCCL itself says about the communicator: ncclCommCount, ncclCommUserRank, ncclCommCuDevice and that device's
 * PCI bus id (so that a multi-GPU run can prove "N ranks on N distinct devices" from its own output; -1 / "" where the
 * library lacks the query).  Any pointer may be NULL. */
int cask_hip_rccl_comm_info(const cask_hip_comm *comm, int32_t *nranks, int32_t *rank, int32_t *device,
                            char *pci_bus_id, int32_t pci_len);

/* cask_hip_allreduce_fn / cask_hip_exchange_fn with user = the communicator: in-place sum of `count` doubles over
 * the ranks; gather of every rank's slice (uneven: one broadcast per rank inside a group) into d_full.  Both are
 * enqueued on `stream` and return without waiting. */
int cask_hip_rccl_allreduce(double *d_values, int32_t count, void *stream, void *comm);
int cask_hip_rccl_allgather(const double *d_local, double *d_full, void *stream, void *comm);

#ifdef __cplusplus
}
#endif
#endif /* CASK_HIP_RCCL_H */

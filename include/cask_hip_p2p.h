/*
 * cask_hip_p2p.h -- multi-GPU exchange step of the sharded SpMV without a collective:
 * peer-to-peer loads over xGMI from vectors shared between the per-GPU processes.
 *
 * The reference has no multi-device code; its only precedent is the per-pipe contiguous row
 * split of Spmv::preprocess (src/runtime/Spmv.cpp:334-364) and the num_controllers memory
 * banks of GeneratedSpmvImplementation (src/runtime/GeneratedImplSupport.hpp:56).  Here one
 * process drives one GPU and owns a contiguous row block plus the matching slice of x.  The
 * slice lives in a SHARED allocation: its owner exports a 64-byte handle, every peer opens
 * it and gets a device pointer it can load from directly (xGMI is point-to-point, a remote
 * load is one hop).  Per product a rank pulls the x entries its block references from the
 * owners' slices (cask_hip_halo_pull_device: one small kernel, no host involvement, legal
 * inside a HIP graph) into the tail of its own extended x, then runs the ordinary local kernel.
 *
 * Plain C, same conventions as cask_hip.h (status codes, cask_hip_last_error()).
 */
#ifndef CASK_HIP_P2P_H
#define CASK_HIP_P2P_H

#include <stdint.h>

#include "cask_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CASK_HIP_SHARED_HANDLE_BYTES 64

/* Allocate `bytes` of zeroed FINE-GRAINED device memory on the current device (coherent at system scope while
 * kernels run: peers store into it and its owner polls flags in it) and export it.
 * handle_out receives CASK_HIP_SHARED_HANDLE_BYTES bytes to pass to the peer processes
 * (any byte transport: torch.distributed object collectives, a file, a socket). */
int cask_hip_shared_alloc(int64_t bytes, void **d_ptr_out, unsigned char *handle_out);
/* Release an allocation made by cask_hip_shared_alloc (after every peer has closed it). */
int cask_hip_shared_free(void *d_ptr);
/* Map a peer's allocation into this process; the pointer is loadable from kernels on the
 * current device.  Never call it on a handle exported by this same process. */
int cask_hip_shared_open(const unsigned char *handle, void **d_ptr_out);
int cask_hip_shared_close(void *d_ptr);

/* Plain copies for callers without a device-array library (and the mapping self-test:
 * reading a few bytes of a freshly opened peer allocation through the runtime fails with a
 * status code where a kernel would fault). */
int cask_hip_copy_to_device(void *d_dst, const void *h_src, int64_t bytes);
int cask_hip_copy_to_host(void *h_dst, const void *d_src, int64_t bytes);

/* dst[j] = *(const double *)src_addr[j]  for j < n_halo.  src_addr is a DEVICE array of
 * absolute device addresses (own or peer allocations, 8-byte aligned), built once per
 * matrix from the opened base pointers.  Asynchronous on `stream`. */
int cask_hip_halo_pull_device(int64_t n_halo, const uint64_t *d_src_addr, double *d_dst, void *stream);

/* Fold the exchange into the product kernel.  After this call the columns >= n_own of `m` are
 * halo columns: column n_own + j is read from *(const double *)d_src_addr[j] by the product
 * kernel itself (the x window of a workgroup at a seam is staged with remote loads; everything
 * else is untouched), so a sharded product is ONE launch: no pull kernel, no launch boundary,
 * no copy of the halo.  The x argument of a product then only needs its first n_own entries.
 * d_src_addr is a DEVICE array of n_cols - n_own absolute addresses (own or peer allocations,
 * 8-byte aligned), borrowed until the handle is destroyed or the call is repeated; NULL
 * returns the handle to the plain layout.  MERGE design points only.  The caller orders
 * accesses across ranks exactly as for cask_hip_halo_pull_device.  Takes the place of the
 * per-controller x upload of Spmv::spmv (src/runtime/Spmv.cpp:144-183): there every pipe gets
 * its own copy of x through dramWrite, here a rank reads the entries it needs where they live. */
int cask_hip_csr_set_halo_sources(cask_hip_matrix *m, int32_t n_own, const uint64_t *d_src_addr);

/* ---- push all-gather: the operand exchange of BASELINE configs[3] without a collective library ---------------
 * Every rank owns TWO gathered vectors (world * stride doubles each; rank g's slice at g*stride: the padded
 * layout of cask_hip_rccl_comm_set_stride) and a flag array (world ints) in shared allocations
 * (cask_hip_shared_alloc) that every peer has opened.  One exchange = one launch on `stream`: this rank's
 * `stride` doubles at d_local are stored into slot [rank] of EVERY rank's gathered vector (16-byte write-through
 * stores over xGMI, own copy included), then a sequence number into slot [rank] of every rank's flag array; the
 * launch ends when all world flags of this exchange have arrived here.  *d_full_out is the local gathered vector
 * the product that follows must read -- the two alternate, so that a peer may already push the next exchange
 * while this rank's product still runs.  Stream order on every rank (exchange k, product k, exchange k+1, ...) is
 * all the cross-rank ordering needed.  Polls are bounded (about two seconds); cask_hip_push_check (synchronous)
 * reports a timeout.  All ranks must issue the same sequence of exchanges.
 *   full_addr   [2 * world] this process's mappings of every rank's first, then second gathered vector
 *               (own entries: the local pointers)
 *   flag_addr   [world] likewise for the flag regions: CASK_HIP_PUSH_FLAG_BYTES each, zero-initialised
 *               (cask_hip_shared_alloc does that), 8-byte aligned
 * Replaces, like the RCCL all-gather, the per-pipe x upload of Spmv::spmv (src/runtime/Spmv.cpp:144-183). */
#define CASK_HIP_PUSH_MAX_WORLD 64
/* vector flags [64 ints], reserved [64 ints], scalar tables [2][64][4] of 16-byte {value, sequence} granules */
#define CASK_HIP_PUSH_FLAG_BYTES (2 * 64 * 4 + 2 * 64 * 4 * 16)
typedef struct cask_hip_push cask_hip_push;
int cask_hip_push_create(int32_t rank, int32_t world, int64_t stride, const uint64_t *full_addr, const uint64_t *flag_addr,
                         cask_hip_push **out);
int cask_hip_push_destroy(cask_hip_push *p);
int cask_hip_push_allgather(cask_hip_push *p, const double *d_local, double **d_full_out, void *stream);
int cask_hip_push_check(cask_hip_push *p);
/* Where this rank's slice sits inside the gathered vector the NEXT exchange fills (slot [rank]; it alternates with the
 * vectors).  A producer that writes its slice there and passes that pointer as d_local saves the own copy: the exchange
 * then only stores to the peers. */
int cask_hip_push_own_slot(cask_hip_push *p, double **d_slot_out);
/* In-place sum of 1..4 doubles over the ranks of `push` (a cask_hip_push *), same transport: one one-wave launch on
 * `stream` stores the values into every peer's scalar table (16-byte {value, sequence number} granules: one trip),
 * waits for all world contributions and adds them in rank order (every rank ends with the same bits).  Has the type of cask_hip_allreduce_fn: pass it with
 * allreduce_user = the cask_hip_push to cask_hip_solve_device and a row-sharded CG / BiCG pass needs no collective
 * library at all (SparseLinearSolvers.hpp:200-232: the dot products of a pass). */
int cask_hip_push_allreduce(double *d_values, int32_t count, void *stream, void *push);

#ifdef __cplusplus
}
#endif
#endif /* CASK_HIP_P2P_H */

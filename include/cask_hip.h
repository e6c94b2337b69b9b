/*
 * cask_hip.h -- thin C ABI of the MI355X (gfx950) SpMV engine that replaces the
 * Maxeler DFE backend of CASK.
 *
 * Plain C: opaque handles, plain pointers and sizes, int status codes.  No C++
 * types, no torch types.  Every entry point names the reference interface it
 * stands in for (paths relative to the reference checkout).  Status 0 is
 * success; on failure cask_hip_last_error() returns a thread-local message.
 *
 * Parameter mapping from the reference's architecture parameters
 * (src/runtime/GeneratedImplSupport.hpp:56, src/frontend/params.json):
 *   input_width     -> lanes_per_row   (rows per wavefront = 64 / lanes_per_row)
 *   cache_size      -> tile_width      (x-vector tile staged in LDS, doubles)
 *   num_pipes       -> workgroups      (derived: grid size; not a parameter)
 *   num_controllers -> GPUs            (one process per GPU; see cask_amd/dist.py)
 *   max_rows        -> no limit        (HBM holds any int32-indexed matrix)
 */
#ifndef CASK_HIP_H
#define CASK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASK_HIP_ABI_VERSION 7   /* 7: variant SLICE (row-mapped slices for short rows + nonzero-mapped blocks for long ones, one launch);
                                  * VECTOR sends rows far longer than its lanes suit to the long-row pieces; SCAN's far_columns = 1 / 2
                                  * removed (they cut fabric traffic and cost more time than they saved), CASK_HIP_PRECOND_ILU0_MC removed (behind Jacobi): rejected like the rest;
                                  * cask_hip_spmv stages host vectors through pinned / registered memory (cask_hip_host_entry_*);
                                  * 6: measured losers removed -- variant MERGE_PAIR (5, xcd_remap = 2), index16 = 3 / 4 (run records),
                                  * far_columns = 1 / 2 for MERGE: each is rejected with CASK_HIP_ERR_INVALID and a message naming
                                  * the replacement; 5: cask_hip_spmv_windows_device; 4: variant SCAN, CASK_HIP_PRECOND_ILU0_MC,
                                  * solver stride with an exchange callback */

/* status codes */
#define CASK_HIP_OK               0
#define CASK_HIP_ERR_INVALID      1   /* bad argument (maps to std::invalid_argument) */
#define CASK_HIP_ERR_RUNTIME      2   /* HIP runtime failure (maps to std::runtime_error) */
#define CASK_HIP_ERR_NO_DEVICE    3   /* no gfx950 device visible */
#define CASK_HIP_ERR_ALLOC        4

/* kernel variants */
#define CASK_HIP_VARIANT_AUTO     0   /* pick from row-length statistics */
#define CASK_HIP_VARIANT_VECTOR   1   /* lanes_per_row lanes of a wavefront per row (1 = thread per row) */
#define CASK_HIP_VARIANT_MERGE    2   /* merge-based: equal (rows+nnz) items per workgroup, products and x tile in LDS */
#define CASK_HIP_VARIANT_MERGE_WAVE 3 /* merge-based, persistent software-pipelined waves (no workgroup barrier, x from L2) */
#define CASK_HIP_VARIANT_MERGE_PAIR_REMOVED 5 /* ABI 4-5: MERGE with two blocks per workgroup.  Removed in ABI 6 (it won on
                                       * one BASELINE family, by 2 %); the value is rejected, never reused */
#define CASK_HIP_VARIANT_SCAN     4   /* nonzero-mapped: equal nonzeros per workgroup, thread-owned runs of products and a
                                       * segmented scan of the carries; no row_ptr stream.  tile_width = x window staged in
                                       * LDS (per block: the densest column range of that width) */
#define CASK_HIP_VARIANT_SLICE    6   /* short rows row-mapped, long rows nonzero-mapped, ONE launch (ABI 7): rows of at most
                                       * K = lanes_per_row nonzeros (1..8; 0 = 4) are sorted by length inside windows of
                                       * consecutive rows and stored as jagged planes -- a thread per row, sums in registers,
                                       * y coalesced through a 16-bit slot map; longer rows are SCAN blocks over a compacted
                                       * copy (wg_size, items_per_thread 4 / 8, tile_width as for SCAN).  The reference packs
                                       * several short rows into one cycle under a lane mask:
                                       * src/spmv/src/ParallelCsrReadControl.java:119-145,262-276 */

typedef struct cask_hip_matrix cask_hip_matrix;   /* device-resident CSR + launch plan */

/* Architecture parameters of one SpMV "design point"; 0 in any field = default
 * (tile_width, xcd_remap, nontemporal, index16: 0 = default, -1 = off).
 * The DSE (cask_hip_tune) sweeps these per matrix exactly as the reference's
 * DSE sweeps num_pipes/input_width/cache_size (src/runtime/Dse.cpp:103-109). */
typedef struct cask_hip_params {
  int32_t variant;          /* CASK_HIP_VARIANT_*                                   */
  int32_t lanes_per_row;    /* VECTOR: 1,2,4,...,64; SLICE: K, the longest row (nonzeros) a slice thread takes: 1..8 */
  int32_t tile_width;       /* doubles of x staged in LDS per workgroup; -1 = no tile */
  int32_t wg_size;          /* threads per workgroup: 64,128,256,512,1024            */
  int32_t items_per_thread; /* MERGE / MERGE_WAVE: merge items per lane: 2,4,8,16    */
  int32_t xcd_remap;        /* 1 = contiguous row blocks per XCD (8 XCDs), -1 = off */
  int32_t nontemporal;      /* 1 = stream values/col_ind with nontemporal loads, -1 = off           */
  int32_t index16;          /* MERGE with an x tile: 1 = stream tile-relative slot indices instead of 32-bit columns
                             * (12 bits each, packed per thread, where the kernel has that layout; else 16 bits),
                             * 2 = 16-bit slots only, -1 = off (3 / 4, ABI 5's run records, are rejected).
                             * cask_hip_csr_get_params reports what the plan streams: 1 = 12-bit, 2 = 16-bit.
                             * The reference's own stream compaction is the RLE of empty-row runs,
                             * src/runtime/Spmv.hpp:213-250 */
  int32_t far_columns;      /* retired (ABI 7): 0 / -1 only.  ABI 4-6 served the scattered nonzeros of a SCAN plan through a
                             * column-panel pre-gather (1: its own launch, 2: producer workgroups of the product launch) -- the
                             * reference's column blocking (SparseMatrix.hpp:459-482) applied to the far part only; it cut the
                             * fabric traffic of the webbase-like matrix from 2.3x to 1.28x the algorithmic bytes and cost more
                             * time than it saved (docs/experiments.md).  1 / 2 are rejected with a reason */
} cask_hip_params;

typedef struct cask_hip_csr_info {
  int32_t n_rows, n_cols;
  int64_t nnz;
  int32_t grid;             /* workgroups per launch                  */
  int32_t lds_bytes;        /* dynamic LDS per workgroup              */
  int32_t n_long_rows;      /* rows handled by the long-row path      */
  int32_t n_split_rows;     /* rows split over several workgroups     */
  int32_t max_row_nnz;
  int32_t empty_rows;
  double  mean_row_nnz;
  int64_t algorithmic_bytes;/* 12*nnz + 4*(n_rows+1) + 8*n_cols + 8*n_rows */
  int32_t fuses_dot;        /* 1: the design point has the fused dot epilogue (composed solver passes possible) */
  int32_t reserved;
} cask_hip_csr_info;

typedef struct cask_hip_device_props {
  char    name[64];
  char    arch[32];
  int32_t compute_units;
  int32_t lds_bytes_per_cu;
  int32_t wavefront_size;
  int32_t clock_mhz;
  int64_t hbm_bytes;
  int32_t l2_bytes;
  int32_t reserved;
} cask_hip_device_props;

/* one measured design point (the measured counterpart of the reference's
 * estimated_gflops entries in dse_out.json, src/main.cpp:81-117) */
typedef struct cask_hip_tune_point {
  cask_hip_params params;
  double  usec;             /* the score: microseconds per SpMV, COLD (launches rotate over device copies of the
                             * matrix that together exceed 2x the 256 MiB Infinity Cache, so every launch reads
                             * HBM); matrices under 4 MB are timed warm only                                   */
  double  gflops;           /* 2*nnz / usec                            */
  double  gbytes_per_s;     /* algorithmic bytes / usec                */
  int32_t valid;            /* 0 = rejected (e.g. LDS over budget)     */
  int32_t copies;           /* device copies the cold timing rotated over (1 = warm only) */
  double  usec_warm;        /* one copy replayed back to back (what a solver iteration sees) */
} cask_hip_tune_point;

const char *cask_hip_last_error(void);
int cask_hip_abi_version(void);

/* Devices.  Replaces cask::model::DeviceModel's constants
 * (src/runtime/Model.hpp:93-136) with what the HIP runtime reports. */
int cask_hip_device_count(int32_t *count);
int cask_hip_device_props_get(int32_t device, cask_hip_device_props *out);
/* PCI bus id of a device ("0000:c1:00.0"): what tells two ranks apart that both call their GPU "device 0" -- a multi-GPU
 * run lists one per rank so that its output proves N ranks sat on N distinct devices.  (The reference has one device per
 * process: max_open / device name in src/runtime/Spmv.cpp:236-262.) */
int cask_hip_device_pci_bus_id(int32_t device, char *out, int32_t len);

/* Upload a 0-based CSR matrix (host pointers, borrowed for the call) to the
 * current HIP device and build the launch plan.  Replaces
 * Spmv::preprocess (src/runtime/Spmv.cpp:329-365) together with the matrix
 * half of writeDataForPartition / Spmv_<id>_dramWrite
 * (Spmv.cpp:144-183, GeneratedImplSupport.hpp:44-49): the matrix is uploaded
 * ONCE here, not on every spmv() call.  params may be NULL (all defaults). */
int cask_hip_csr_create(int32_t n_rows, int32_t n_cols, int64_t nnz,
                        const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                        const cask_hip_params *params, cask_hip_matrix **out);

/* Same, but the three arrays already live in device memory and stay owned by
 * the caller: they must outlive the handle and must not be modified while it
 * exists (the launch plan caches what it derived from them; re-create the
 * handle after a change).  Columns are range-checked here. */
int cask_hip_csr_create_device(int32_t n_rows, int32_t n_cols, int64_t nnz,
                               const int32_t *d_row_ptr, const int32_t *d_col_ind,
                               const double *d_values,
                               const cask_hip_params *params, cask_hip_matrix **out);

int cask_hip_csr_destroy(cask_hip_matrix *m);

/* Change the design point (re-plans; the matrix is not re-uploaded). */
int cask_hip_csr_set_params(cask_hip_matrix *m, const cask_hip_params *params);
/* The resolved design point (AUTO and defaults filled in). */
int cask_hip_csr_get_params(const cask_hip_matrix *m, cask_hip_params *out);
int cask_hip_csr_get_info(const cask_hip_matrix *m, cask_hip_csr_info *out);

/* y = A x with host vectors: x up, one launch, y down, synchronously.
 * Replaces Spmv::spmv (src/runtime/Spmv.cpp:185-328): the x half of
 * dramWrite, the run function Spmv_<id>(...) and dramRead
 * (GeneratedImplSupport.hpp:31-49; the reference re-uploads the MATRIX too, every call). x has n_cols entries, y n_rows.
 * How the vectors travel (ABI 7; CASK_HIP_HOST_ENTRY_* below, or the environment variable CASK_HIP_HOST_ENTRY =
 * auto | pageable | staged | register_cache):
 *   staged (the default for 64 KiB .. 4 MiB of vectors; smaller and larger ones go the pageable way, which is as fast
 *     there): x is copied into the handle's pinned staging buffer by a few host threads
 *     (one core moves 500 KB in 25-35 us: most of what a call used to cost), pulled over PCIe by a copy kernel in front of
 *     the product, and the product writes y straight into pinned host memory; a threaded copy out.  The caller learns that y
 *     is there from a word a one-thread launch stores behind the product (polled; 4.5 us earlier than the queue's signal;
 *     bounded -- after 20 ms the call synchronises, which is also what reports a fault).
 *     Safe for any pointer: the caller's memory is only ever touched by the CPU.
 *   registered: a vector inside a range given to cask_hip_host_register is read / written by the GPU in place (no host
 *     copy at all).  That is a CONTRACT, not a cache: the range must stay mapped until cask_hip_host_unregister -- a GPU
 *     access to a registered range whose pages were returned to the OS is a fatal fault, and nothing can detect a free()
 *     behind the engine's back.  CASK_HIP_HOST_ENTRY=register_cache applies it to every vector a call sees (16 ranges,
 *     least recently used out) for processes that can promise that -- e.g. with glibc's mmap threshold raised so that
 *     free() never unmaps.
 *   pageable: hipMemcpyAsync from / to the caller's pageable memory (ABI <= 6). */
int cask_hip_spmv(cask_hip_matrix *m, const double *x, double *y);
#define CASK_HIP_HOST_ENTRY_AUTO      0
#define CASK_HIP_HOST_ENTRY_PAGEABLE  1
#define CASK_HIP_HOST_ENTRY_STAGED    2
#define CASK_HIP_HOST_ENTRY_REGISTER_CACHE 3
/* Process-wide override of the choice above (tools, tests); returns the previous mode. */
int cask_hip_host_entry_mode(int mode);
/* Declare [ptr, ptr + bytes) long-lived host memory the GPU may access in place (hipHostRegister): cask_hip_spmv then
 * moves a vector that lies inside it without any host copy.  Ranges are reference-counted by their start address. */
int cask_hip_host_register(const void *ptr, size_t bytes);
int cask_hip_host_unregister(const void *ptr);

/* y = A x with device vectors on `stream` (a hipStream_t; NULL = default
 * stream), asynchronously.  d_x must be 16-byte aligned (any hipMalloc'd vector is).  This is the device run function alone, the
 * counterpart of the timed region in Spmv.cpp:270-285. */
int cask_hip_spmv_device(cask_hip_matrix *m, const double *d_x, double *d_y, void *stream);

/* k products in stream order from ONE host call: product i is y = A_(i mod n_mats) x.  All handles have the
 * shape of mats[0].  What a caller that issues many products back to back (a benchmark rotating over device
 * copies of a matrix, a solver's inner loop) uses to keep the per-launch host cost at the runtime's launch cost
 * instead of a language binding's call cost. */
int cask_hip_spmv_sequence_device(cask_hip_matrix *const *mats, int32_t n_mats, const double *d_x, double *d_y,
                                  int32_t k, void *stream);

/* `windows` timed windows of k products each, back to back on `stream`, from ONE host call: windows + 1 timing
 * events (created with hipEventDisableSystemFence: a timing event needs no system-scope fence, and that fence
 * costs ~4 us per record between dependent launches), event r -> event r + 1 brackets exactly the k launches of
 * window r; product i of the whole call is y = A_(i mod n_mats) x.  Returns after the last event has completed
 * with usec[r] = the device time of window r in microseconds.  as_graph = 1: the windows x k launches are the
 * kernel nodes of ONE graph with an event-record node at every window boundary (captured once, launched twice: the
 * first launch is the lead-in, the events keep the second's times) -- graph nodes are the cheapest launches the
 * runtime has; fails with CASK_HIP_ERR_RUNTIME where the runtime cannot record an external event inside a capture
 * (the caller falls back to as_graph = 0: stream launches).  Replaces the reference's timed loop
 * (src/runtime/Spmv.cpp:265-301: impl.Spmv x nIterations between two clock reads, divided by nIterations) with a
 * distribution instead of one sample. */
int cask_hip_spmv_windows_device(cask_hip_matrix *const *mats, int32_t n_mats, const double *d_x, double *d_y,
                                 int32_t k, int32_t windows, int32_t as_graph, double *usec, void *stream);

/* y = A x and *d_result = w . y in one pass (device vectors, asynchronous on `stream`): with a
 * MERGE design point every workgroup leaves its rows' share of the dot behind and a one-workgroup
 * kernel adds the shares in block order (reproducible); other design points run the product and
 * the dot separately.  This is the "Ap = A p; alpha = rsold / (p.Ap)" pair of the reference's CG
 * (mkl_dcsrsymv + cblas_ddot, src/runtime/SparseLinearSolvers.hpp:206-208) without the second
 * pass over p and Ap; cask_hip_cg / cask_hip_bicg use the same epilogue internally. */
int cask_hip_spmv_dot_device(cask_hip_matrix *m, const double *d_x, double *d_y, const double *d_w,
                             double *d_result, void *stream);

/* y = A^T x (device vectors; x has n_rows entries, y n_cols).  Built on first
 * use as a second CSR (explicit transpose, no atomics).  No reference
 * counterpart (DfeBiCgSolver is declared only, SparseLinearSolvers.hpp:56-61). */
int cask_hip_spmv_transpose_device(cask_hip_matrix *m, const double *d_x, double *d_y, void *stream);

/* Time `iters` back-to-back launches (after `warmup`) with HIP events on the
 * handle's own stream; median microseconds per launch. */
int cask_hip_spmv_time(cask_hip_matrix *m, const double *d_x, double *d_y,
                       int32_t warmup, int32_t iters, double *usec_median, double *usec_min);

/* Measured design-space exploration: evaluates every point of the cross
 * product variants x lanes x tiles x wg_sizes x items in the reference's
 * sweep order (first list fastest; src/runtime/Utils.hpp:158-202), leaves the
 * fastest valid point active on the handle and returns all measurements.
 * Replaces dse_run/better (src/runtime/Dse.cpp:12-74), measured not modelled.
 * Every point is timed as a HIP graph of back-to-back launches, cold (ranked on
 * this) and warm; `iters` = launches per graph (0 = default), `warmup` = untimed
 * replays.  The one DSE of the engine: build/main, tools/dse.py, bench.py and
 * cask_amd.dse all rank design points with this call. */
int cask_hip_tune(cask_hip_matrix *m,
                  const int32_t *variants, int32_t n_variants,
                  const int32_t *lanes, int32_t n_lanes,
                  const int32_t *tiles, int32_t n_tiles,
                  const int32_t *wg_sizes, int32_t n_wg_sizes,
                  const int32_t *items, int32_t n_items,
                  int32_t warmup, int32_t iters,
                  cask_hip_tune_point *results, int32_t max_results, int32_t *n_results,
                  int32_t *best_index);

/* BLAS-1 on device vectors, the CG building blocks the reference takes from
 * MKL (cblas_ddot / cblas_daxpy / cblas_daxpby,
 * src/runtime/SparseLinearSolvers.hpp:190-229).  Scalars live in device
 * memory so a solver iteration never synchronises with the host:
 *   ddot  : *d_result = sum x_i*y_i   (deterministic two-stage reduction)
 *   daxpy : y += sign * (*d_num / *d_den) * x        (d_den may be NULL => 1)
 *   daxpby: y = alpha_h * x + sign * (*d_num / *d_den) * y  (d_num NULL => beta_h)
 */
int cask_hip_ddot_device(int64_t n, const double *d_x, const double *d_y, double *d_result, void *stream);
int cask_hip_daxpy_device(int64_t n, double sign, const double *d_num, const double *d_den,
                          const double *d_x, double *d_y, void *stream);
int cask_hip_daxpby_device(int64_t n, double alpha, const double *d_x,
                           double sign, double beta, const double *d_num, const double *d_den,
                           double *d_y, void *stream);

/* Un-preconditioned CG on a full symmetric CSR handle, the recurrence of
 * pcg<double, IdentityPreconditioner> (SparseLinearSolvers.hpp:162-239):
 * absolute test r.r <= tol^2, `iterations` written at the end of each
 * non-converged pass.  Host vectors; x holds the initial guess. */
int cask_hip_cg(cask_hip_matrix *m, const double *rhs, double *x, int32_t maxiters, double tol,
                int32_t *iterations, int32_t *converged, double *usec_per_iteration);
/* Classical BiCG with A and A^T (BASELINE config 5). */
int cask_hip_bicg(cask_hip_matrix *m, const double *rhs, double *x, int32_t maxiters, double tol,
                  int32_t *iterations, int32_t *converged, double *usec_per_iteration);

/* ---- CG / BiCG on device vectors; one rank of a row-sharded solve -------------------------------------
 * The same recurrences as cask_hip_cg / cask_hip_bicg (pcg<double, IdentityPreconditioner>,
 * SparseLinearSolvers.hpp:162-239; BiCG: the declared-only DfeBiCgSolver, :56-61) with device vectors on
 * `stream`, plus what a rank of a row-sharded solve (BASELINE configs 3 and 5 on several GPUs) needs:
 *   allreduce  sums `count` doubles at d_values over all ranks, in place, ordered on `stream` (RCCL
 *              all-reduce of the dot products).  NULL = single rank.
 *   exchange   classic mode only: gathers the operand slices of all ranks into d_full (RCCL all-gather
 *              of x; n_full entries), for blocks whose column indices address the gathered vector.  With
 *              d_shared_base == NULL and stride > 0 the solver's private vector slots are `stride` doubles
 *              long, so that the callback may read `stride` doubles from d_local: the padded-stride layout
 *              (rank g's slice at g*stride; cask_hip_rccl_comm_set_stride) makes the gather ONE collective
 *              whatever the row partition.  NULL = the block reads its halo itself
 *              (cask_hip_csr_set_halo_sources) or has none.
 *   d_shared_base / stride   sharded, in-kernel halo: the vectors peers read (r, p, rt, pt) live in this
 *              rank's shared allocation (cask_hip_shared_alloc), slot k at base + k*stride doubles; 3
 *              slots for CG, 6 for BiCG; the halo tables of A and A^T point at slot 0 of the peers'
 *              allocations (same stride on every rank).  NULL = private vectors.
 * mode: composed = the product launch composes p = r + beta*p on the fly (2 launches per CG pass, 3 per
 * BiCG pass; needs MERGE plans), classic = product, x/r update and p update are separate launches.
 * d_x holds the initial guess and receives the solution (n_rows entries of this rank); d_x and d_rhs are
 * 16-byte aligned.  Every rank takes the same decisions from the same all-reduced scalars.  A handle carries ONE
 * solver workspace (and, per launch, the halo offset of the operand): one solve or product per handle at a time --
 * distinct handles may be used from distinct threads / streams. */
#define CASK_HIP_SOLVER_CG        1
#define CASK_HIP_SOLVER_BICG      2
#define CASK_HIP_SOLVER_AUTO      0
#define CASK_HIP_SOLVER_COMPOSED  1
#define CASK_HIP_SOLVER_CLASSIC   2
typedef int (*cask_hip_allreduce_fn)(double *d_values, int32_t count, void *stream, void *user);
typedef int (*cask_hip_exchange_fn)(const double *d_local, double *d_full, void *stream, void *user);
typedef struct cask_hip_solver_config {
  int32_t kind;                  /* CASK_HIP_SOLVER_CG (0 = CG) / _BICG */
  int32_t mode;                  /* CASK_HIP_SOLVER_AUTO / _COMPOSED / _CLASSIC */
  double *d_shared_base;
  int64_t stride;
  cask_hip_allreduce_fn allreduce;
  void *allreduce_user;
  cask_hip_exchange_fn exchange;
  void *exchange_user;
  int64_t n_full;                /* exchange: entries of the gathered operand */
} cask_hip_solver_config;
/* mt: the row block of A^T for a sharded BiCG; NULL = built from m (single rank). */
int cask_hip_solve_device(cask_hip_matrix *m, cask_hip_matrix *mt, const cask_hip_solver_config *cfg,
                          const double *d_rhs, double *d_x, int32_t maxiters, double tol,
                          int32_t *iterations, int32_t *converged, double *usec_per_iteration, void *stream);

/* ---- preconditioning -------------------------------------------------------------
 * The reference's preconditioned CG, pcg<T, Precon> (src/runtime/SparseLinearSolvers.hpp:162-239),
 * knows IdentityPreconditioner (:64-74) and ILUPreconditioner (:77-156): an ILU(0) sweep on the
 * pattern of the matrix, then z = U^-1 (L^-1 r) with two mkl_dcsrtrsv calls on the factors
 * extracted WITH the diagonal (src/runtime/MklLayer.hpp:29-85).  Here the factorisation runs once on
 * the host (it is sequential in the reference too) and every application on the device:
 * level-scheduled triangular solves.  JACOBI (z = r / diag) has no reference counterpart; it is the
 * preconditioner that costs one streaming pass.  ILU0 reproduces the reference bit for bit,
 * including that its "L" solve divides by U's diagonal (L is extracted with the diagonal of the
 * factored matrix): M is then not A's incomplete factorisation and PCG stagnates on most systems --
 * the reference's own test pins the iterate after 2000 stagnating passes
 * (test/LinearSolvers.cpp:54-77).  ILU0_UNIT applies the same factors the textbook way (unit lower
 * diagonal) and is the one to use for solving.  A preconditioner handle is built from ANY square
 * CSR matrix with strictly ascending columns per row (the reference hands pcg the lower triangle of
 * a symmetric matrix and factors exactly that). */
typedef struct cask_hip_precond cask_hip_precond;
#define CASK_HIP_PRECOND_JACOBI    1
#define CASK_HIP_PRECOND_ILU0      2   /* the reference's: both triangular solves divide by the stored diagonal */
#define CASK_HIP_PRECOND_ILU0_UNIT 3   /* textbook ILU(0): same factors, L applied with a unit diagonal          */
#define CASK_HIP_PRECOND_ILU0_MC_REMOVED 4 /* ABI 4-6: ILU(0) of the matrix permuted colour by colour (greedy multicolouring), 2 x colours
                                        * wide launches per application -- not the reference's factors, and behind Jacobi end to end
                                        * on the system it was built for.  Removed in ABI 7; the value is rejected, never reused */
int cask_hip_precond_create(int32_t kind, int32_t n, int64_t nnz, const int32_t *row_ptr,
                            const int32_t *col_ind, const double *values, cask_hip_precond **out);
int cask_hip_precond_destroy(cask_hip_precond *p);
/* ILU0: the factored values in the pattern of the input (ILUPreconditioner::pc): strictly-lower
 * entries hold the multipliers, the rest the upper factor. */
int cask_hip_precond_factor_values(const cask_hip_precond *p, double *values_out);
/* dependency levels of the two triangular solves and kernel launches per application */
int cask_hip_precond_info(const cask_hip_precond *p, int32_t *levels_lower, int32_t *levels_upper,
                          int32_t *launches_per_apply);
/* z = M^-1 r: host vectors (ILUPreconditioner::apply, :143-151) / device vectors on `stream` */
int cask_hip_precond_apply(cask_hip_precond *p, const double *r, double *z);
int cask_hip_precond_apply_device(cask_hip_precond *p, const double *d_r, double *d_z, void *stream);
/* Solve T x = rhs with the lower (lower != 0) or upper triangle of a CSR matrix, diagonal taken
 * from the matrix: cask::mkl::unittrsolve (src/runtime/MklLayer.hpp:29-85, mkl_dcsrtrsv with
 * diag = 'N').  Host vectors; level-scheduled on the device. */
int cask_hip_trsolve(int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_ind,
                     const double *values, int32_t lower, const double *rhs, double *x);
/* Preconditioned CG on a full symmetric CSR handle; precond == NULL is cask_hip_cg.  The test is
 * r.z <= tol^2 like the reference's. */
int cask_hip_pcg(cask_hip_matrix *m, cask_hip_precond *precond, const double *rhs, double *x,
                 int32_t maxiters, double tol, int32_t *iterations, int32_t *converged,
                 double *usec_per_iteration);

#ifdef __cplusplus
}
#endif
#endif /* CASK_HIP_H */

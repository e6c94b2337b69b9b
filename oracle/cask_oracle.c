/*
 * cask_oracle.c -- CPU restatement of the reference's SpMV / CG / ILU / triangular-solve arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cask_amd/ or include/ may link,
 * import or call this file; it is the checker for tests/, for
 * __graft_entry__.smoke() and for the cpu_baseline leg of bench.py.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * arithmetic it restates.  The reference itself cannot be compiled in this
 * image (SparseMatrix.hpp:16 needs <Eigen/Sparse>, Utils.hpp:9 needs Boost,
 * Spmv.cpp:5-6 need the un-vendored dfe-snippets; none are installed), so
 * there is no oracle/_ref build.  Pinning is done against the known answers of
 * the reference's own gtest suites and against MKL (the reference's named CPU
 * path) -- see tests/test_oracle.py and tests/golden/make_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared  (no FMA contraction: the
 * reference builds with plain g++ and un-fused multiply-then-add).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* y = A*x for a 0-based CSR matrix, one row at a time, products added in
 * stored (ascending-column) order starting from 0.0.
 * Restates CsrMatrix::dot -> DokMatrix::dot (src/runtime/SparseMatrix.hpp:422-424,
 * 255-264: "result[row] += b[e.first] * e.second" over a std::map ordered by
 * column) and is the same order as Eigen's row-major product used as the
 * golden in test/test_spmv.cpp:45-47. */
void oracle_csr_spmv(int32_t n_rows, const int32_t *row_ptr, const int32_t *col_ind,
                     const double *values, const double *x, double *y)
{
    for (int32_t r = 0; r < n_rows; r++) {
        double acc = 0.0;
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++)
            acc += x[col_ind[k]] * values[k];
        y[r] = acc;
    }
}

/* y = A^T * x (A is n_rows x n_cols, y has n_cols entries).  The reference has
 * no transpose product; BiCG (BASELINE config 5) needs one.  Defined as the
 * row-ordered product of the explicitly transposed matrix, i.e. for output
 * column c the terms arrive in ascending row order. */
void oracle_csr_spmv_t(int32_t n_rows, int32_t n_cols, const int32_t *row_ptr,
                       const int32_t *col_ind, const double *values, const double *x, double *y)
{
    for (int32_t c = 0; c < n_cols; c++) y[c] = 0.0;
    for (int32_t r = 0; r < n_rows; r++)
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++)
            y[col_ind[k]] += x[r] * values[k];
}

/* y = A*x where only the lower triangle of symmetric A is stored (CSR, 0-based).
 * Restates SymCsrMatrix::dot (SparseMatrix.hpp:512-514: expand with
 * explicitSymmetric(), :156-189, then DokMatrix::dot) -- i.e. row r sums its
 * mirrored entries in ascending column order.  This is also the operator that
 * mkl_dcsrsymv('l') applies in pcg (SparseLinearSolvers.hpp:189,206).
 * Work arrays are allocated here; returns 0 on success. */
int oracle_symcsr_spmv(int32_t n, const int32_t *row_ptr, const int32_t *col_ind,
                       const double *values, const double *x, double *y)
{
    /* count entries per row of the expanded matrix */
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    if (!cnt) return 1;
    for (int32_t r = 0; r < n; r++)
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            int32_t c = col_ind[k];
            cnt[r + 1]++;
            if (c != r) cnt[c + 1]++;
        }
    for (int32_t r = 0; r < n; r++) cnt[r + 1] += cnt[r];
    int64_t total = cnt[n];
    int32_t *ecol = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    double *eval = (double *)malloc((size_t)(total ? total : 1) * sizeof(double));
    int64_t *fill = (int64_t *)malloc(((size_t)n + 1) * sizeof(int64_t));
    if (!ecol || !eval || !fill) { free(cnt); free(ecol); free(eval); free(fill); return 1; }
    memcpy(fill, cnt, ((size_t)n + 1) * sizeof(int64_t));
    /* Row r of the expansion = its stored lower entries (cols <= r, ascending)
     * followed by mirrored entries from later rows (cols > r, ascending because
     * rows are visited in ascending order). */
    for (int32_t r = 0; r < n; r++)
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            int32_t c = col_ind[k];
            ecol[fill[r]] = c; eval[fill[r]] = values[k]; fill[r]++;
        }
    for (int32_t r = 0; r < n; r++)
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            int32_t c = col_ind[k];
            if (c != r) { ecol[fill[c]] = r; eval[fill[c]] = values[k]; fill[c]++; }
        }
    for (int32_t r = 0; r < n; r++) {
        double acc = 0.0;
        for (int64_t k = cnt[r]; k < cnt[r + 1]; k++) acc += x[ecol[k]] * eval[k];
        y[r] = acc;
    }
    free(cnt); free(ecol); free(eval); free(fill);
    return 0;
}

/* The comparison the reference's integration test applies:
 * dfesnippets::numeric_utils::almost_equal(got, exp, 1E-8, 1E-11)
 * (test/test_utils.hpp:36).  dfe-snippets is an un-vendored submodule
 * (.gitmodules:1-3), so its exact form is not under /root/reference; restated
 * conservatively as: equal, or within abs_tol, or within rel_tol of the larger
 * magnitude.  Returns 1 when "almost equal". */
int oracle_almost_equal(double got, double expected, double rel_tol, double abs_tol)
{
    if (got == expected) return 1;
    double diff = fabs(got - expected);
    if (diff <= abs_tol) return 1;
    double mag = fmax(fabs(got), fabs(expected));
    return diff <= rel_tol * mag;
}

/* Number of positions where got/expected are not almost_equal
 * (cask::test::check, test/test_utils.hpp:29-41); first_bad gets the first
 * mismatching index or -1. */
int64_t oracle_count_mismatches(int64_t n, const double *got, const double *expected,
                                double rel_tol, double abs_tol, int64_t *first_bad)
{
    int64_t bad = 0;
    if (first_bad) *first_bad = -1;
    for (int64_t i = 0; i < n; i++)
        if (!oracle_almost_equal(got[i], expected[i], rel_tol, abs_tol)) {
            if (bad == 0 && first_bad) *first_bad = i;
            bad++;
        }
    return bad;
}

/* ---- BLAS-1 as the reference's CG uses it (cblas_ddot/daxpy/daxpby,
 * SparseLinearSolvers.hpp:190-229), sequential order. ---- */
double oracle_ddot(int64_t n, const double *a, const double *b)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}
void oracle_daxpy(int64_t n, double alpha, const double *x, double *y)
{
    for (int64_t i = 0; i < n; i++) y[i] = y[i] + alpha * x[i];
}
/* y = alpha*x + beta*y */
void oracle_daxpby(int64_t n, double alpha, const double *x, double beta, double *y)
{
    for (int64_t i = 0; i < n; i++) y[i] = alpha * x[i] + beta * y[i];
}

/* Un-preconditioned CG exactly as pcg<double, IdentityPreconditioner>
 * (SparseLinearSolvers.hpp:162-239): A given by its LOWER triangle (the
 * SymCsrMatrix::matrix the reference passes), absolute test rsnew <= tol^2,
 * tol = 1e-5 and maxiters = 2000 in the reference (:166-167), `iterations`
 * is only written at the END of a non-converged pass (:231) so a solve that
 * converges in pass i reports i-1 (0 if it converges in pass 0 or 1).
 * x holds the initial guess on entry.  Returns 1 if converged, 0 if not,
 * -1 on allocation failure. */
int oracle_pcg_identity(int32_t n, const int32_t *row_ptr, const int32_t *col_ind,
                        const double *values, const double *rhs, double *x,
                        int32_t maxiters, double tol, int32_t *iterations)
{
    double *r = (double *)malloc((size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *Ap = (double *)malloc((size_t)n * sizeof(double));
    if (!r || !p || !Ap) { free(r); free(p); free(Ap); return -1; }
    int converged = 0;
    /* r = b - A x  (:189-190) */
    oracle_symcsr_spmv(n, row_ptr, col_ind, values, x, r);
    oracle_daxpby(n, 1.0, rhs, -1.0, r);
    /* z = r ; p = z ; rsold = r.z  (:193-198) */
    memcpy(p, r, (size_t)n * sizeof(double));
    double rsold = oracle_ddot(n, r, r);
    for (int32_t i = 0; i < maxiters; i++) {
        oracle_symcsr_spmv(n, row_ptr, col_ind, values, p, Ap);   /* :206 */
        double alpha = rsold / oracle_ddot(n, p, Ap);              /* :208 */
        oracle_daxpy(n, alpha, p, x);                              /* :210 */
        oracle_daxpby(n, -alpha, Ap, 1.0, r);                      /* :212 */
        double rsnew = oracle_ddot(n, r, r);                       /* :215-218 */
        if (rsnew <= tol * tol) { converged = 1; break; }          /* :220-226 */
        oracle_daxpby(n, 1.0, r, rsnew / rsold, p);                /* :229 */
        rsold = rsnew;
        *iterations = i;                                           /* :231 */
    }
    free(r); free(p); free(Ap);
    return converged;
}

/* CG on a FULL (symmetry-expanded) CSR matrix -- same recurrence as above with
 * the general row-ordered product; this is what a GPU CG over io::readMatrix
 * output (IO.hpp:151-163 expands symmetry) computes. */
int oracle_cg_full(int32_t n, const int32_t *row_ptr, const int32_t *col_ind,
                   const double *values, const double *rhs, double *x,
                   int32_t maxiters, double tol, int32_t *iterations)
{
    double *r = (double *)malloc((size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *Ap = (double *)malloc((size_t)n * sizeof(double));
    if (!r || !p || !Ap) { free(r); free(p); free(Ap); return -1; }
    int converged = 0;
    oracle_csr_spmv(n, row_ptr, col_ind, values, x, r);
    oracle_daxpby(n, 1.0, rhs, -1.0, r);
    memcpy(p, r, (size_t)n * sizeof(double));
    double rsold = oracle_ddot(n, r, r);
    for (int32_t i = 0; i < maxiters; i++) {
        oracle_csr_spmv(n, row_ptr, col_ind, values, p, Ap);
        double alpha = rsold / oracle_ddot(n, p, Ap);
        oracle_daxpy(n, alpha, p, x);
        oracle_daxpby(n, -alpha, Ap, 1.0, r);
        double rsnew = oracle_ddot(n, r, r);
        if (rsnew <= tol * tol) { converged = 1; break; }
        oracle_daxpby(n, 1.0, r, rsnew / rsold, p);
        rsold = rsnew;
        *iterations = i;
    }
    free(r); free(p); free(Ap);
    return converged;
}

/* ---- preconditioning: ILUPreconditioner, unittrsolve, pcg<double, ILUPreconditioner> ----------- */

/* position of (r, c) in a CSR row with ascending columns, or -1 (DokMatrix::isNnz, SparseMatrix.hpp) */
static int64_t csr_find(const int32_t *row_ptr, const int32_t *col_ind, int32_t r, int32_t c)
{
    for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++)
        if (col_ind[k] == c) return k;
    return -1;
}

/* In-place ILU(0) of ILUPreconditioner's constructor (SparseLinearSolvers.hpp:88-113), statement by
 * statement: for i = 1..n-1, for every stored (i,k) in ascending k while k < i: skip if (k,k) is not
 * stored; a[i][k] /= a[k][k]; then for every stored (i,j) with j >= k+1: if (k,j) is stored,
 * a[i][j] -= a[k][j] * a[i][k].  Columns must ascend within a row (the reference iterates a std::map). */
void oracle_ilu0(int32_t n, const int32_t *row_ptr, const int32_t *col_ind, double *a)
{
    for (int32_t i = 1; i < n; i++) {
        for (int64_t kk = row_ptr[i]; kk < row_ptr[i + 1]; kk++) {
            const int32_t k = col_ind[kk];
            if (k >= i) break;
            const int64_t dk = csr_find(row_ptr, col_ind, k, k);
            if (dk < 0 || a[dk] == 0.0) continue;          /* !pc.isNnz(k, k): absent OR a stored zero (SparseMatrix.hpp:219-225) */
            a[kk] = a[kk] / a[dk];
            const double beta = a[kk];
            for (int64_t jj = row_ptr[i]; jj < row_ptr[i + 1]; jj++) {
                const int32_t j = col_ind[jj];
                if (j < k + 1) continue;
                const int64_t kj = csr_find(row_ptr, col_ind, k, j);
                if (kj >= 0 && a[kj] != 0.0) a[jj] = a[jj] - a[kj] * beta;   /* pc.isNnz(k, j), :107 */
            }
        }
    }
}

/* x = T^-1 b for the lower (lower != 0) or upper triangle of a CSR matrix with the diagonal taken from
 * the matrix: mkl_dcsrtrsv(uplo, 'N', 'N') as cask::mkl::unittrsolve calls it (MklLayer.hpp:29-85).
 * MKL is a binary; the restatement is the textbook substitution over the stored entries in order. */
static void trsolve_impl(int32_t n, const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                         int lower, int unit, const double *b, double *x)
{
    for (int32_t step = 0; step < n; step++) {
        const int32_t r = lower ? step : n - 1 - step;
        double s = b[r], diag = 0.0;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            const int32_t c = col_ind[k];
            if (c == r) diag = values[k];
            else if (lower ? c < r : c > r) s -= values[k] * x[c];
        }
        x[r] = unit ? s : s / diag;
    }
}
void oracle_trsolve(int32_t n, const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                    int lower, const double *b, double *x)
{
    trsolve_impl(n, row_ptr, col_ind, values, lower, 0, b, x);
}

/* z = U^-1 (L^-1 r) with L / U = lower / upper triangle of the factored matrix INCLUDING its diagonal
 * (ILUPreconditioner::apply :143-151 over getLowerTriangular/getUpperTriangular, SparseMatrix.hpp:227-253). */
void oracle_ilu_apply(int32_t n, const int32_t *row_ptr, const int32_t *col_ind, const double *factored,
                      const double *r, double *tmp, double *z)
{
    oracle_trsolve(n, row_ptr, col_ind, factored, 1, r, tmp);
    oracle_trsolve(n, row_ptr, col_ind, factored, 0, tmp, z);
}
/* the textbook application of the same factors: unit diagonal in L (not what the reference does) */
void oracle_ilu_apply_unit(int32_t n, const int32_t *row_ptr, const int32_t *col_ind, const double *factored,
                           const double *r, double *tmp, double *z)
{
    trsolve_impl(n, row_ptr, col_ind, factored, 1, 1, r, tmp);
    trsolve_impl(n, row_ptr, col_ind, factored, 0, 0, tmp, z);
}

/* pcg<double, ILUPreconditioner> (SparseLinearSolvers.hpp:162-239): the matrix is given by its LOWER
 * triangle, which is also what the preconditioner factors (`Precon precon{a}` :171 receives the same
 * CsrMatrix the symmetric product uses).  jacobi = 1 replaces the ILU by z = r / diag, jacobi = 2 applies
 * the ILU factors with a unit lower diagonal (neither has a reference counterpart; same recurrence).  full != 0: the arrays hold the whole symmetric matrix instead -- product
 * and preconditioner both see all of it (the sensible use; what a GPU solve over io::readMatrix output does).
 * Returns 1 converged / 0 not / -1 out of memory. */
int oracle_pcg_precond(int32_t n, const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                       int jacobi, int full, const double *rhs, double *x, int32_t maxiters, double tol,
                       int32_t *iterations)
{
    const int64_t nnz = row_ptr[n];
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *r = (double *)malloc(nb), *z = (double *)malloc(nb), *p = (double *)malloc(nb);
    double *Ap = (double *)malloc(nb), *tmp = (double *)malloc(nb);
    double *f = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    if (!r || !z || !p || !Ap || !tmp || !f) { free(r); free(z); free(p); free(Ap); free(tmp); free(f); return -1; }
    memcpy(f, values, (size_t)nnz * sizeof(double));
    if (jacobi != 1) oracle_ilu0(n, row_ptr, col_ind, f);
#define ORACLE_PRECOND(src, dst)                                                              \
    do {                                                                                      \
        if (jacobi == 2) {                                                                    \
            oracle_ilu_apply_unit(n, row_ptr, col_ind, f, (src), tmp, (dst));                 \
        } else if (jacobi) {                                                                  \
            for (int32_t q = 0; q < n; q++) {                                                 \
                const int64_t dq = csr_find(row_ptr, col_ind, q, q);                          \
                (dst)[q] = (dq >= 0 && values[dq] != 0.0) ? (src)[q] * (1.0 / values[dq]) : (src)[q]; \
            }                                                                                 \
        } else {                                                                              \
            oracle_ilu_apply(n, row_ptr, col_ind, f, (src), tmp, (dst));                      \
        }                                                                                     \
    } while (0)
    int converged = 0;
#define ORACLE_PRODUCT(src, dst)                                                              \
    do {                                                                                      \
        if (full) oracle_csr_spmv(n, row_ptr, col_ind, values, (src), (dst));                 \
        else      oracle_symcsr_spmv(n, row_ptr, col_ind, values, (src), (dst));              \
    } while (0)
    ORACLE_PRODUCT(x, r);                                          /* :189-190 */
    oracle_daxpby(n, 1.0, rhs, -1.0, r);
    ORACLE_PRECOND(r, z);                                          /* :193 */
    memcpy(p, z, (size_t)n * sizeof(double));                      /* :195 */
    double rsold = oracle_ddot(n, r, z);                           /* :198 */
    for (int32_t i = 0; i < maxiters; i++) {
        ORACLE_PRODUCT(p, Ap);                                     /* :206 */
        double alpha = rsold / oracle_ddot(n, p, Ap);              /* :208 */
        oracle_daxpy(n, alpha, p, x);                              /* :210 */
        oracle_daxpby(n, -alpha, Ap, 1.0, r);                      /* :212 */
        ORACLE_PRECOND(r, z);                                      /* :215 */
        double rsnew = oracle_ddot(n, r, z);                       /* :218 */
        if (rsnew <= tol * tol) { converged = 1; break; }          /* :220-226 */
        oracle_daxpby(n, 1.0, z, rsnew / rsold, p);                /* :229 */
        rsold = rsnew;
        *iterations = i;                                           /* :231 */
    }
#undef ORACLE_PRECOND
#undef ORACLE_PRODUCT
    free(r); free(z); free(p); free(Ap); free(tmp); free(f);
    return converged;
}

/* Classical (Fletcher) BiCG on a general CSR matrix with A and A^T products.
 * PARITY UNPINNED: the reference declares DfeBiCgSolver::solve but never
 * defines it (SparseLinearSolvers.hpp:56-61) and its only working solver is
 * Eigen's BiCGSTAB (SparseLinearSolvers.cpp:22-31); the recurrence below is
 * the textbook one, with the same absolute stopping rule and `iterations`
 * convention as pcg (:220-231) so the two solvers read alike. */
int oracle_bicg(int32_t n, const int32_t *row_ptr, const int32_t *col_ind,
                const double *values, const double *rhs, double *x,
                int32_t maxiters, double tol, int32_t *iterations)
{
    size_t nb = (size_t)n * sizeof(double);
    double *r = (double *)malloc(nb), *rt = (double *)malloc(nb);
    double *p = (double *)malloc(nb), *pt = (double *)malloc(nb);
    double *q = (double *)malloc(nb), *qt = (double *)malloc(nb);
    int converged = 0;
    if (!r || !rt || !p || !pt || !q || !qt) { converged = -1; goto done; }
    oracle_csr_spmv(n, row_ptr, col_ind, values, x, r);
    oracle_daxpby(n, 1.0, rhs, -1.0, r);
    memcpy(rt, r, nb); memcpy(p, r, nb); memcpy(pt, r, nb);
    double rho = oracle_ddot(n, rt, r);
    for (int32_t i = 0; i < maxiters; i++) {
        oracle_csr_spmv(n, row_ptr, col_ind, values, p, q);
        oracle_csr_spmv_t(n, n, row_ptr, col_ind, values, pt, qt);
        double alpha = rho / oracle_ddot(n, pt, q);
        oracle_daxpy(n, alpha, p, x);
        oracle_daxpby(n, -alpha, q, 1.0, r);
        oracle_daxpby(n, -alpha, qt, 1.0, rt);
        double rr = oracle_ddot(n, r, r);
        if (rr <= tol * tol) { converged = 1; break; }
        double rho_new = oracle_ddot(n, rt, r);
        double beta = rho_new / rho;
        oracle_daxpby(n, 1.0, r, beta, p);
        oracle_daxpby(n, 1.0, rt, beta, pt);
        rho = rho_new;
        *iterations = i;
    }
done:
    free(r); free(rt); free(p); free(pt); free(q); free(qt);
    return converged;
}

/* ---- The reference's DFE stream format (src/runtime/Spmv.cpp:42-107,
 * Spmv.hpp:14-20, SparseMatrix.hpp:459-482), restated for the
 * device-function-triple compatibility path. ----
 *
 * oracle_partition_decode_spmv evaluates one partition exactly as the DFE
 * design does (SpmvKernel.java:61-78 + BramSpmvReductionKernel :250-309):
 * for each column block b, for each row r, add the products of the block's
 * packed {value,int32 index} records, gathering x at b*cache_size + index,
 * then add the per-block partials in block order.  colptr holds, per block,
 * n cumulative row ends (no leading 0; Spmv.cpp:69-79 copies row_ptr of
 * sliceColumns() which starts with row 0's END because :466-470 pushes one
 * entry per row).  Each block's record run is padded to a multiple of
 * input_width records (utils::align, Spmv.cpp:76-77, Utils.hpp:62-69).
 * Middle blocks may be run-length encoded (bit 31 set => a run of empty rows,
 * Spmv.hpp:213-237) when rle != 0. */
int oracle_partition_decode_spmv(int32_t n_rows, int32_t n_blocks, int32_t cache_size,
                                 int32_t input_width, int32_t rle,
                                 const int32_t *colptr, int64_t colptr_len,
                                 const uint8_t *records /* 12-byte packed */,
                                 int64_t n_records, const double *x, double *y)
{
    for (int32_t r = 0; r < n_rows; r++) y[r] = 0.0;
    int64_t cp = 0, rec = 0;
    for (int32_t b = 0; b < n_blocks; b++) {
        int encoded = rle && b != 0 && b != n_blocks - 1;
        int32_t row = 0;
        int64_t prev_end = 0, block_records = 0;
        while (row < n_rows) {
            if (cp >= colptr_len) return 2;
            uint32_t e = (uint32_t)colptr[cp++];
            if (encoded && (e & 0x80000000u)) { row += (int32_t)(e & 0x7fffffffu); continue; }
            int64_t len = (int64_t)e - prev_end;
            prev_end = e;
            double acc = 0.0;
            for (int64_t k = 0; k < len; k++) {
                if (rec >= n_records) return 3;
                double v; int32_t idx;
                memcpy(&v, records + 12 * rec, 8);
                memcpy(&idx, records + 12 * rec + 8, 4);
                acc += x[(int64_t)b * cache_size + idx] * v;
                rec++; block_records++;
            }
            y[row] += acc;
            row++;
        }
        /* skip the zero records that pad the block to a multiple of input_width */
        int64_t pad = (input_width - block_records % input_width) % input_width;
        rec += pad;
    }
    return 0;
}

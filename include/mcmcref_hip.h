/*
 * mcmcref_hip.h -- C ABI of libmcmcref_hip.so, the MI355X (gfx950) implementation of the
 * draw-vs-reference statistics hot path of StefanSko/mcmc-db (`mcmc_ref` 0.1.4).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  A Python
 * host binds it with ctypes (mcmc-db_amd/mcmc_ref_hip/_ffi.py; the reference-side stub is
 * shown in INTEGRATION.md).  Each entry point names the reference interface it replaces
 * (file:line relative to the reference repo root).
 *
 * Conventions
 *   - every function returns MCR_OK (0) or a negative MCR_E* code and never throws/aborts;
 *     mcr_last_error(ctx) returns a human-readable message for the last failure on ctx.
 *   - the caller owns every host buffer; the library owns all device scratch in mcr_ctx.
 *   - one mcr_ctx == one GPU + its HIP streams ("lanes", see mcr_init).  Calls on one ctx must be
 *     serialised by the caller; different ctxs are independent (one process per GPU, one ctx per
 *     thread, or several ctxs in one thread).
 *   - tensors are described by element strides (stride_c, stride_n, stride_p), so both the
 *     Arrow column layout [P][C][N] and `Draws.to_numpy` layout [C][N][P] (src/mcmc_ref/draws.py:28-29)
 *     are accepted without a host-side copy.
 *   - dtype: MCR_F64 (the reference's only dtype) or MCR_F32 (widened to f64 on load).
 *   - NaN/Inf in the draws are rejected with MCR_ENONFINITE (the reference's behaviour for
 *     them is undefined: sort order with NaN).
 */
#ifndef MCMCREF_HIP_H
#define MCMCREF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCR_VERSION 100 /* 0.1.0 */

#define MCR_OK 0
#define MCR_EINVAL (-1)         /* null pointer, negative size, unsupported dtype/quantile count */
#define MCR_EMINCHAINS (-2)     /* C < min_chains: "... require at least k chains; got m chain(s)"
                                   (src/mcmc_ref/diagnostics.py:25-28, 49-52, 65-68) */
#define MCR_EMINCHAINS_ARG (-3) /* min_chains < 1: "min_chains must be >= 1; got k" (diagnostics.py:88-90) */
#define MCR_ENONFINITE (-4)     /* NaN/Inf found in the draws */
#define MCR_EHIP (-5)           /* HIP runtime error (message has the hipError string) */
#define MCR_ENOMEM (-6)         /* host or device allocation failed */
#define MCR_ENODEVICE (-7)      /* no usable HIP device / bad device index */
#define MCR_ECOMM (-8)          /* RCCL error (multi-GPU gather) */
#define MCR_ELAYOUT (-9)        /* mcr_summarize_files: rows not in (chain, draw) order or chains of unequal length */

#define MCR_F64 0
#define MCR_F32 1

#define MCR_MAX_QUANTILES 32

typedef struct mcr_ctx mcr_ctx;

/* Per-parameter results, struct-of-arrays, caller allocated.  Every non-NULL array has P
 * entries, except q (P * n_q, row-major [p][k]) and q_lo (n_q).  NULL members are skipped; when
 * every diagnostics member (rhat* / ess_* / lag_*) is NULL the rank / fold / autocovariance
 * kernels are not launched at all (Backend.stats-only call).
 *
 *  mean, std, q   Backend.stats(): pooled mean, population std (ddof=0), linear-interpolated
 *                 quantiles (src/mcmc_ref/backends.py:14-24, backends_arrow.py:36-51,
 *                 backends_numpy.py:40-47).  q_lo[k] = floor((M-1)*q_k): the order-statistic
 *                 index (integer, bit-exact gate).
 *  median         statistics.median of the pooled draws, the fold point (diagnostics.py:97).
 *  rhat           split_rhat(): max(rhat_bulk, rhat_tail) with Python max() NaN ordering
 *                 (diagnostics.py:13-40); rhat_bulk / rhat_tail are its two operands.
 *  ess_bulk/tail  ess_bulk(), ess_tail() (diagnostics.py:43-73).
 *  lag_bulk/tail  number of autocorrelation terms accumulated before the first negative rho
 *                 (diagnostics.py:171-177) -- integer, bit-exact gate.
 */
typedef struct mcr_summary {
    double* mean;
    double* std;
    double* q;
    double* median;
    double* rhat;
    double* rhat_bulk;
    double* rhat_tail;
    double* ess_bulk;
    double* ess_tail;
    int64_t* lag_bulk;
    int64_t* lag_tail;
    int64_t* q_lo;
} mcr_summary;

/* Accumulated HIP-event time of one kernel since profiling was last reset. */
typedef struct mcr_kernel_time {
    char name[48];
    int64_t launches;
    double total_ms;
} mcr_kernel_time;

/* ---- lifecycle -------------------------------------------------------------------- */
int mcr_version(void);
int mcr_device_count(void);
/* Creates a context bound to HIP device `device` (own non-blocking streams, lazily grown
 * workspaces).  Fails with MCR_ENODEVICE when no GPU is present: there is no CPU fallback.
 * Environment read here: MCR_LANES (streams + workspaces that consecutive calls rotate over,
 * default 4, max MCR_MAX_INFLIGHT), MCR_GRAPH=1 (hipGraph capture / replay of the launch
 * sequence; off by default), MCR_WORKSPACE_MB (see mcr_set_workspace_limit).
 * Limits: C <= 256 chains, C * N < 2^31 pooled draws per parameter (rank codes are 32-bit),
 * any number of parameters (chunked through the workspace). */
int mcr_init(int device, mcr_ctx** out);
void mcr_free(mcr_ctx* ctx);
const char* mcr_last_error(const mcr_ctx* ctx); /* ctx may be NULL: last mcr_init failure */
/* Upper bound for device scratch (bytes); parameters are processed in chunks that fit.
 * Default: MCR_WORKSPACE_MB env or 8192 MiB. */
int mcr_set_workspace_limit(mcr_ctx* ctx, size_t bytes);

/* How the parameters of a call of this shape are cut into workspace chunks: parameters [k * n, (k + 1) * n) are
 * processed together, n = *params_per_chunk (a function of the shape, the strides' layout class, the workspace limit and
 * MCR_FFT).  The reference has no counterpart (its loop is per parameter, src/mcmc_ref/convert.py:140-147); tests and
 * bench.py use it to put oracle-checked parameters on both sides of every chunk edge. */
int mcr_plan_chunks(mcr_ctx* ctx, int64_t C, int64_t N, int64_t P, int64_t stride_c, int64_t stride_n, int64_t stride_p,
                    int diagnostics, int64_t* params_per_chunk);

/* How many autocorrelation lags this context has re-derived the reference's way so far: no tier of the ESS walk takes
 * a `rho < 0` decision (src/mcmc_ref/diagnostics.py:171-177) on a value within MCR_RHO_BAND (default 1e-10) of zero --
 * segment records + mean correction below lag 256, tree sums or FFTs beyond; such a lag is recomputed with _autocorr's
 * own left-to-right sums and the chain's left-to-right mean (diagnostics.py:180-193) first.  A diagnostic: tests use it
 * to show that the guard ran.  Waits for the calls in flight. */
int mcr_rho_guard_count(mcr_ctx* ctx, int64_t* rederived);

/* ---- device memory plumbing (for device-resident benchmarking and pipelines) --------- */
int mcr_dev_alloc(mcr_ctx* ctx, size_t bytes, void** dptr);
int mcr_dev_free(mcr_ctx* ctx, void* dptr);
int mcr_memcpy_h2d(mcr_ctx* ctx, void* dptr, const void* hptr, size_t bytes);
int mcr_memcpy_d2h(mcr_ctx* ctx, void* hptr, const void* dptr, size_t bytes);
int mcr_sync(mcr_ctx* ctx);

/* ---- the hot path ------------------------------------------------------------------ */
/* Everything the reference computes per parameter over a (C chains x N draws x P params)
 * tensor, in one call: replaces the per-parameter Python loops of
 *   convert._compute_diagnostics            (src/mcmc_ref/convert.py:134-147)
 *   reference.diagnostics_for_model         (src/mcmc_ref/reference.py:92-104)
 *   Backend.stats                            (src/mcmc_ref/backends_arrow.py:36-51)
 * `draws` is a HOST pointer (copied to the device inside the call). Synchronous. */
int mcr_summarize(mcr_ctx* ctx, const void* draws, int dtype, int64_t C, int64_t N, int64_t P,
                  int64_t stride_c, int64_t stride_n, int64_t stride_p, int min_chains,
                  const double* quantiles, int n_q, mcr_summary* out);
/* Same with `draws` already resident in this ctx's device memory.  Synchronous. */
int mcr_summarize_dev(mcr_ctx* ctx, const void* draws_dev, int dtype, int64_t C, int64_t N,
                      int64_t P, int64_t stride_c, int64_t stride_n, int64_t stride_p,
                      int min_chains, const double* quantiles, int n_q, mcr_summary* out);
/* Asynchronous form: enqueues the whole pipeline on the ctx stream and returns; results
 * land in `out` (which must stay valid) when mcr_summarize_wait() returns.  At most
 * MCR_MAX_INFLIGHT enqueues may be outstanding per ctx. */
#define MCR_MAX_INFLIGHT 8
int mcr_summarize_enqueue(mcr_ctx* ctx, const void* draws_dev, int dtype, int64_t C, int64_t N,
                          int64_t P, int64_t stride_c, int64_t stride_n, int64_t stride_p,
                          int min_chains, const double* quantiles, int n_q, mcr_summary* out);
int mcr_summarize_wait(mcr_ctx* ctx);
/* Waits for the OLDEST outstanding enqueue only and fills its `out`; later enqueues keep running, so a
 * caller can hold a rolling window of MCR_MAX_INFLIGHT calls without ever draining the device. */
int mcr_summarize_wait_one(mcr_ctx* ctx);

/* Many independent models in one call (BASELINE configs 2/3: the packaged corpus): every model is a
 * device-resident tensor of this ctx; the calls are pipelined through the lanes with a rolling window
 * of MCR_MAX_INFLIGHT, so small models overlap.  Replaces the per-model loop of
 * generate.generate_reference_corpus -> convert_file (src/mcmc_ref/generate.py:77-96).  outs[i] receives
 * model i.  Stops at the first failing model and returns its code (earlier models are delivered). */
typedef struct mcr_model_desc {
    const void* draws_dev;
    int dtype;
    int min_chains;
    int64_t C, N, P;
    int64_t stride_c, stride_n, stride_p;
} mcr_model_desc;
int mcr_summarize_models(mcr_ctx* ctx, const mcr_model_desc* models, int n_models, const double* quantiles,
                         int n_q, mcr_summary* outs);

/* diagnostics.split_rhat / ess_bulk / ess_tail for ONE parameter given as possibly ragged
 * chains (src/mcmc_ref/diagnostics.py:13-73): `pooled` holds the chains back to back, chain c
 * is pooled[chain_off[c] .. chain_off[c+1]).  Host pointers.  out arrays have 1 entry.
 * Optional debug outputs (host, length chain_off[C], may be NULL): z_bulk / z_tail =
 * _rank_normalize(x) / _rank_normalize(_fold_chains(x)) (diagnostics.py:93-133), rank_bulk /
 * rank_tail = the average ranks (exact multiples of 0.5). */
int mcr_diagnose_chains(mcr_ctx* ctx, const double* pooled, const int64_t* chain_off, int C,
                        int min_chains, mcr_summary* out, double* z_bulk, double* z_tail,
                        double* rank_bulk, double* rank_tail);

/* compare.compute_basic_stats (src/mcmc_ref/compare.py:58-64): mean and population std of n
 * host values; n == 0 gives NaN, NaN.  Also the streaming "moments" kernel (HBM-bound). */
int mcr_basic_stats(mcr_ctx* ctx, const void* values, int dtype, int64_t n, double* mean,
                    double* std);
/* Pooled mean / population std per parameter of a device-resident tensor (one HBM pass). */
int mcr_moments_dev(mcr_ctx* ctx, const void* draws_dev, int dtype, int64_t C, int64_t N,
                    int64_t P, int64_t stride_c, int64_t stride_n, int64_t stride_p,
                    double* mean, double* std);
/* compare.compare_stats inner arithmetic (src/mcmc_ref/compare.py:38-43) on n (ref, actual)
 * pairs: rel = |a-r| / max(|r|, 1e-12), pass = rel <= tol (NaN -> 0).  Host pointers. */
int mcr_compare(mcr_ctx* ctx, const double* ref, const double* actual, int64_t n, double tol,
                double* rel_error, uint8_t* passed);

/* ---- extensions (named in the north star, ABSENT from the reference: parity unpinned by it) ----- */
/* Two-sample Kolmogorov-Smirnov statistic and Wasserstein-1 distance per parameter between the
 * reference draws ref[P][Mr] and the actual draws act[P][Ma] (host pointers, row-major, finite).
 * Definitions follow scipy.stats.ks_2samp(...).statistic and scipy.stats.wasserstein_distance:
 * both samples are sorted on the device and compared in one merge-path pass.  SURVEY.md rows X1, X2. */
int mcr_two_sample(mcr_ctx* ctx, const double* ref, int64_t Mr, const double* act, int64_t Ma, int64_t P,
                   double* ks, double* w1);
/* Population covariance matrix (ddof = 0, like compare.py:63) of P parameters over M pooled draws,
 * draws[P][M] host row-major -> cov[P][P].  The one dense contraction of the path: fp64 MFMA
 * (v_mfma_f64_16x16x4f64; LDS-staged 64 x 64 tiles, upper triangle, split over the draw axis).
 * numpy.cov(x, ddof=0).  SURVEY.md row X3. */
int mcr_covariance(mcr_ctx* ctx, const double* draws, int64_t M, int64_t P, double* cov);
/* The same on device-resident buffers of this ctx (draws_dev [P][M], cov_dev [P][P]); P <= 8192. */
int mcr_covariance_dev(mcr_ctx* ctx, const double* draws_dev, int64_t M, int64_t P, double* cov_dev);

/* ---- measurement ------------------------------------------------------------------- */
/* When on, every kernel launch is bracketed by HIP events on the ctx stream. */
int mcr_profile_enable(mcr_ctx* ctx, int on);
int mcr_profile_reset(mcr_ctx* ctx);
/* Synchronises, resolves the events and fills up to `max` entries; *n = entries available. */
int mcr_profile_get(mcr_ctx* ctx, mcr_kernel_time* out, int max, int* n);
/* Fills a device buffer with the synthetic stress tensor of SURVEY.md 8(d) C4 (iid
 * N(p, sigma_p), counter-based, layout [P][C][N]) without touching the host. */
int mcr_fill_synthetic(mcr_ctx* ctx, void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P,
                       uint64_t seed);

/* What this device's HBM delivers, measured: best of `iters` passes of a read-only streaming kernel over `bytes`
 * (read_gbps) and of a device-to-device copy (copy_gbps = bytes read + written per second); either may be NULL.
 * bench.py prints it as `peak_measured` beside the 8 TB/s specification peak. */
int mcr_hbm_probe(mcr_ctx* ctx, size_t bytes, int iters, double* read_gbps, double* copy_gbps);
/* The same for the parameter block [p0, p0 + P) of that tensor: draws_dev receives P * C * N elements that equal the
 * corresponding slice of the whole tensor (a rank of a P-split model generates only its own columns). */
int mcr_fill_synthetic_at(mcr_ctx* ctx, void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P, int64_t p0,
                          uint64_t seed);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, models / parameter blocks sharded with no data-path
 * exchange, and ONE collective at the end -- an RCCL all-gather of fixed-size per-parameter summary
 * records over xGMI.  The reference has no counterpart (it is single-process); this replaces the loop
 * over models of generate.generate_reference_corpus (src/mcmc_ref/generate.py:77-96) being run on N
 * devices.  librccl is loaded on first use (dlopen); failures return MCR_ECOMM.
 * Ranks must agree on a 128-byte id: rank 0 calls mcr_comm_unique_id and hands the bytes to the others
 * by any means (mcmc_ref_hip.shard uses a file keyed on MASTER_ADDR / MASTER_PORT), then every rank
 * calls mcr_comm_init (collective).  All buffers are host pointers; calls are synchronous.
 * EVERY collective has a deadline: ncclCommInitRank runs on a helper thread the caller watches, and the
 * waits for all-gather / all-reduce / barrier poll the communicator's stream and ncclCommGetAsyncError
 * against MCR_COMM_TIMEOUT_S (default 300).  A peer that never arrives or dies -- the reference's loop
 * simply continues past a failed recipe, src/mcmc_ref/generate.py:77-96 -- ends the call with MCR_ECOMM
 * naming the call and the rank after that time instead of blocking for ever; the communicator is
 * aborted (ncclCommAbort) and every later call on it fails at once.  MCR_COMM_NONBLOCKING=1 creates
 * the communicator with ncclCommInitRankConfig(blocking = 0); MCR_COMM_BLOCKING=1 switches the
 * deadlines off (plain blocking calls); mcr_comm_has_deadline tells.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mcr_comm mcr_comm;
#define MCR_COMM_ID_BYTES 128
#define MCR_RECORD_DOUBLES 16 /* one per-parameter summary record = 128 bytes */
int mcr_comm_unique_id(void* id, size_t len);
int mcr_comm_init(mcr_ctx* ctx, const void* id, int world, int rank, mcr_comm** out);
void mcr_comm_free(mcr_comm* comm);
int mcr_comm_world(const mcr_comm* comm);
int mcr_comm_rank(const mcr_comm* comm);
int mcr_comm_has_deadline(const mcr_comm* comm); /* 1: every wait on this communicator is bounded */
/* ncclAllGather: every rank sends `count` doubles and receives world * count doubles in rank order. */
int mcr_comm_all_gather(mcr_comm* comm, const double* send, int64_t count, double* recv);
/* ncclAllReduce in place over n doubles; op: 0 = sum, 1 = max, 2 = min (max-over-ranks clock, all-valid flag). */
int mcr_comm_all_reduce(mcr_comm* comm, double* vals, int64_t n, int op);
/* Drains this context's lanes, then synchronises the ranks. */
int mcr_comm_barrier(mcr_comm* comm);

/* ------------------------------------------------------------------------------------------------
 * Parquet ingest: draws file -> device tensor.
 * Replaces pq.read_table / pq.ParquetFile + to_numpy on the way into the statistics
 * (src/mcmc_ref/store.py:79-95 open_draws, src/mcmc_ref/convert.py:61-65, backends_numpy.py:35):
 * footer and page headers are parsed on the host, page payloads are Snappy-decompressed and
 * PLAIN / RLE_DICTIONARY decoded by HIP kernels straight into device memory.
 * Supported: flat schemas; INT32 / INT64 / FLOAT / DOUBLE columns, REQUIRED or OPTIONAL without
 * nulls; UNCOMPRESSED / SNAPPY; data pages v1 and v2; any number of row groups and pages.
 * Anything else fails with MCR_EINVAL and a message naming the feature.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mcr_parquet mcr_parquet;

#define MCR_PQ_F64 0 /* out_dev is double[num_rows]  (ints and floats are converted to double) */
#define MCR_PQ_I64 1 /* out_dev is int64_t[num_rows] (INT32 / INT64 columns only: chain, draw) */

/* Physical types as in parquet.thrift (mcr_parquet_column_type). */
#define MCR_PQ_BOOLEAN 0
#define MCR_PQ_INT32 1
#define MCR_PQ_INT64 2
#define MCR_PQ_INT96 3
#define MCR_PQ_FLOAT 4
#define MCR_PQ_DOUBLE 5
#define MCR_PQ_BYTE_ARRAY 6
#define MCR_PQ_FIXED_LEN_BYTE_ARRAY 7

typedef struct {
    const mcr_parquet* file;
    int column;    /* index into the file's (flat) schema */
    int out_kind;  /* MCR_PQ_F64 or MCR_PQ_I64 */
    void* out_dev; /* device pointer, num_rows 8-byte elements, rows in file order */
} mcr_parquet_request;

/* Parses the metadata of a Parquet file image held in host memory.  `bytes` must stay valid and
 * unchanged until mcr_parquet_close (the image is not copied).  Needs no device: ctx may be NULL
 * (the message of a failure is then read with mcr_last_error(NULL)). */
int mcr_parquet_open(mcr_ctx* ctx, const void* bytes, size_t len, mcr_parquet** out);
void mcr_parquet_close(mcr_parquet* f);
int64_t mcr_parquet_num_rows(const mcr_parquet* f);
int mcr_parquet_num_columns(const mcr_parquet* f);
const char* mcr_parquet_column_name(const mcr_parquet* f, int column); /* NULL if out of range */
int mcr_parquet_column_type(const mcr_parquet* f, int column);         /* -1 if out of range */
/* Page table as parsed from the page headers (introspection / tests).  info[10] = {column, page
 * type, value encoding, codec, payload file offset, compressed size, uncompressed size, values,
 * first row, page index of the chunk's dictionary page or -1}. */
int mcr_parquet_num_pages(const mcr_parquet* f);
int mcr_parquet_page_info(const mcr_parquet* f, int page, int64_t* info);

/* Decodes the requested columns (of one or many files) with ONE upload + two kernel launches;
 * synchronous on return.  Requests may mix files, columns and output kinds. */
int mcr_parquet_decode(mcr_ctx* ctx, const mcr_parquet_request* reqs, int n_reqs);

/* dst[p][k] = src[p][order[k]]: puts rows that are not stored in (chain, draw) order into the
 * order `_chains_from_table` produces (src/mcmc_ref/convert.py:150-161).  order is a host array. */
int mcr_gather_rows_dev(mcr_ctx* ctx, const double* src_dev, int64_t P, int64_t M, const int64_t* order,
                        double* dst_dev);

/* ------------------------------------------------------------------------------------------------
 * Many draws files -> statistics in ONE call (the per-model loop of reference.stats /
 * diagnostics_for_model, src/mcmc_ref/reference.py:30-104, over a list of
 * draws/<model>.draws.parquet files): mmap + footer parse on the host, one batched GPU decode, the
 * chain / draw bookkeeping of convert._chains_from_table, same-shape neighbours summarised as one
 * tensor, results in a host-side set.  Parameters = every numeric column except `chain` and `draw`,
 * in schema order.  Files whose rows are not in (chain, draw) order, or whose chains differ in
 * length while diagnostics are requested, end the call with MCR_ELAYOUT (use mcr_parquet_decode +
 * mcr_gather_rows_dev + mcr_diagnose_chains for those).  diagnostics = 0: Backend.stats only
 * (pooled mean / std / quantiles; any chain structure).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mcr_fileset mcr_fileset;
#define MCR_FS_MEAN 0
#define MCR_FS_STD 1
#define MCR_FS_Q 2 /* [P][n_q] */
#define MCR_FS_MEDIAN 3
#define MCR_FS_RHAT 4
#define MCR_FS_ESS_BULK 5
#define MCR_FS_ESS_TAIL 6
#define MCR_FS_RHAT_BULK 7
#define MCR_FS_RHAT_TAIL 8
#define MCR_FS_LAG_BULK 9 /* truncation lags as doubles */
#define MCR_FS_LAG_TAIL 10
#define MCR_FS_FIELDS 11
int mcr_summarize_files(mcr_ctx* ctx, const char* const* paths, int n_paths, int min_chains,
                        const double* quantiles, int n_q, int diagnostics, mcr_fileset** out);
int mcr_fileset_size(const mcr_fileset* fs);
int64_t mcr_fileset_params(const mcr_fileset* fs, int file);
int64_t mcr_fileset_chains(const mcr_fileset* fs, int file);
int64_t mcr_fileset_draws(const mcr_fileset* fs, int file); /* draws per chain (of the first chain when they differ) */
const char* mcr_fileset_param_name(const mcr_fileset* fs, int file, int64_t param);
const double* mcr_fileset_field(const mcr_fileset* fs, int file, int field); /* P doubles (MCR_FS_Q: P * n_q) */
/* The whole set in two blocks, for callers that turn it into dictionaries (the shape reference.stats /
 * diagnostics_for_model return, src/mcmc_ref/reference.py:30-104) without a call per file and field:
 * mcr_fileset_export writes one row of 10 + n_q doubles per parameter, files and parameters in order -- mean, std,
 * median, rhat, ess_bulk, ess_tail, rhat_bulk, rhat_tail, lag_bulk, lag_tail, q[0 .. n_q) -- and returns the number of
 * rows of the set (at most cap_rows are written; rows may be NULL to ask); mcr_fileset_names writes the parameters'
 * names in the same order, each terminated by NUL, and returns the bytes that takes (written only if they fit cap). */
int64_t mcr_fileset_export(const mcr_fileset* fs, double* rows, int64_t cap_rows);
int64_t mcr_fileset_names(const mcr_fileset* fs, char* buf, int64_t cap);
/* Where the host-clock time of the mcr_summarize_files call that built `fs` went, in milliseconds (the path it replaces,
 * pq.read_table + the per-parameter loops of src/mcmc_ref/store.py:79-95 / reference.py:30-104, is host-bound, so the
 * split is part of the measurement): ms[MCR_FS_PH_*]; returns MCR_FS_PHASES (at most `cap` entries are written). */
#define MCR_FS_PH_OPEN 0    /* open + fstat of every file, staging buffers */
#define MCR_FS_PH_READ 1    /* MCR_IO_THREADS host threads (default 8): pread of the whole file images into pinned memory,
                               footer + page-header parse from there; uploads issued behind them in >= 2 MB pieces */
#define MCR_FS_PH_PLAN 2    /* request list, page table, table uploads */
#define MCR_FS_PH_DECODE 3  /* wait for the uploads + Snappy / page decode kernels + the chain / draw layout kernel */
#define MCR_FS_PH_STATS 4   /* the statistics pipeline of every job (enqueue, kernels, result copies) */
#define MCR_FS_PH_COLLECT 5 /* results into the set */
#define MCR_FS_PH_CLOSE 6   /* close, free of the parsed metadata */
#define MCR_FS_PH_TOTAL 7
#define MCR_FS_PHASES 8
int mcr_fileset_phases(const mcr_fileset* fs, double* ms, int cap);
/* How many tensors (kernel pipelines) the files of the set were summarised as: files of one (chains, draws) shape that sit
 * next to each other in the call's arena -- ordered by the chain count the footers' column statistics suggest -- are one. */
int mcr_fileset_jobs(const mcr_fileset* fs);
void mcr_fileset_free(mcr_fileset* fs);

#ifdef __cplusplus
}
#endif
#endif /* MCMCREF_HIP_H */

/* gato_hip.h - C ABI of the MI355X-native gato PCG / Schur hot path (libgato_hip.so).
 *
 * Drop-in boundary: these entry points are what the reference's binding layer would call
 * instead of its CUDA driver.  Every function cites the reference interface it replaces
 * (file:line relative to the reference checkout).  Plain pointers and sizes only; no torch
 * or pybind11 types.  All functions return 0 on success or a negative GATO_E* code, and
 * gato_last_error() gives the message (the reference prints "GPUassert" and exit()s,
 * include/gato_defines.h:42-51).
 *
 * Conventions:  S = STATE_SIZE, C = CONTROL_SIZE, K = KNOT_POINTS (runtime here; compile-time
 * macros in the reference, CMakeLists.txt:18), n = S+C, N = n*K - C.
 * dtype: GATO_F32 (the reference's only arithmetic type) or GATO_F64.  `void*` data pointers
 * are float* or double* according to the solver's dtype.  Device layouts are the reference's:
 *   G_dense : per knot [Q_k S*S | R_k C*C] col-major, last knot Q only  ((S*S+C*C)*K - C*C elems, gato_defines.h:36)
 *   C_dense : per knot k<K-1 [A_k S*S | B_k S*C] col-major              ((S*S+S*C)*(K-1) elems, gato_defines.h:37)
 *   S, Pinv : per block-row [left|main|right], each S*S col-major       (3*S*S*K elems, gato_utils.cuh:44-73)
 *   gamma, lambda : S*K ; g, dz : N ; c : S*K
 * `stream` is a hipStream_t passed as void* (NULL = default stream, as the reference uses,
 * gato_defines.h:22).
 */
#ifndef GATO_HIP_H
#define GATO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GATO_F32 0
#define GATO_F64 1

#define GATO_OK 0
#define GATO_EINVAL (-1)   /* bad argument / shape mismatch */
#define GATO_ESHAPE (-2)   /* (S,C) has no compiled instantiation */
#define GATO_EHIP (-3)     /* HIP runtime error */
#define GATO_ENODEV (-4)   /* no usable GPU */
#define GATO_ETIMEOUT (-5) /* in-kernel hand-off timed out (persistent PCG) */

/* PCG kernel selection (gato_pcg.cuh:505-553 picks K4 vs K5 by co-residency, check_sms) */
#define GATO_PCG_AUTO 0
#define GATO_PCG_RESIDENT 1  /* A5: matrices register-resident for the whole solve, one persistent launch */
#define GATO_PCG_STREAMING 2 /* A6: matrices re-read from HBM every iteration, two launches per iteration */

/* Preconditioner of the whole-solve entries = the reference's compile switches BLOCK_J_PRECON / SS_PRECON
 * (include/gato_defines.h:9-10; src/gato_schur.cuh:407-429,965-970), a runtime option here ("precon_mode"). */
#define GATO_PRECON_STAIR 0        /* both 1 (the reference's setting): 3-band symmetric stair */
#define GATO_PRECON_BLOCK_JACOBI 1 /* SS_PRECON 0: main blocks -theta^-1 only */
#define GATO_PRECON_POINT_JACOBI 2 /* both 0: diag(1 / S.main_ii) */

/* Threads: a gato_solver is used by one host thread at a time (its plan, counters and work buffers are per solver); different
 * solvers may be driven from different threads and streams concurrently - the process-wide state (the admission of persistent
 * multi-workgroup launches to the chip, the cached solver of gato_linsys_solve_*, the mirror pool) is locked, gato_last_error is
 * per thread. */
typedef struct gato_solver gato_solver;

/* ---- library / device ------------------------------------------------------------------- */
const char *gato_last_error(void);
int gato_version(void);
/* Number of (S,C) instantiations compiled in, and the i-th one. */
int gato_num_shapes(void);
int gato_shape(int i, int *S, int *C);
/* Replaces check_sms (gato_utils.cuh:829-854): device properties needed to size the grids. */
int gato_device_info(int device, int *num_cus, int *lds_bytes, char *name, int name_len);

/* Infers (S,C,K) from the lengths of the linsys_solve arguments (the reference gets them from
 * -DSTATE_SIZE/-DCONTROL_SIZE/-DKNOT_POINTS, install.bash:6-16): S*K = len_c, N = len_g, and
 * S = number of leading single-entry identity rows of C (row-block 0 of C, gato_schur.cuh:725). */
int gato_infer_shape(const int *C_row, int len_C_row, int len_g, int len_c, int *S, int *C, int *K);

/* ---- solver object: owns the device workspace the reference allocates per call
 * (gpu_library.cu:36-45, gato_pcg.cuh:486-492) -------------------------------------------- */
int gato_solver_create(int S, int C, int K, int dtype, int device, gato_solver **out);
/* Batch of B independent systems of one shape and one sparsity pattern (new - the
 * reference solves one system per call, its `testiters` loop re-solves the same one, gpu_library.cu:169).
 * Every per-system device array of the stage-level calls is then B arrays back to back; the CSR structure
 * (indptr/indices) is shared, G_val / C_val hold nnz entries per system.  All stages run the whole batch
 * in one launch each (grid.y = system); the PCG runs one workgroup per system when a system fits one CU. */
int gato_solver_create_batched(int S, int C, int K, int B, int dtype, int device, gato_solver **out);
int gato_solver_destroy(gato_solver *s);
/* Workspace device pointers (valid for the solver's lifetime), for stage-level tests:
 * which: 0 G_dense, 1 C_dense, 2 Ginv_dense, 3 S, 4 Pinv, 5 gamma, 6 lambda, 7 dz, 8 iters(int),
 * 10 eta history (double[max_iters+1]: eta = r.Pinv r after the initial step and after every iteration; filled when
 * option record_eta = 1 and max_iters <= 4096 - the reference only prints it under DEBUG_MODE, gato_pcg.cuh:397-400) */
void *gato_solver_buffer(gato_solver *s, int which);
/* Options: pcg_mode (GATO_PCG_*), pcg_threads (0 = auto; threads per workgroup of the resident
 * kernel), pcg_groups (0 = auto; workgroups of the resident kernel), true_warm_start (0 = the
 * reference's behaviour: lambda restarts from zero, gato_pcg.cuh:303; 1 = d_lambda of gato_pcg /
 * gato_linsys_device is read as the initial guess, r0 = gamma - S lambda0), pcg_variant (0 = the reference's PCG
 * recurrence; 1 = opt-in single-reduction Chronopoulos-Gear recurrence of the multi-workgroup and the cluster launches: one
 * inter-workgroup / cross-GPU exchange per iteration instead of two, same solution to solver tolerance, different rounding),
 * coop_launch (1 = the multi-workgroup resident / semi-resident launches through hipLaunchCooperativeKernel - what the
 * reference does, gato_pcg.cuh:502-526 - so that the runtime guarantees their co-residency beside kernels of other streams
 * and processes; +17 us per launch, default 0; the single-reduction kernel and the LDS-DMA ring keep the plain launch),
 * xcd_pack (-1 auto:
 * launches of up to 32 workgroups are placed on one XCD - a placement hint, never needed for correctness; 0 off),
 * xcd_sel (which of the eight XCDs hosts such a launch: -1 = measured once per solver and geometry with a millisecond of
 * trial launches before the first one, 0..7 fixed; read-only last_xcd_sel),
 * asm_mode (whole-solve entries: 0 = auto - convert + Schur + stair as ONE fused launch when K*B <= 2 x CUs, the
 * stage kernels otherwise; 1 = stage kernels; 2 = fused; both give bit-identical buffers), pcg_semi (-1 = auto: K
 * beyond the register file runs as one persistent launch - semi-resident, or with the block rows streamed through an
 * LDS-DMA ring once the matrices one launch streams are far beyond the Infinity Cache (measured cross-overs: 450 MB of S + Pinv
 * in fp32, 550 MB in fp64, 700 MB at STATE_SIZE 32); 0 = the streaming kernels; 1 / 2 / 3 force the
 * semi-resident launch with / without resident rows / the LDS-DMA ring), time_pcg (record
 * hipEvents around the PCG launch), time_stages (hipEvents around assembly / PCG / dz of the whole-solve entries),
 * precon_mode (GATO_PRECON_*), knot_lo / knot_hi (the stage-level entries gato_convert / gato_form_schur / gato_form_ss /
 * gato_compute_dz then work on the knots [knot_lo, knot_hi) only - a rank of a multi-GPU solve assembles just what its
 * PCG shard reads; reset by every whole-solve call), timeout_ms (bound of every in-kernel spin, default 2000), max_workgroups (CUs a
 * persistent launch may count on; 0 = all of the device), no_single_lds / stamp_pcg (1 = kernel build with cycle stamps and the
 * timing-only switches of `ablate`, 2 = the switches alone) / stamp_asm / ablate (diagnostics). */
int gato_solver_set_option(gato_solver *s, const char *name, int value);
int gato_solver_get_option(gato_solver *s, const char *name, int *value);

/* ---- stage-level entry points on DEVICE pointers (one per reference launch wrapper) ------ */
/* A1  form_schur's first launch, gato_convert_kkt_format (gato_schur.cuh:745-756, :902).
 * Zeroes G_dense/C_dense itself (the reference relies on cuda_calloc, gpu_library.cu:36-37). */
int gato_convert(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val,
                 const int *d_C_row, const int *d_C_col, const void *d_C_val, double rho,
                 void *d_G_dense, void *d_C_dense, void *stream);
/* A2  gato_form_schur_jacobi (gato_schur.cuh:462-494, :942).  Writes S (left, main, right),
 * Pinv.main, gamma and the inverses Q^-1,R^-1 into d_Ginv_dense (separate buffer; the
 * reference overwrites d_G_dense in place, :238-259, racing with its neighbours - D3). */
int gato_form_schur(gato_solver *s, const void *d_G_dense, const void *d_C_dense, const void *d_g,
                    const void *d_c, void *d_S, void *d_Pinv, void *d_gamma, void *d_Ginv_dense,
                    void *stream);
/* A3  gato_form_ss (gato_schur.cuh:652-670, :967): Pinv.left / Pinv.right. */
int gato_form_ss(gato_solver *s, const void *d_S, void *d_Pinv, void *stream);
/* A4-A8  solve_pcg<T> (gato_pcg.cuh:476-567).  lambda is reset to 0 (D5: warm_start is a no-op in
 * the reference, gato_pcg.cuh:303).  d_iters receives the reference's iteration count (index of
 * the iteration that met |eta| < exit_tol, else max_iters; gato_pcg.cuh:311-313,:406-408; -1 = a hand-off of a
 * persistent launch timed out, see gato_pcg_status).  Asynchronous on `stream`: enqueue only - no host wait, no
 * trial launches, nothing but d_lambda / d_iters written for the caller (checked by
 * tests/test_gpu_parity.py::test_pcg_entry_is_enqueue_only). */
int gato_pcg(gato_solver *s, const void *d_S, const void *d_Pinv, const void *d_gamma,
             void *d_lambda, double exit_tol, int max_iters, int *d_iters, void *stream);
/* Placement of the one-XCD persistent launches (2..32 workgroups; replaces nothing in the reference - its cooperative
 * launch, gato_pcg.cuh:502-526, has no notion of placement): measures once which of the eight XCDs hosts the geometry the
 * solver's CURRENT options plan (16 short trial launches on solver-owned scratch buffers, each waited for: BLOCKING,
 * ~1 ms).  gato_solver_create calls it for the default geometry (env GATO_NO_TUNE=1 skips that); call it again after
 * changing pcg_threads / pcg_groups / pcg_variant / max_workgroups / xcd_pack.  A geometry that was never measured runs
 * on XCD 0; results never depend on the placement.  No-op for batches, cluster ranks and other geometries. */
int gato_solver_tune(gato_solver *s, void *stream);
/* Hand-off time-outs (the workgroups of a persistent launch were not co-resident - the case the reference excludes
 * with cudaLaunchCooperativeKernel + check_sms, gato_pcg.cuh:502-526, gato_utils.cuh:829-854): the launch writes
 * iters = -1 (in-band, no second call needed) and the id of the launch into the solver's status word, which no kernel
 * ever clears.  gato_pcg_status synchronises the stream of the latest PCG launch and returns GATO_ETIMEOUT if ANY
 * launch timed out since the previous call (*status = 1), else GATO_OK.  Option timeout_ms bounds every spin (2000). */
int gato_pcg_status(gato_solver *s, int *status);
/* The fallback: if a persistent launch of the most recent gato_linsys_device / _blocks call timed out, re-run its PCG
 * through the streaming kernels (no in-launch hand-off, any residency) and recompute dz into the same output buffers -
 * a slower correct answer instead of an error.  Synchronises `stream`; *recovered = 1 if that happened.  Multi-
 * workgroup launches of one process never get there by themselves: a launch that does not fit beside those still in
 * flight on other streams waits for them (per-device CU budget). */
int gato_solver_recover(gato_solver *s, int *recovered, void *stream);
/* Device time of the most recent PCG launch(es) of gato_pcg, measured with hipEvents recorded on the
 * launch stream immediately around the kernel launch(es) (the reference times whole solves with
 * cudaEvents, gpu_library.cu:167-187).  Enabled by option "time_pcg" = 1; synchronises on the stop event. */
int gato_pcg_last_ms(gato_solver *s, float *ms);
/* Stage times of the most recent gato_linsys_device / _blocks call as data (the reference prints them:
 * "Forming Schur took", gato_schur.cuh:907-913,972-982; solve time gpu_library.cu:186-198): ms[0] = scatter + Schur +
 * preconditioner, ms[1] = PCG, ms[2] = dz.  Enabled by option "time_stages" = 1. */
int gato_last_stage_ms(gato_solver *s, float *ms);
/* A9  compute_dz (gato_schur.cuh:1012-1022) with d_Ginv_dense = inverses from gato_form_schur. */
int gato_compute_dz(gato_solver *s, const void *d_Ginv_dense, const void *d_C_dense, const void *d_g,
                    const void *d_lambda, void *d_dz, void *stream);

/* ---- A13  gato_linsys (gpu_library.cu:25-83) on DEVICE CSR inputs: convert, Schur, stair,
 * PCG, dz, all on `stream`, into the solver's workspace; d_lambda/d_dz may be NULL (results stay
 * in the workspace, gato_solver_buffer 6/7).  Asynchronous. */
int gato_linsys_device(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val,
                       const int *d_C_row, const int *d_C_col, const void *d_C_val,
                       const void *d_g, const void *d_c, double exit_tol, int max_iters, double rho,
                       void *d_lambda, void *d_dz, void *stream);

/* Batched gato_linsys_device (solver from gato_solver_create_batched): d_G_val[B][nnz_G], d_C_val[B][nnz_C],
 * d_g[B][N], d_c[B][S*K] -> d_lambda[B][S*K], d_dz[B][N], d_iters[B]. */
int gato_linsys_device_batched(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val, int nnz_G,
                               const int *d_C_row, const int *d_C_col, const void *d_C_val, int nnz_C,
                               const void *d_g, const void *d_c, double exit_tol, int max_iters, double rho,
                               void *d_lambda, void *d_dz, int *d_iters, void *stream);

/* ---- L4  main_call (gpu_library.cu:85-234) on HOST pointers: H2D of the CSR, `testiters`
 * timed repeats of the whole solve, D2H of lambda and dz.  ms_out[testiters] (may be NULL)
 * receives the per-repeat time the reference prints (gpu_library.cu:186-198).
 * lambda_in is read only when warm_start != 0 and then ignored by the PCG exactly as the
 * reference does (D5). */
int gato_linsys_solve_f32(const int *G_row, int len_G_row, const int *G_col, const float *G_val, int nnz_G,
                          const int *C_row, int len_C_row, const int *C_col, const float *C_val, int nnz_C,
                          const float *g, int len_g, const float *c, int len_c, const float *lambda_in,
                          int S, int C, int K, int testiters, float exit_tol, int max_iters,
                          int warm_start, float rho, float *lambda_out, float *dz_out,
                          int *iters_out, float *ms_out);
/* The host entries keep the solver and staging buffers of the most recent (S, C, K, dtype) for the next call;
 * gato_release_cache() frees them (optional; e.g. before unloading the library). */
int gato_release_cache(void);
int gato_linsys_solve_f64(const int *G_row, int len_G_row, const int *G_col, const double *G_val, int nnz_G,
                          const int *C_row, int len_C_row, const int *C_col, const double *C_val, int nnz_C,
                          const double *g, int len_g, const double *c, int len_c, const double *lambda_in,
                          int S, int C, int K, int testiters, double exit_tol, int max_iters,
                          int warm_start, double rho, double *lambda_out, double *dz_out,
                          int *iters_out, float *ms_out);

/* ---- knot-sharded PCG across GPUs (NEW work: the reference is single-device, gato_utils.cuh:831,
 * and has no communication library).  One process per GPU; rank r owns block
 * rows [k0,k1) of S / Pinv (contiguous, in rank order) and keeps the full gamma.  Per PCG iteration each
 * rank runs two launches of the streaming kernel on its shard and exchanges one fixed-size RECORD after
 * each:  record = [partial dot | first S-block | last S-block] of the vector just produced (2S+1
 * elements of the solver dtype).  The caller all-gathers the records of all ranks (RCCL
 * all_gather over xGMI; gloo in the CPU tests) and hands the gathered array [nranks][2S+1] to the next
 * call - that one collective carries both the global dot (summed in rank order: deterministic) and the
 * neighbour halos.  Ghost blocks of r and p are advanced locally from the neighbours' upsilon / r~
 * blocks, so there are exactly two collectives per iteration and no host synchronisation.
 *   init    : r = gamma, lambda = 0, r~ = Pinv r                       -> send = record(r.r~ , r~)
 *   phase_a : p = r~ + beta p, upsilon = S p      (needs gathered B records of it-1 and it-2 / init)
 *                                                                      -> send = record(p.upsilon, upsilon)
 *   phase_b : lambda += alpha p, r -= alpha upsilon, r~ = Pinv r  (needs gathered B record of it-1 / init
 *             and the gathered A record of this iteration)              -> send = record(r.r~, r~)
 *   finish  : last exit test; d_lambda_full_out (S*K_total) = own slice of lambda, zero elsewhere
 *             (sum-all-reduce it to assemble lambda); d_iters as gato_pcg.
 * S/Pinv/gamma pointers are FULL-system arrays (block row 0 first); all pointers are device pointers. */
int gato_shard_pcg_init(gato_solver *s, int rank, int nranks, int k0, int k1, const void *d_S, const void *d_Pinv,
                        const void *d_gamma, double exit_tol, int max_iters, void *d_send, void *stream);
int gato_shard_pcg_phase_a(gato_solver *s, int it, const void *d_recvB_cur, const void *d_recvB_prev, void *d_send,
                           void *stream);
int gato_shard_pcg_phase_b(gato_solver *s, int it, const void *d_recvB_cur, const void *d_recvA, void *d_send,
                           void *stream);
int gato_shard_pcg_finish(gato_solver *s, const void *d_recvB_last, void *d_lambda_full_out, int *d_iters,
                          void *stream);
/* Host-side convergence poll between iterations (synchronises `stream`): *done = 1 once the exit test has fired.
 * Every rank computes the same sums, so every rank reads the same value. */
int gato_shard_pcg_done(gato_solver *s, int *done, void *stream);

/* ---- multi-GPU cluster: the persistent PCG launch with a device-initiated cross-GPU hand-off (NEW work, SURVEY.md
 * section 8e; replaces the reference's grid barriers gato_pcg.cuh:363,378,393,428 across GPUs).  One process per
 * GPU.  Rank r owns the balanced contiguous knot range gato_cluster_knot_range gives and a MIRROR (a few KB of
 * fine-grained device memory) that the peers write with system-scope stores over xGMI: per hand-off the rank's total
 * goes into every rank's mirror and the rank's boundary blocks into the neighbours'; a rank polls only its own
 * mirror.  Two hand-offs per iteration, no host involvement, no collective library inside the loop.  While the cluster
 * has at most 256 workgroups in all the exchange is flat (every workgroup's partial straight into every mirror, option
 * cluster_flat = 1, default); larger clusters gather inside each GPU first, then across GPUs.
 *   gato_cluster_create   allocates and zeroes the mirror, returns its 64-byte hipIpcMemHandle_t in ipc_handle_out
 *   gato_cluster_connect  ipc_handles = nranks x 64 bytes in rank order (all-gathered by the caller), and / or
 *                         ptrs[r] = the mirror of a rank living in THIS process (gato_cluster_local_mirror);
 *                         afterwards every rank must pass a host barrier before the first gato_cluster_pcg
 *   gato_cluster_pcg      this rank's part of one solve: full-system S / Pinv / gamma / lambda arrays, of which only
 *                         the rows of the rank's range are read / written; same exit_tol and max_iters on every
 *                         rank; asynchronous; d_iters as gato_pcg (-1: a hand-off timed out, on every rank alike).
 *                         Option pcg_variant = 1 on EVERY rank: the single-reduction recurrence - ONE cross-GPU exchange per
 *                         iteration instead of the reference's two reductions (gato_pcg.cuh:353-394); the ranks then also
 *                         read Pinv on the knots k0-1, k1 and gamma on k0-2..k1+1.  On return lambda also holds the right
 *                         neighbour's first block at row k1 (it arrives inside the launch): what dz of knot k1-1 needs
 *                         (gato_schur.cuh:833-838)
 *   gato_cluster_linsys   this rank's part of a WHOLE solve (gato_linsys, gpu_library.cu:25-83): stage kernels on the knots
 *                         its shard reads, gato_cluster_pcg, dz on its range - nothing crosses the host or a collective in
 *                         between; inputs replicated, d_lambda / d_dz full-length arrays of which the rank's rows are written */
int gato_cluster_knot_range(int K, int rank, int nranks, int *k0, int *k1);
int gato_cluster_create(gato_solver *s, int rank, int nranks, void *ipc_handle_out);
void *gato_cluster_local_mirror(gato_solver *s);
int gato_cluster_connect(gato_solver *s, const void *ipc_handles, void *const *ptrs);
/* Workgroups x threads this rank's launch would use; 0 x 0: its knots do not fit one persistent launch on this GPU
 * (the caller then takes the gato_shard_pcg_* schedule on every rank). */
int gato_cluster_fits(gato_solver *s, int *groups, int *threads);
int gato_cluster_pcg(gato_solver *s, const void *d_S, const void *d_Pinv, const void *d_gamma, void *d_lambda,
                     double exit_tol, int max_iters, int *d_iters, void *stream);
int gato_cluster_linsys(gato_solver *s, const int *d_G_row, const int *d_G_col, const void *d_G_val, const int *d_C_row,
                        const int *d_C_col, const void *d_C_val, const void *d_g, const void *d_c, double exit_tol,
                        int max_iters, double rho, void *d_lambda, void *d_dz, int *d_iters, void *stream);
int gato_cluster_destroy(gato_solver *s);
/* The hand-off epochs of a cluster only grow (32 bits; a launch takes 2 max_iters + 8 on every rank alike: about ten million
 * 200-iteration solves).  gato_cluster_launches_left: how many launches with this max_iters still fit (the same number on every
 * rank).  When it reaches 0 the caller renews the epoch space on EVERY rank at the same solve: wait for the rank's own launches,
 * host barrier over the ranks, gato_cluster_rewind (zeroes the mirror and the level-1 slots, counters back to 0), second barrier.
 * gato_cluster_pcg / gato_cluster_linsys refuse a launch that does not fit (GATO_EINVAL). */
int gato_cluster_launches_left(gato_solver *s, int max_iters, long long *left);
int gato_cluster_rewind(gato_solver *s);

/* ---- direct block input (new): the caller already holds the per-knot blocks in the
 * reference's dense layouts - d_G_blocks as G_dense WITHOUT rho, d_C_blocks as C_dense - so the CSR scatter is
 * skipped; rho is added to the diagonals of Q_k, R_k on the way into the solver's workspace.  Otherwise identical
 * to gato_linsys_device (batched solvers take B systems back to back). */
int gato_linsys_device_blocks(gato_solver *s, const void *d_G_blocks, const void *d_C_blocks, const void *d_g,
                              const void *d_c, double exit_tol, int max_iters, double rho, void *d_lambda, void *d_dz,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GATO_HIP_H */

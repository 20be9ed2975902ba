/*
 * picasso_hip.h — C ABI of libpicasso_hip.so, the MI355X (gfx950) backend for
 * Picasso's localization hot path.
 *
 * Conventions follow the one native binding the reference already has
 * (picasso/ext/pygpufit/gpufit.py:40-76,338-366, Gpufit's C interface):
 *   - the caller owns every buffer; arrays are C-contiguous;
 *   - every call returns an int status, 0 = ok; pmi_last_error() gives the text;
 *   - no torch / numpy types cross this boundary, only pointers and sizes.
 *
 * Two families of entry points:
 *   pmi_<op>        host buffers in, host buffers out (what a ctypes/cffi/cgo
 *                   binding of the reference would call; does H2D/D2H inside);
 *   pmi_<op>_dev    device pointers + a HIP stream (void* = hipStream_t, NULL =
 *                   default stream); asynchronous; used to keep a movie
 *                   resident in HBM and to chain identify -> fit without a
 *                   host round trip.
 *
 * Reference interfaces replaced (paths relative to jungmannlab/picasso v0.10.3):
 *   pmi_identify*      picasso/localize.py:639-749 identify (-> :247-292
 *                      identify_in_image, :97-134 _local_maxima, :202-244
 *                      _net_gradient, :295-337 ROI crop, :395-401 frame bounds)
 *   pmi_net_gradient   picasso/localize.py:202-244 _net_gradient (+ :153-181
 *                      _gradient_at) on one frame with the caller's unit vectors
 *   pmi_get_spots*     picasso/localize.py:1115-1145 get_spots (-> :917-931
 *                      _cut_spots_numba, :1101-1112 _to_photons)
 *   pmi_gaussmle*      picasso/gaussmle.py:409-475 gaussmle / :478-530
 *                      gaussmle_async (-> :533-742 sigma, :745-954 sigmaxy)
 *   pmi_locs_from_fits_dev  picasso/gaussmle.py:957-1037 locs_from_fits
 *   pmi_zfit*          picasso/zfit.py:327-382 _fit_z (per-localization loop)
 *   pmi_avgroi*        picasso/avgroi.py:45-65 fit_spots
 *   pmi_gausslq*       picasso/gausslq.py:206-300 fit_spot / fit_spots /
 *                      fit_spots_parallel (scipy.optimize.leastsq = MINPACK lmdif
 *                      on the residuals of :151-203, start values of :95-112)
 *   pmi_locs_from_fits_lq_dev  picasso/gausslq.py:404-484 locs_from_fits
 *                      (+ :547-589 localization_precision)
 *   pmi_localize_lq_dev     picasso/localize.py:1682-1815 localize with
 *                      fitting_method="gausslq"
 *   pmi_render_*       picasso/render.py:37-175 render -> :798-853 _render_hist,
 *                      :1020-1070 _render_gaussian (ang=None), :177-232, :451-467, :494-575
 *   pmi_xcorr, pmi_rcc_pairs   picasso/imageprocess.py:27-50 xcorr, :53-161
 *                      get_image_shift (up to its curve_fit), :164-217 rcc
 *   pmi_localize_mle_dev    picasso/localize.py:1682-1815 localize with
 *                      fitting_method="gaussmle" (identify -> get_spots -> fit
 *                      -> table) as one asynchronous device pipeline
 */
#ifndef PICASSO_HIP_H
#define PICASSO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMI_OK            0
#define PMI_ERR_CAPACITY  1  /* output capacity too small; *out_n = rows needed */
#define PMI_ERR_ARG      -1
#define PMI_ERR_HIP      -2
#define PMI_ERR_NODEVICE -3

/* movie pixel types (the reference casts every frame to float32,
 * picasso/localize.py:332; integers up to 24 bits are exact) */
enum pmi_dtype { PMI_U16 = 0, PMI_U8 = 1, PMI_I16 = 2, PMI_U32 = 3, PMI_I32 = 4, PMI_F32 = 5 };
/* picasso/gaussmle.py:413 method: "sigma" | "sigmaxy" */
enum pmi_mle_method { PMI_MLE_SIGMA = 0, PMI_MLE_SIGMAXY = 1 };

#define PMI_MAX_BOX 21   /* odd box sizes 3..21 */

/* ---- library / device ------------------------------------------------ */
int         pmi_version(void);
const char *pmi_last_error(void);            /* gpufit_get_last_error analogue  */
int         pmi_device_count(void);          /* gpufit_cuda_available analogue: 0 = no GPU */
/* The device of the CALLING THREAD (HIP keeps it per thread).  Everything the library keeps on a device — scratch banks,
 * unit-vector tables, FFT plans, side streams, the statistics of the last fit — is keyed by the device current in the thread
 * that calls in, so one process may drive several GPUs from one host thread each (at most 16 devices).  A device that does
 * not exist is refused (PMI_ERR_ARG) and the thread stays where it was.                                       */
int         pmi_set_device(int device);
int         pmi_get_device(int *device);     /* the calling thread's current device */
int         pmi_device_info(char *name, size_t name_len, int *compute_units,
                            size_t *total_mem_bytes);

/* ---- device memory, for hosts without a HIP binding ------------------- */
int pmi_malloc(void **dptr, size_t bytes);
int pmi_free(void *dptr);
int pmi_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int pmi_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int pmi_stream_synchronize(void *stream);
/* A non-blocking HIP stream of the library's own (for callers without torch): the *_dev entry points queued on
 * it do not order against default-stream copies, so the upload of the next frame chunk (pmi_memcpy_h2d from
 * another host thread) overlaps them.  pmi_memcpy_d2h_async queues a copy on a stream; the host buffer is valid
 * after pmi_stream_synchronize.                                                                              */
int pmi_stream_create(void **stream);
int pmi_stream_destroy(void *stream);
int pmi_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int pmi_release_scratch(void);               /* frees the library's cached scratch buffers */
/* Scratch of the calls that follow ON THE CALLING THREAD comes from bank 0 (default) or 1: a caller that keeps two
 * pipelines in flight on two streams (the fit of one frame range beside the scan of the next) selects a bank before
 * queueing each; two host threads that drive one stream each select a bank each, once.  Calls that share a bank must
 * not overlap in time (one thread at a time per bank). */
int pmi_scratch_bank(int bank);

/* ---- identify --------------------------------------------------------- *
 * movie: (F, Y, X) pixels of `dtype`.  roi4 = {y0, x0, y1, x1} already
 * normalised to the frame (numpy slice semantics), or NULL.  Frames outside
 * [f_lo, f_hi] (inclusive) are skipped.  A pixel is reported when it is the
 * first maximum of its box x box window and its net gradient is > min_ng.
 * Output rows are ordered by (frame, y, x); coordinates are frame coordinates.
 * If more than `cap` rows exist, returns PMI_ERR_CAPACITY and *out_n = needed.
 * Every pixel type compares as float32, as in the reference (picasso/localize.py:332).  uint16 / int16 / uint8 movies take the
 * packed scan; float32, int32 and uint32 movies — whatever they hold: counts, fractions, negatives, values beyond 16 bits,
 * NaN, +-inf — are scanned in one pass on 16-bit keys (the upper half of the order-preserving integer image of the float32
 * the reference would see; 32-bit integers are converted as its cast does), with the first-argmax rule, the net gradient and
 * the threshold decided on those float32 values; 32-bit integer movies of 16-bit counts on frames of at most 256 columns are
 * narrowed to uint16 chunk by chunk instead; the generic kernel serves boxes 19 / 21 and crops narrower than a stencil.
 * Same table whichever kernel runs. */
int pmi_identify(const void *movie, int dtype, int64_t F, int64_t Y, int64_t X,
                 int box, double min_ng, const int64_t *roi4, int64_t f_lo, int64_t f_hi,
                 int32_t *out_frame, int32_t *out_y, int32_t *out_x, float *out_ng,
                 int64_t cap, int64_t *out_n);

/* 32-bit integer movies of 16-bit counts on narrow frames (above) pass through a uint16 copy, `frames` frames at a time
 * (0 = default: as many as fit 1 GiB).  A memory knob; the table does not depend on it. */
int pmi_identify_set_narrow_chunk(int64_t frames);

/* Device form.  d_out_n is a device int64 receiving the row count (rows beyond
 * cap are counted but not written).  Nothing is synchronised. */
int pmi_identify_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                     int box, double min_ng, const int64_t *roi4, int64_t f_lo, int64_t f_hi,
                     int32_t *d_frame, int32_t *d_y, int32_t *d_x, float *d_ng,
                     int64_t cap, int64_t *d_out_n, void *stream);

/* ---- get_spots -------------------------------------------------------- *
 * spots[i] = float32(movie[frame, y-r:y+r+1, x-r:x+r+1]); then
 * (s - baseline) * sensitivity / gain in float32, in that order.            */
int pmi_get_spots(const void *movie, int dtype, int64_t F, int64_t Y, int64_t X,
                  const int32_t *frame, const int32_t *y, const int32_t *x, int64_t N,
                  int box, double baseline, double sensitivity, double gain,
                  float *out_spots);
/* d_n: optional device count (rows = min(*d_n, N)); NULL = N rows. */
int pmi_get_spots_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                      const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x,
                      int64_t N, const int64_t *d_n, int box, double baseline,
                      double sensitivity, double gain, float *d_spots, void *stream);

/* ---- gaussmle --------------------------------------------------------- *
 * spots: (N, box, box) float32 photons.  Outputs as gaussmle.py:455-459
 * allocates them: thetas (N,6) = x, y, photons, bg, sx, sy in box-origin
 * coordinates; crlbs (N,6); loglik (N); iterations (N) int32.              */
int pmi_gaussmle(const float *spots, int64_t N, int box, double eps, int max_it, int method,
                 float *thetas, float *crlbs, float *loglik, int32_t *iterations);
int pmi_gaussmle_dev(const float *d_spots, int64_t N, const int64_t *d_n, int box,
                     double eps, int max_it, int method, float *d_thetas, float *d_crlbs,
                     float *d_loglik, int32_t *d_iterations, void *stream);
/* Fused ROI extraction + photon conversion + fit straight from the movie
 * (no (N,box,box) round trip through HBM). */
int pmi_gaussmle_movie_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                           const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x,
                           int64_t N, const int64_t *d_n, int box, double baseline,
                           double sensitivity, double gain, double eps, int max_it, int method,
                           float *d_thetas, float *d_crlbs, float *d_loglik,
                           int32_t *d_iterations, void *stream);

/* How the Newton loop of pmi_gaussmle* runs.  The reference (picasso/gaussmle.py:745-857 under numba) keeps theta
 * and its accumulators in float32 arrays and evaluates every per-pixel intermediate in float64; its convergence
 * test |delta| < eps (:844-852, :632-638) is a discrete decision.
 *   PMI_MLE_FAST    the float32 loop only: ~1e-5 px from the reference, but a step that lands within rounding
 *                   distance of eps can end the fit an iteration earlier or later than the reference does;
 *   PMI_MLE_REFIT   (default) float32 loop, and every spot on which the two arithmetics can part — its largest tested
 *                   step came within `margin` (relative) of eps in some iteration; a curvature term was not negative;
 *                   a width fell below 0.5 px; a parameter swung back and forth without its steps shrinking (the
 *                   iteration does not contract); a pixel lay far off the model (|data / model - 1| or
 *                   |data / model^2| above 16); the fit took more than 64 iterations; or, seen from the Fisher matrix at the
 *                   fitted theta, the per-parameter update does not contract there (lambda_max of the normalised
 *                   Fisher matrix above 1.9: a rounding difference would be multiplied by 1 - lambda_max per
 *                   iteration) — is fitted again from its initial
 *                   parameters in the reference's own arithmetic (float64 intermediates, float32 stores, the
 *                   reference's summation order) inside the same call;
 *   PMI_MLE_STRICT  every spot in the reference's arithmetic.
 * Process-wide; the environment variable PMI_MLE_MODE = fast | refit | strict overrides the mode.
 * pmi_mle_last_refit_count: spots the calling thread's last pmi_gaussmle*_dev / pmi_localize_mle_dev call on `stream`
 * fitted again (synchronises the stream); pmi_mle_last_flag_reasons: of those, how many each criterion flagged, in the
 * order margin, curvature, narrow width, swing, far-off pixel, slow, unstable (n <= 7 counters; a spot can carry
 * several, and one that carries `unstable` beside another was fitted again twice, to the same result).        */
enum pmi_mle_mode { PMI_MLE_FAST = 0, PMI_MLE_REFIT = 1, PMI_MLE_STRICT = 2 };
int pmi_mle_set_mode(int mode, double margin);
int pmi_mle_get_mode(int *mode, double *margin);
/* math.erf / math.exp of the reference (gaussmle.py:279, 295, 313, 357) are the C library's under numba, and faithful, not
 * correctly rounded: which last bit they return depends on the libm.  A fit that contracts mostly forgets such a bit when
 * it rounds to float32; a fit that does not — 3x3 boxes whose width collapses, fits that wander for a thousand iterations —
 * carries it into another trajectory.  The reference-arithmetic kernel (the re-fit of PMI_MLE_REFIT, every spot of
 * PMI_MLE_STRICT) evaluates the two functions
 *   PMI_LIBM_GLIBC   operation for operation as glibc >= 2.28 on x86-64 with FMA does (the libm of the machines the
 *                    reference runs on, and of the oracle's) — csrc/libm_glibc.h; the kernel takes 12 - 18 % longer;
 *   PMI_LIBM_DEVICE  with the device library's functions (the behaviour up to round 5; on 1e6 ordinary 7x7 fits no row
 *                    differs);
 *   PMI_LIBM_AUTO    (default) glibc's bits wherever a result was ever seen to hang on them: every spot of
 *                    PMI_MLE_STRICT at any box, and the re-fit of PMI_MLE_REFIT on boxes up to 5x5; the device
 *                    library's in the re-fit of larger boxes, where the default mode's contract (the reference's
 *                    iteration count on every row, 1e-3 px wherever it converged) has not depended on them in 1e8
 *                    fuzzed fits and the list's latency is part of the timed step (2 % of config 2's).
 * Process-wide; the environment variable PMI_MLE_LIBM = auto | glibc | device overrides it.                       */
enum pmi_libm { PMI_LIBM_DEVICE = 0, PMI_LIBM_GLIBC = 1, PMI_LIBM_AUTO = 2 };
int pmi_mle_set_libm(int which);
int pmi_mle_get_libm(int *which);
/* Diagnostic: d_out[i] = f(d_x[i]) for n float64 device values — fn 0 / 1: exp / erf as the kernel evaluates them under
 * PMI_LIBM_GLIBC (the test compares them with the host's C library, bit for bit); fn 2 / 3: the device library's.     */
int pmi_libm_eval_dev(int fn, const double *d_x, int64_t n, double *d_out, void *stream);
int pmi_mle_last_refit_count(int64_t *n_refit, void *stream);
int pmi_mle_last_flag_reasons(int64_t *counts, int n, void *stream);

/* ---- locs_from_fits (gaussmle.py:957-1037) ---------------------------- *
 * Builds the 17-column localization table as structure-of-arrays, row i from
 * identification i (rows stay in identification order = frame order).
 * d_cols: 17 device pointers in this order, each N elements of 4 bytes:
 *  0 frame(u32) 1 x 2 y 3 photons 4 sx 5 sy 6 bg 7 lpx 8 lpy 9 ellipticity
 * 10 net_gradient 11 log_likelihood 12 iterations(u32) 13 photons_unc
 * 14 bg_unc 15 sx_unc 16 sy_unc   (all float32 unless noted)               */
#define PMI_LOC_COLUMNS 17
int pmi_locs_from_fits_dev(const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x,
                           const float *d_ng, const float *d_thetas, const float *d_crlbs,
                           const float *d_loglik, const int32_t *d_iterations, int64_t N,
                           const int64_t *d_n, int box, void *const *d_cols, void *stream);

/* ---- whole path on a resident movie ----------------------------------- *
 * identify -> fused cut+fit -> table, one asynchronous submission.  d_table is
 * one device block of PMI_LOC_COLUMNS * cap * 4 bytes (column c starts at
 * element c*cap).  d_out_n: device int64 row count.  When it comes back
 * larger than cap the table is untouched: resubmit with cap >= *d_out_n.
 * Schedule: a large frame range is cut in two and the scan of the second half runs beside the fit of the first, on
 * a stream of the library's that joins the caller's stream at the start of the call and is joined again before the
 * table is written (the scan is bound by memory requests, the fit by VALU issue); the table is the same, bit for bit.
 * pmi_localize_set_ranges(1) keeps a call on the caller's stream alone (2 = default).                  */
int pmi_localize_set_ranges(int ranges);
/* Pixel hand-off (uint16 movies, boxes up to 15; default off): the scan's exact stage, which holds a candidate's
 * neighbourhood in registers, also leaves the box rows in a compact buffer and the fit's start-value kernel reads those
 * (one or two cache lines per spot) instead of `box` lines of the movie.  Measured on config 2: fit -0.11 ms, scan
 * +0.14 ms (7 more scattered 16-byte stores per candidate on a kernel bound by its memory-side requests) - the same
 * table, no gain, hence off. */
int pmi_localize_set_handoff(int on);
/* Deferred exact stage (default on): on uint16 / uint8 / int16 movies, boxes up to 7 and a positive threshold the packed
 * scan of pmi_localize_mle_dev only emits CANDIDATES (window maximum, floor, neighbour rule) and the start-value kernel
 * of the fit — which reads a candidate's rows anyway — evaluates the float32 net gradient in the reference's (k, l)
 * order, the first-argmax rule and the threshold (picasso/localize.py:97-134, 202-244, 288): the scan no longer re-reads
 * nine lines per candidate.  Same table, bit for bit.  The identification / fit scratch then holds cap + cap / 2 + 4096
 * candidates; if a call finds more, *d_out_n reports the number of CANDIDATES (an upper bound of the rows needed), nothing is
 * fitted and the table is untouched, exactly as for a table that is too small.  0: the exact stage stays in the scan. */
int pmi_localize_set_defer(int on);
int pmi_localize_mle_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                         int box, double min_ng, const int64_t *roi4, int64_t f_lo, int64_t f_hi,
                         double baseline, double sensitivity, double gain,
                         double eps, int max_it, int method,
                         void *d_table, int64_t cap, int64_t *d_out_n, void *stream);

/* ---- net gradient at given pixels (picasso/localize.py:202-244 _net_gradient) ---- *
 * image: one (Y, X) float32 frame; y, x: n pixel positions; uy, ux: (box, box)
 * float32 unit-vector tables (the centre entry is not read).  out_ng[i] = sum over
 * the box x box window around (y, x), centre excluded, of gy*uy + gx*ux with
 * central differences, accumulated in float32 in the reference's order.  Index -1
 * wraps to the last row / column as the reference's unchecked indexing does;
 * positions whose window reaches past the far edge are refused.                   */
int pmi_net_gradient(const float *image, int64_t Y, int64_t X, const int32_t *y, const int32_t *x,
                     int64_t n, int box, const float *uy, const float *ux, float *out_ng);

/* ---- gausslq (picasso/gausslq.py:206-300) ------------------------------- *
 * spots: (N, box, box) float32 photons, box odd in [3, 21].  thetas (N,6) =
 * x, y, photons, bg, sx, sy with x, y relative to the box CENTRE (the least-
 * squares model is point-sampled on the grid -r..r, gausslq.py:228).  info /
 * nfev (N, int32, may be NULL) are MINPACK's termination code and the number of
 * residual evaluations, what leastsq(full_output=1) would report.
 * The _dev forms never wait for their stream (round 4): per batch of 2 Mi spots they queue the start values, five rounds
 * of (Jacobian + QR, step) — more for boxes above 7x7 —, one kernel that finishes on the device whatever fit is still
 * running, and the second pass of the mode over a device-side list; every count stays on the device.
 * pmi_localize_lq_dev keeps two frame ranges in flight like pmi_localize_mle_dev (pmi_localize_set_ranges).
 * Arithmetic (modes below): MINPACK adds the box^2 residual rows of a column norm, a Householder product or Q^T f one
 * after the other; tree reductions over the lanes of a spot's group differ from that in the last bits of float64, which
 * matters where one of lmdif's tests (the gain ratio against 1e-4 / 0.25 / 0.75, the termination tests, lmpar's 10 %
 * band, qrfac's pivot choice) is decided within those bits.  REFIT fits every such spot AGAIN from its start values with
 * the sums in MINPACK's order, inside the same call (pmi_gausslq_last_refit_count: how many spots of the calling thread's
 * last call were); STRICT, the default, uses MINPACK's order for every spot from the start.                    */
int pmi_gausslq(const float *spots, int64_t N, int box, float *thetas, int32_t *info, int32_t *nfev);
/* How those sums run (the counterpart of pmi_mle_set_mode for scipy.optimize.leastsq, picasso/gausslq.py:240-242):
 *   PMI_LQ_FAST    tree sums only: theta within ~1e-3 px on all but ~2e-5 of adversarial spots, no second fit;
 *   PMI_LQ_REFIT   tree sums, and the spots with a decision inside rounding distance of its threshold, a pivot tie or a
 *                  nearly rank-deficient Jacobian fitted again in MINPACK's order.  theta, info and nfev are lmdif's on
 *                  99.998 % of adversarial spots; 5e-6 of them end beyond 1e-3 px (up to 2.4 px measured) with the same
 *                  `info` — a float32 rounding of the stored model flipped by the last bits of a tree sum, which no
 *                  test of the fit sees;
 *   PMI_LQ_STRICT  (default) every sum over the residual rows — enorm, qrfac's Householder products, Q^T fvec — in
 *                  MINPACK's sequential order from the first Jacobian on (one chain per column, side by side in the lanes
 *                  of a spot's group): theta, info and nfev are lmdif's on EVERY spot, bit for bit.  Since round 5
 *                  (the image columns on the lanes, csrc/gausslq_w.hip) also the faster mode up to 9x9 boxes (7x7: 0.8x
 *                  the time of REFIT); 1.1x REFIT's time at 11x11 and 13x13, 1.8x at 15x15, 2.4x at 21x21.
 * Process-wide; the environment variable PMI_LQ_MODE = fast | refit | strict overrides the mode.                 */
enum pmi_lq_mode { PMI_LQ_FAST = 0, PMI_LQ_REFIT = 1, PMI_LQ_STRICT = 2 };
int pmi_gausslq_set_mode(int mode);
int pmi_gausslq_get_mode(int *mode);
int pmi_gausslq_last_refit_count(int64_t *n_refit);
/* ... and which test sent them there (n <= 10 counters: pivot choice, lmpar's band, 0.1 fnorm1 < fnorm, the gain-ratio
 * thresholds, the ftol tests, a reduction at the noise level, the xtol test, [7] strict mode: a float32 rounding of the
 * model within a few float64 ulps of a tie / a Jacobian outside the range of the short divisions; a spot can carry
 * several; [8], [9]: rounds queued by the first pass, second passes run)                                     */
int pmi_gausslq_last_tie_reasons(int64_t *counts, int n);
int pmi_gausslq_dev(const float *d_spots, int64_t N, const int64_t *d_n, int box, float *d_thetas,
                    int32_t *d_info, int32_t *d_nfev, void *stream);
int pmi_gausslq_movie_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                          const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x,
                          int64_t N, const int64_t *d_n, int box, double baseline,
                          double sensitivity, double gain, float *d_thetas, int32_t *d_info,
                          int32_t *d_nfev, void *stream);
/* 11-column table of gausslq.locs_from_fits: 0 frame(u32) 1 x 2 y 3 photons
 * 4 sx 5 sy 6 bg 7 lpx 8 lpy 9 ellipticity 10 net_gradient.  em != 0 doubles
 * the variance of lpx/lpy (EMCCD excess noise, gausslq.py:584-585).           */
#define PMI_LQ_COLUMNS 11
int pmi_locs_from_fits_lq_dev(const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x,
                              const float *d_ng, const float *d_thetas, int64_t N, const int64_t *d_n,
                              int em, void *const *d_cols, void *stream);
/* identify -> fused cut + least-squares fit -> table; d_table holds
 * PMI_LQ_COLUMNS * cap * 4 bytes, column c at element c*cap.                  */
int pmi_localize_lq_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                        int box, double min_ng, const int64_t *roi4, int64_t f_lo, int64_t f_hi,
                        double baseline, double sensitivity, double gain, int em,
                        void *d_table, int64_t cap, int64_t *d_out_n, void *stream);

/* ---- zfit (picasso/zfit.py:254-291, 327-382) ---------------------------- *
 * Per localization: argmin over z in [-1000, 1000] of
 * (sqrt(sx) - sqrt(wx(z)))^2 + (sqrt(sy) - sqrt(wy(z)))^2, wx/wy degree-6
 * polynomials (cx7/cy7, highest power first), by the bounded Brent minimiser of
 * scipy.optimize.minimize_scalar (xatol 1e-5, maxiter 500).  Outputs are float64:
 * z before the magnification factor and the squared residual (d_zcalib^2).   */
int pmi_zfit(const float *sx, const float *sy, int64_t N, const double *cx7, const double *cy7,
             double *z, double *sq_residual);
int pmi_zfit_dev(const float *d_sx, const float *d_sy, int64_t N, const int64_t *d_n, const double *cx7,
                 const double *cy7, double *d_z, double *d_sq_residual, void *stream);

/* ---- avg (picasso/avgroi.py:24-65) -------------------------------------- *
 * theta (N,6) = [0, 0, sum, sum, 1, 1] with a float64 ROI sum.               */
int pmi_avgroi(const float *spots, int64_t N, int box, float *theta);
int pmi_avgroi_dev(const float *d_spots, int64_t N, const int64_t *d_n, int box, float *d_theta, void *stream);

/* ---- render (picasso/render.py:37-175 render with blur_method None / "gaussian") --- *
 * x, y (and lpx, lpy) are the float32 columns of the localization table in
 * camera pixels.  The viewport (y_min, x_min)-(y_max, x_max) and oversampling
 * define an image of ny x nx = ceil(oversampling * extent) float32 pixels
 * (pmi_render_dims; render.py:177-232); localizations strictly inside the
 * viewport are drawn, *n_rendered receives their number.
 *   hist:     image[int(y'), int(x')] += 1                       (render.py:451-467)
 *   gaussian: separable Gaussian over +-3 sigma, sigma = oversampling *
 *             max(lp, min_blur_width), in TABLE ORDER per pixel   (render.py:494-575);
 *             iso != 0: both widths = their mean ("gaussian_iso", :1148-1216)
 * The image buffer is overwritten (zeroed first).  The Gaussian _dev form
 * synchronises the stream once: the number of (tile, localization) pairs sizes
 * its sort buffers.                                                          */
int pmi_render_dims(double oversampling, double y_min, double x_min, double y_max, double x_max,
                    int64_t *ny, int64_t *nx);
int pmi_render_hist(const float *x, const float *y, int64_t N, double oversampling,
                    double y_min, double x_min, double y_max, double x_max,
                    float *image, int64_t ny, int64_t nx, int64_t *n_rendered);
int pmi_render_hist_dev(const float *d_x, const float *d_y, int64_t N, double oversampling,
                        double y_min, double x_min, double y_max, double x_max,
                        float *d_image, int64_t ny, int64_t nx, int64_t *d_n_rendered, void *stream);
int pmi_render_gaussian(const float *x, const float *y, const float *lpx, const float *lpy, int64_t N,
                        double oversampling, double y_min, double x_min, double y_max, double x_max,
                        double min_blur_width, int iso, float *image, int64_t ny, int64_t nx, int64_t *n_rendered);
int pmi_render_gaussian_dev(const float *d_x, const float *d_y, const float *d_lpx, const float *d_lpy,
                            int64_t N, double oversampling, double y_min, double x_min, double y_max,
                            double x_max, double min_blur_width, int iso, float *d_image, int64_t ny,
                            int64_t nx, int64_t *d_n_rendered, void *stream);

/* ---- cross-correlation for RCC undrift (picasso/imageprocess.py:27-217) ---- *
 * pmi_xcorr: out = fftshift(real(ifft2(fft2(A) * conj(fft2(B))))) / sqrt(Y*X),
 * float64 images of Y x X (imageprocess.py:27-50).
 * pmi_rcc_pairs: for every pair i < j of n_seg float64 images (pair index in
 * the order of imageprocess.py:197-205), everything get_image_shift (:53-161)
 * does before its curve_fit: correlation, centre crop to `roi` (0 = none;
 * crop_yx = {Y_, X_} rows/columns removed on each side), first maximum in
 * row-major order (peak_yx[2p], peak_yx[2p+1], cropped coordinates) and the
 * box x box window around it (fit_rois[p*box*box ..]).  valid[p] = 1 when the
 * window lies inside the cropped correlation, 0 when numpy's slicing would
 * truncate it (the reference then reports a zero shift), -1 when one of the
 * two images is empty (zero shift by imageprocess.py:85-86).                  */
int pmi_xcorr(const double *image_a, const double *image_b, int64_t Y, int64_t X, double *out);
int pmi_rcc_pairs(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                  int32_t *peak_yx, int32_t *valid, double *fit_rois, int32_t *crop_yx);
/* The same for an explicit list of n_pairs (i, j) index pairs — the share of one rank when the
 * n(n-1)/2 correlations of RCC are split over several GPUs.      */
int pmi_rcc_pair_list(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                      const int32_t *pairs, int64_t n_pairs, int32_t *peak_yx, int32_t *valid,
                      double *fit_rois, int32_t *crop_yx);

/* ---- the sharded path: every rank's localization table on every GPU ---- *
 * The reference has no distributed code; its workers split the movie frame by frame inside one process
 * (picasso/localize.py:438-454).  Here each GPU (one process per GPU) localizes a contiguous frame range and the
 * tables are all-gathered with RCCL, called from this library (loaded at the first pmi_comm_* call).
 *   pmi_comm_available  PMI_OK when RCCL and the entry points used here resolve on this rank; creates nothing.  A
 *                       host AGREES on this over its own channel before any rank enters pmi_comm_init: the
 *                       communicator set-up is itself a collective, and a rank that cannot join it leaves the others
 *                       waiting inside ncclCommInitRank;
 *   pmi_comm_unique_id  rank 0 fills a 128-byte id, which the HOST hands to every rank (file, socket, MPI, ...);
 *   pmi_comm_init       every rank, on its own device (pmi_set_device first);
 *   pmi_allgather_locs  d_table: this rank's ncols x cap column-major table of 4-byte cells (PMI_LOC_COLUMNS or
 *                       PMI_LQ_COLUMNS columns, the same cap on every rank), d_n its device row count;
 *                       d_all_tables receives world x ncols x cap cells (rank-major), d_all_counts world counts.
 *                       One grouped submission on `stream`, asynchronous, no host synchronisation;
 *   pmi_compact_gathered_dev  the gathered tables as ONE ncols x table_cap column-major table, rows in rank order
 *                       (= frame order for contiguous frame shards, picasso/gaussmle.py:1036), total in *d_total.  */
int pmi_comm_available(void);
int pmi_comm_unique_id(void *id128);
int pmi_comm_init(const void *id128, int world, int rank, void **comm);
int pmi_comm_info(void *comm, int *world, int *rank);         /* as RCCL reports them (ncclCommCount / ncclCommUserRank) */
int pmi_comm_library_path(char *path, size_t path_len);         /* the librccl the collectives resolved to (dladdr) */
int pmi_comm_destroy(void *comm);
int pmi_allgather_locs(void *comm, const void *d_table, int ncols, int64_t cap, const int64_t *d_n,
                       void *d_all_tables, int64_t *d_all_counts, void *stream);
int pmi_compact_gathered_dev(const void *d_all_tables, const int64_t *d_all_counts, int world, int ncols, int64_t cap,
                             void *d_table, int64_t table_cap, int64_t *d_total, void *stream);

/* ---- sub-pixel correlation peak (picasso/imageprocess.py:121-141) ---------------------------- *
 * The reference fits a * exp(-0.5 ((x - xc)^2 + (y - yc)^2) / s^2) + b to the box x box window around
 * the correlation maximum with scipy.optimize.curve_fit(p0 = [max, 0, 0, 1, min], bounds = ([0, -inf,
 * -inf, 0, 0], inf)) = least_squares(method="trf", jac="2-point"): the Trust Region Reflective algorithm
 * of scipy 1.15.3, restated for one thread per window (csrc/peakfit.hip).
 * pmi_peak_fit: rois (n, box, box) float64 -> popt (n, 5) = a, xc, yc, s, b and scipy's termination
 * status (1 gtol, 2 ftol, 3 xtol, 4 both; 0 max_nfev, where curve_fit raises RuntimeError; -2: window minimum
 * < 0, where curve_fit raises "x0 is infeasible"; -3: a NaN or an infinity in the window, where curve_fit raises
 * ValueError).
 * pmi_rcc_shifts: all of get_image_shift (:53-161) for a list of (i, j) pairs of the n_seg float64
 * images — correlation, centre crop, first maximum, window, fit — -> shift_yx (n_pairs, 2) = (-yc, -xc);
 * fit_status as above, or -1 where the reference returns (0, 0) without fitting (empty image, window
 * truncated by the border).                                                                         */
/* Makes (and caches) the FFT plans of Y x X images ahead of the first correlation: rocFFT compiles a plan's kernels when
 * the plan is made (2.5 s at 2048 x 2048).  Thread-safe; meant for a side thread of the host while it localizes. */
int pmi_fft_prewarm(int64_t Y, int64_t X);
int pmi_peak_fit(const double *rois, int64_t n, int box, double *popt, int32_t *status);
int pmi_rcc_shifts(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                   const int32_t *pairs, int64_t n_pairs, double *shift_yx, int32_t *fit_status);

/* ---- timing hooks for bench.py (HIP events on the given stream) ------- */
int pmi_event_create(void **event);
int pmi_event_record(void *event, void *stream);
int pmi_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on stop */
int pmi_event_destroy(void *event);
/* Milliseconds the last pmi_identify_dev / pmi_gaussmle*_dev spent in its
 * dominant kernel, measured with HIP events around that kernel when enabled. */
/* name of the scan kernel the calling thread's last identify / localize call launched, as rocprofv3 prints it (e.g.
 * "identify_scan_u16_fast_kernel<3, 3, 1, 0, false>", + " defer" when the exact stage may be left to the fit) */
int pmi_last_scan_kernel(char *name, size_t name_len);
int pmi_set_kernel_timing(int enabled);
int pmi_last_kernel_ms(float *scan_ms, float *fit_ms);

#ifdef __cplusplus
}
#endif
#endif /* PICASSO_HIP_H */

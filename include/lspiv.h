/*
 * lspiv.h -- C ABI of liblspiv_hip.so, the MI355X (gfx950) LSPIV cross-correlation engine.
 *
 * This is the drop-in boundary for the hot path behind pyorc's Frames.get_piv().  The
 * reference has no native boundary at all: the seam is the Python-level `engine` string
 * (pyorc/api/frames.py:118,176-177) that is forwarded to pyorc/velocimetry/ffpiv.py and
 * from there to six symbols of the third-party `ffpiv` package.  Every entry point below
 * names the reference call it replaces.  INTEGRATION.md shows the ctypes binding and the
 * five-line patch a pyorc maintainer would add.
 *
 * Conventions
 *   - plain C types only; all arrays are C-contiguous, caller-allocated, never retained;
 *   - frames are (T, H, W) of dtype LSPIV_U8 / LSPIV_F32 / LSPIV_F64; a stack of T frames
 *     holds P = T-1 pairs (pair p = frames p and p+1, labelled with frame p+1's time stamp,
 *     pyorc/velocimetry/ffpiv.py:403);
 *   - window index is row-major k*n_cols+m (pyorc/velocimetry/ffpiv.py:469);
 *   - every function returns LSPIV_OK (0) or a negative status; lspiv_last_error() returns
 *     the thread-local message of the last failure on this thread;
 *   - "_dev" variants take DEVICE pointers (HBM-resident stacks, multi-GPU shards) and a
 *     stream handle (NULL = the library's own per-device stream); host variants stage
 *     through pinned buffers and synchronise before returning.  The library's stream is
 *     NON-BLOCKING (it does not synchronise with the NULL stream): device data handed to a
 *     "_dev" call must already be complete, or the call must be given the stream that
 *     produces it; lspiv_memcpy_h2d / _d2h / lspiv_memset_dev run on the library's stream and
 *     wait for it.  "_dev" calls that need temporaries (normalize) share one scratch buffer
 *     per device: calls that overlap in time must be issued on one stream;
 *   - NaN conventions follow the reference: skipped / invalid windows yield NaN, never an
 *     error (pyorc/velocimetry/ffpiv.py:93-97,465-466).
 */
#ifndef LSPIV_H
#define LSPIV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSPIV_ABI_VERSION 5   /* 5 (round 6): lspiv_chunk_alignment(wy, wx) without a grid now returns the alignment that is right on EVERY grid (75
                               * where it returned 25: callers that cut chunks on it stay bit-reproducible on large grids); additions:
                               * lspiv_upload_frames, lspiv_trace / lspiv_trace_read; the host-pointer projection entry points no longer
                               * share the PIV host entry points' lock and workspaces.
                               * (round 5 added lspiv_chunk_alignment_grid, lspiv_piv_velocity_at, lspiv_ensemble_flag_digest, lspiv_kernel_times + the "time_kernel" option and the test hook lspiv_debug_hold_lock; no version change: nothing existing moved)
                               * 4 (round 4, additions only): lspiv_build_info, the float64 rescue of the ensemble's final fit
                               * (lspiv_ensemble_set_retain / _stats / _flag / _partials / _finish_partials), lspiv_stream_release,
                               * lspiv_stream_create_priority; 3: lspiv_rescue_stats,
                               * lspiv_project_frames_u8[_dev], the rescue / v_sign / norm_clip / std_ddof / round_odd options */

/* status codes (mapped by the Python shim onto the reference's exception types) */
#define LSPIV_OK            0
#define LSPIV_EINVAL       -1  /* bad argument                      -> ValueError    */
#define LSPIV_ESHAPE       -2  /* frame smaller than window, T < 2  -> ValueError    */
#define LSPIV_ENOMEM       -3  /* HBM / pinned allocation failed    -> MemoryError   */
#define LSPIV_EHIP         -4  /* HIP runtime error                 -> RuntimeError  */
#define LSPIV_ENODEV       -5  /* no gfx950 device visible          -> RuntimeError  */
#define LSPIV_EUNSUPPORTED -6  /* window size outside kernel range  -> ValueError    */

/* frame dtypes (pyorc frame stacks are uint8 after normalize/project_cv, float32 after
 * edge_detect/smooth/time_diff, float64 out of project_numpy) */
#define LSPIV_U8  0
#define LSPIV_F32 1
#define LSPIV_F64 2

/* largest interrogation window side the kernels accept (ffpiv.cross_corr itself has no upper bound).  2..64:
 * register-resident kernels; above 64 (any parity, square or not, e.g. 96 x 96 and 128 x 128 for 4K footage): the
 * LDS-resident DFT kernel, as long as 2 wy wx floats fit the 160 KB of a CU -- 128 x 128 does; anything larger up to this
 * limit runs the same transforms on slots of HBM scratch: correct, an order of magnitude slower per sample. */
#define LSPIV_MAX_WINDOW 512

/* ---------------------------------------------------------------- library / device ------- */
int         lspiv_abi_version(void);
const char* lspiv_version(void);                        /* "lspiv-hip <version> (gfx950) src <source hash>" */
/* Provenance of the loaded binary (csrc/Makefile compiles both in): LSPIV_BUILD_KERNEL_HASH = first 16 hex digits of the
 * sha256 over csrc/{piv_fft_impl.h, fft_regs.h, common.h, piv_rescue.hip} (the fused PIV kernels; the committed profile
 * summaries are keyed to it), LSPIV_BUILD_SOURCE_HASH = the same over every .hip, .h and .cpp file of csrc/ (sorted by name) and this
 * header.  pyorc_amd._lib.load() recomputes the second from the tree and refuses a stale binary.  "unknown": not built by
 * csrc/Makefile; "": unknown selector. */
#define LSPIV_BUILD_KERNEL_HASH 0
#define LSPIV_BUILD_SOURCE_HASH 1
const char* lspiv_build_info(int what);
const char* lspiv_last_error(void);
int         lspiv_device_count(int* n);                 /* 0 devices is LSPIV_OK with *n = 0 */
int         lspiv_set_device(int device);               /* per calling thread               */
int         lspiv_get_device(int* device);
int         lspiv_device_name(int device, char* buf, size_t len);
int         lspiv_synchronize(void);                    /* hipDeviceSynchronize             */
/* run-time options: "walk" = 1 the default time-walking kernels (segments anchored every lspiv_chunk_alignment pairs of
 * the absolute pair index: results do not depend on the chunking as long as chunks start on such anchors), 0 per-pair
 * kernels (independent of any chunking, ~23 % slower), n > 1 anchor length n, -1 back to the LSPIV_WALK environment
 * variable.
 * Three engine semantics could not be pinned on a real ffpiv run (ffpiv is absent from /root/reference, SURVEY.md
 * section 8c A5 / A7); each is an option whose default is the oracle's reading, so that matching a real ffpiv is a
 * one-line default flip (environment presets: LSPIV_BORDER_PEAK, LSPIV_SIGNAL_MODE, LSPIV_SIGNAL_POSITIVE):
 *   "border_peak"      arg-max of a correlation plane on the plane border (no 3-point fit possible): 0 u = v = NaN
 *                      (default), 1 the plane centre, i.e. zero displacement (OpenPIV's scalar routine), 2 the integer peak;
 *   "signal_mode"      signal_threshold (pyorc/velocimetry/ffpiv.py:93-97) scores 0 each window PAIR: both windows need
 *                      the fraction (default; CHANGELOG 0.9.5 "any of the 2 interrogation window in a window pair"), 1 each
 *                      window POSITION over all frames of the chunk ("fraction of non-zero pixels in the window stack");
 *   "signal_positive"  the score counts 0 samples != 0 ("non-zero pixels", default), 1 samples > 0 ("above zero"). */
/* Four more readings became switches in round 3 (environment presets LSPIV_V_SIGN, LSPIV_NORM_CLIP, LSPIV_STD_DDOF,
 * LSPIV_ROUND_ODD), so that a run of the real ffpiv (tests/golden/regen_from_ffpiv.py) can only end in "combination X matches":
 *   "v_sign"           0 v = row shift of the correlation peak, positive down the image (default; pyorc/api/plot.py:548,576-583
 *                      flips it for plotting only), 1 negated inside the engine;
 *   "norm_clip"        1 the normalised window is clipped at zero (default, A3), 0 plain (a - mean) / std -- served by the
 *                      block-per-window kernels (kinds 3 / 9 / 10), not by the fused FFT kernels;
 *   "std_ddof"         0 population standard deviation in the window normalisation (default), 1 sample (n - 1);
 *   "round_odd"        ffpiv.window.round_to_even for odd sizes (pyorc/api/frames.py:167): 0 half-even of x / 2 (25 -> 24, 27 -> 28;
 *                      default), 1 up (25 -> 26), 2 down (27 -> 26); host side -- pyorc_amd.window.round_to_even reads it. */
/* Float64 rescue pass (round 3): the reference fits EVERY correlation plane in its engine's precision
 * (pyorc/velocimetry/ffpiv.py:465-471); a float32 plane carries ~1e-7 of noise, which the 3-point log fit amplifies beyond
 * 1e-4 on ill-conditioned peaks (a neighbour that is exactly zero, a flat ridge, a tie for the maximum).  The kernels flag
 * those windows and a second kernel re-evaluates them from the frames in float64, overwriting u and v:
 *   "rescue"           1 (default; environment LSPIV_RESCUE) | 0 float32 results as they are;
 *   "rescue_kappa"     plane noise the flag model assumes, in 1e-9 of the plane maximum (default 500);
 *   "rescue_tau"       relative gap between the two largest samples below which the arg-max counts as ambiguous and the whole
 *                      plane is re-evaluated, in 1e-9 (default 4000).
 * float64 HOST stacks are narrowed to float32 while they are staged (the kernels compute in float32).  A frame that rides on a
 * DC offset far above its contrast would lose the low bits of its texture in that conversion, where the reference normalises
 * every window in float64; the per-window normalisation does not see a constant added to a frame, so:
 *   "narrow_offset"    smallest magnitude of a frame's DC offset (the mean of 4096 strided samples, rounded to an integer: a
 *                      function of the frame alone) that is subtracted while narrowing; default 1024 (never triggers on 8-bit-like
 *                      imagery, whose results stay bit-identical to a float32 copy of the stack), -1 never; environment
 *                      LSPIV_NARROW_OFFSET.  lspiv_piv_pairs / lspiv_ensemble_accumulate only, and only without a signal
 *                      threshold (which counts samples != 0).  float64 stacks already in HBM ("_dev") are converted as they are. */
int         lspiv_set_option(const char* name, int value);
int         lspiv_get_option(const char* name, int* value);
/* counters of the rescue pass on `stream` (NULL: the library's own stream), after synchronising with it: stats[0..4] =
 * "fit" / "amb" windows of the last launch, the same two summed over all launches, windows processed by all launches */
int         lspiv_rescue_stats(void* stream, int64_t* stats);
/* which kernel a window size dispatches to: 1 = FFT 32x32, 2 = FFT 64x64, 6 = FFT 8x8 / 16x16, 8 = prime-factor FFT
 * kernels (every other even square window 6..62), 7 / 4 / 5 = odd square windows (and 4x4) 4..7 / 9..15 / 21..31
 * embedded in the 16- / 32- / 64-point FFT kernels, 3 = direct spatial correlation (non-square and odd 17 / 19 / 33..63
 * windows of fewer than 1500 samples), 9 = LDS-resident 2-D transform (any window with a side above 64, and the non-square /
 * odd ones from 1500 samples on), 10 = the same transforms on HBM scratch (windows that outgrow the LDS of a CU: a side
 * above 128); <0 = unsupported.  Host-only. */
int         lspiv_kernel_kind(int wy, int wx);

/* ---------------------------------------------------------------- window grid (host) ----- */
/* replaces ffpiv.window.get_rect_coordinates (pyorc/api/frames.py:85-90) and the implied
 * ffpiv.window.get_axis_shape: n = (dim - win)//(win - overlap) + 1, centres
 * arange(n)*(win-overlap) + win/2 as integer pixel indices. */
int lspiv_grid_shape(int64_t H, int64_t W, int wy, int wx, int oy, int ox, int64_t* n_rows, int64_t* n_cols);
int lspiv_grid_coords(int64_t H, int64_t W, int wy, int wx, int oy, int ox,
                      int64_t* rows /* n_rows */, int64_t* cols /* n_cols */);

/* ---------------------------------------------------------------- memory planner --------- */
/* replaces ffpiv.window.required_memory / available_memory (pyorc/velocimetry/ffpiv.py:120-129):
 * device bytes one call on T frames needs (frames + results [+ planes]); free/total HBM. */
int64_t lspiv_required_bytes(int64_t T, int64_t H, int64_t W, int dtype, int wy, int wx, int oy, int ox,
                             int with_planes);
int     lspiv_available_bytes(int64_t* free_bytes, int64_t* total_bytes);

/* ---------------------------------------------------------------- the hot path ----------- */
/* One fused call per frame chunk: replaces ffpiv.cross_corr + the corr_max / s2n reductions +
 * ffpiv.u_v_displacement of pyorc/velocimetry/ffpiv.py:446-474 (_get_uv_timestep).
 *   u, v          displacement in PIXELS (u = column shift, v = row shift, no sign flip)
 *   corr_max      nanmax of the clipped correlation plane
 *   s2n           corr_max / nanmean(plane)
 *   each (T-1) * n_rows * n_cols float32.
 *   corr_planes   NULL, or (T-1) * n_win * wy * wx float32: the fft-shifted, clipped planes that
 *                 ffpiv.cross_corr returns (NaN planes for windows below signal_threshold).
 *   signal_threshold < 0 disables the pre-mask (reference: None).                           */
int lspiv_piv_pairs(const void* frames, int dtype, int64_t T, int64_t H, int64_t W,
                    int wy, int wx, int oy, int ox, float signal_threshold,
                    float* u, float* v, float* corr_max, float* s2n, float* corr_planes);

/* same on device-resident data.  d_out is 4 planes [u | v | corr_max | s2n], each
 * (T-1)*n_win float32, contiguous (so one all-gather moves a rank's whole result block).    */
int lspiv_piv_pairs_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W,
                        int wy, int wx, int oy, int ox, float signal_threshold,
                        float* d_out, float* d_corr_planes, void* stream);

/* Chunked callers (the time-chunk loop of pyorc/velocimetry/ffpiv.py:140,399-442; multi-GPU time blocks): the same two
 * calls on a chunk whose first pair has index `pair_offset` in the caller's whole stack.  The reference computes every
 * window independently, so its results cannot depend on the chunking; the default kernels here share transforms between
 * consecutive pairs of a window in runs that start at multiples of lspiv_chunk_alignment() of the ABSOLUTE pair index.
 * Chunks that start on such a multiple therefore reproduce, bit for bit, what one call over the whole stack returns
 * (tested); a chunk that starts elsewhere is still correct but its first (alignment - offset % alignment) pairs may
 * differ from the whole-stack run in the last float32 bit.  lspiv_piv_pairs[_dev] == these with pair_offset 0.
 * lspiv_chunk_alignment: pairs; 1 for window sizes served by per-pair kernels and with option "walk" = 0.  Host-only.
 * Round 5: the run length depends on the window GRID as well -- 25 pairs, or 75 on grids with at least as many windows as the chip
 * has lane groups for that window family (1080p 32 x 32 @ 50 %, 64 x 64 @ 75 %, 4K: less per-segment overhead, csrc/common.h
 * walk_anchor).  lspiv_chunk_alignment_grid(H, W, ...) is the figure to cut chunks on for frames of that shape; negative status for
 * a bad shape.  lspiv_chunk_alignment(wy, wx), without a grid, is the alignment that is right for EVERY frame shape (ABI 5: the
 * longest run length of the family, a multiple of every grid's; ABI 4 returned the run length of small grids). */
int lspiv_chunk_alignment(int wy, int wx);
int lspiv_chunk_alignment_grid(int64_t H, int64_t W, int wy, int wx, int oy, int ox);
int lspiv_piv_pairs_at(const void* frames, int dtype, int64_t T, int64_t H, int64_t W,
                       int wy, int wx, int oy, int ox, float signal_threshold, int64_t pair_offset,
                       float* u, float* v, float* corr_max, float* s2n, float* corr_planes);
int lspiv_piv_pairs_dev_at(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W,
                           int wy, int wx, int oy, int ox, float signal_threshold, int64_t pair_offset,
                           float* d_out, float* d_corr_planes, void* stream);
/* lspiv_piv_pairs_at with the per-chunk scaling of _get_ffpiv_timestep (pyorc/velocimetry/ffpiv.py:418-419) applied on the device
 * before the results cross PCIe (round 5): v_x = (u * res_x / dt[pair]).astype(float32), v_y likewise -- the product in float32 (what
 * numpy computes for a float32 array times a Python float), the division in float64, one rounding; dt: seconds per pair, T - 1
 * doubles.  The Python mirror uses it when the resolutions are Python floats (pyorc: camera_config.resolution) and keeps numpy's own
 * arithmetic for anything else. */
int lspiv_piv_velocity_at(const void* frames, int dtype, int64_t T, int64_t H, int64_t W,
                          int wy, int wx, int oy, int ox, float signal_threshold, int64_t pair_offset,
                          double res_x, double res_y, const double* dt,
                          float* v_x, float* v_y, float* corr_max, float* s2n);

/* Host frames into a slice of an HBM-resident stack, the way lspiv_piv_pairs brings them in -- pinned ring, staging threads,
 * float64 narrowed to float32 with the "narrow_offset" guard (so d_dst receives n_frames * H * W samples of LSPIV_F32 for LSPIV_F64
 * input, of `dtype` otherwise) -- without launching anything: what lets the chunk loop of pyorc/velocimetry/ffpiv.py:399-440 keep a
 * LAZY stack resident in HBM, filled in whatever pieces dask delivers (its own blocks: no frame is decoded twice) and launched on
 * the kernels' anchors (pyorc_amd/resident.py).  signal_threshold: that of the PIV call which will read the frames (it decides the
 * guard exactly as there).  Blocking; waits for the library's stream before the first DMA. */
int lspiv_upload_frames(void* d_dst, const void* frames, int dtype, int64_t n_frames, int64_t H, int64_t W,
                        float signal_threshold);

/* replaces ffpiv.u_v_displacement on an existing plane volume (pyorc/velocimetry/ffpiv.py:324,471):
 * planes (P, n_win, wy, wx) float32 -> u, v (P * n_win) float32 in pixels.                   */
int lspiv_u_v_displacement(const float* corr_planes, int64_t P, int64_t n_win, int wy, int wx,
                           float* u, float* v);

/* ---------------------------------------------------------------- ensemble correlation --- */
/* replaces _get_ffpiv_mean (pyorc/velocimetry/ffpiv.py:182-376): corr_sum / corr_count stay in
 * HBM across chunks; per-pair corr_max / s2n (masked to 0 like ffpiv.py:238-241) are returned per
 * chunk for the time averages of ffpiv.py:284-286.  The handle counts the pairs it has seen: chunks are
 * consecutive, and chunkings whose boundaries are multiples of lspiv_chunk_alignment() give the same bits. */
typedef struct lspiv_ensemble lspiv_ensemble;
int lspiv_ensemble_begin(int64_t H, int64_t W, int wy, int wx, int oy, int ox, lspiv_ensemble** handle);
int lspiv_ensemble_accumulate(lspiv_ensemble* handle, const void* frames, int dtype, int64_t T,
                              float corr_min, float s2n_min, float signal_threshold,
                              float* corr_max /* (T-1)*n_win */, float* s2n /* (T-1)*n_win */);
/* same on a device-resident chunk: d_corr_s2n = [corr_max | s2n], 2*(T-1)*n_win float32 in HBM; asynchronous on
 * `stream` (NULL = the library's stream). */
int lspiv_ensemble_accumulate_dev(lspiv_ensemble* handle, const void* d_frames, int dtype, int64_t T,
                                  float corr_min, float s2n_min, float signal_threshold,
                                  float* d_corr_s2n, void* stream);
/* count filter (count < count_min * n_frames -> NaN), mean plane, sub-pixel peak.
 * u, v (n_win) in pixels; corr_count (n_win) float32; corr_mean NULL or n_win*wy*wx float32.  */
int lspiv_ensemble_finish(lspiv_ensemble* handle, float count_min, float n_frames,
                          float* u, float* v, float* corr_count, float* corr_mean);
/* running state out of / into HBM: corr_sum (n_win*wy*wx, fft-shifted planes) and corr_count (n_win).  Multi-GPU
 * ensemble = every rank accumulates its own time block, the states are summed (one all-reduce) and
 * imported on the rank that calls lspiv_ensemble_finish.  `add` != 0 adds to the current state instead of replacing. */
/* Float64 rescue of the FINAL fit (round 4).  lspiv_ensemble_finish fits the mean plane; where that float32 fit is
 * ill-conditioned (same flag model as the per-pair "rescue" option: a neighbour of the peak near zero, a flat ridge, samples tying
 * for the maximum -- rare, and mostly in ensembles of a few pairs) the samples the fit reads are re-evaluated in float64 from the
 * FRAMES, over the pairs that were added to the sum, and u, v are overwritten.  The frames therefore have to be reachable at
 * finish:
 *   - lspiv_ensemble_accumulate (host frames) keeps its upload buffers in HBM until finish / destroy, as long as they fit
 *     LSPIV_ENSEMBLE_RETAIN_BYTES (default: a quarter of the device memory); beyond that the float32 fits stay;
 *   - lspiv_ensemble_accumulate_dev keeps nothing by default (LSPIV_RETAIN_NONE: float32 fits, as in round 3);
 *     LSPIV_RETAIN_COPY makes the handle copy every chunk (same budget), LSPIV_RETAIN_BORROW only records the caller's pointer --
 *     the caller then guarantees that every chunk handed to accumulate_dev stays valid and unchanged until finish;
 *   - a state brought in with lspiv_ensemble_import holds pairs whose frames this handle never saw: lspiv_ensemble_finish keeps the
 *     float32 fits then (the staged finish below is the multi-GPU way).
 * Set the mode before the first accumulate call.  lspiv_ensemble_stats after finish: stats[0..5] = windows flagged, windows
 * re-evaluated, windows left with their float32 fit (no frames / more than four arg-max candidates / list limit), chunks kept,
 * bytes kept, 1 if every chunk could be kept. */
#define LSPIV_RETAIN_NONE   0
#define LSPIV_RETAIN_COPY   1
#define LSPIV_RETAIN_BORROW 2
int lspiv_ensemble_set_retain(lspiv_ensemble* handle, int mode);
int lspiv_ensemble_stats(lspiv_ensemble* handle, int64_t* stats);
/* The same rescue when the sum is spread over several handles (multi-GPU: every rank accumulated its own time block, the
 * states were all-reduced and imported -- replaced, `add` = 0 -- on every rank, so all hold the same sums).  lspiv_ensemble_finish
 * in three stages:
 *   lspiv_ensemble_flag            mean planes, float32 fits, the flagged windows sorted by index: the same *n_records and the
 *                                  same list on every rank;
 *   lspiv_ensemble_partials        this handle's float64 sums over ITS retained chunks, partials[n_records][LSPIV_ENS_PARTIAL_DOUBLES];
 *                                  *complete = 0 (and zeros) if some chunk of this handle could not be kept -- the ranks then agree
 *                                  (one more all-reduce) to call plain lspiv_ensemble_finish instead;
 *   [the caller sums `partials` over the ranks: one float64 all-reduce]
 *   lspiv_ensemble_finish_partials the fits of the flagged windows from the totals, then the outputs of lspiv_ensemble_finish. */
#define LSPIV_ENS_PARTIAL_DOUBLES 20
int lspiv_ensemble_flag(lspiv_ensemble* handle, float count_min, float n_frames, int64_t* n_records);
/* 64-bit digest (FNV-1a) of the last lspiv_ensemble_flag's sorted records -- window, number of candidates, candidate positions --,
 * 0 without records: ranks compare it (one MAX all-reduce) before they add their partials POSITIONALLY; the same count of
 * flagged windows does not prove the same windows (round 5). */
int lspiv_ensemble_flag_digest(lspiv_ensemble* handle, uint64_t* digest);
int lspiv_ensemble_partials(lspiv_ensemble* handle, double* partials, int* complete);
int lspiv_ensemble_finish_partials(lspiv_ensemble* handle, const double* partials, float* u, float* v, float* corr_count,
                                   float* corr_mean);
int lspiv_ensemble_export(lspiv_ensemble* handle, float* corr_sum, float* corr_count);
int lspiv_ensemble_import(lspiv_ensemble* handle, const float* corr_sum, const float* corr_count, int add);
int lspiv_ensemble_destroy(lspiv_ensemble* handle);

/* ---------------------------------------------------------------- next rows */
/* N1 -- orthoprojection: replaces pyorc.project.img_to_ortho applied per frame by project_numpy
 * (pyorc/project.py:123-230), the numba group average (:19-53) and Frames.project's fillna(0.0)
 * (pyorc/api/frames.py:265).  The index maps are the outputs of CameraConfig.map_idx_img_ortho /
 * map_mean_idx_img_ortho (pyorc/api/cameraconfig.py:739-860):
 *   idx_img[K], idx_ortho[K]   nearest neighbour: out[idx_ortho[k]] = img[idx_img[k]] (flat indices; later k wins)
 *   src_idx[M], norm_idx[M]    group means: sample img[src_idx[i]] belongs to group norm_idx[i] (0..G-1)
 *   uidx[G]                    flat output index of every group; M = 0 disables the mean step (reducer != "mean")
 * Output: (T, dst_h, dst_w) float32, the values the reference returns (widened to float64 there).            */
typedef struct lspiv_projection lspiv_projection;
int lspiv_projection_create(int64_t src_h, int64_t src_w, int64_t dst_h, int64_t dst_w,
                            const int64_t* idx_img, const int64_t* idx_ortho, int64_t K,
                            const int64_t* src_idx, const int64_t* norm_idx, int64_t M,
                            const int64_t* uidx, int64_t G, lspiv_projection** handle);
int lspiv_project_frames(lspiv_projection* handle, const void* frames, int dtype, int64_t T, float* out);
int lspiv_project_frames_dev(lspiv_projection* handle, const void* d_frames, int dtype, int64_t T, float* d_out,
                             void* stream);
/* A plan without group means (G = 0, Frames.project with a reducer other than "mean": pyorc/project.py:196-199) gives every
 * cell a source byte or 0, so a uint8 stack may stay uint8: (T, dst_h, dst_w) uint8 holding the values lspiv_project_frames
 * returns as float32 -- a quarter of the bytes, and get_piv runs its uint8 kernels on them.  LSPIV_EINVAL for a plan
 * with group means.                                                                                            */
int lspiv_project_frames_u8(lspiv_projection* handle, const uint8_t* frames, int64_t T, uint8_t* out);
int lspiv_project_frames_u8_dev(lspiv_projection* handle, const uint8_t* d_frames, int64_t T, uint8_t* d_out, void* stream);
int lspiv_projection_destroy(lspiv_projection* handle);

/* N1, method "cv" -- replaces pyorc.project.project_cv (pyorc/project.py:56-120): cv2.undistort(img, camera_matrix,
 * dist_coeffs) (pyorc/cv.py:1392-1413) followed by cv2.warpPerspective(img, M, (dst_w, dst_h), flags=INTER_AREA)
 * (pyorc/cv.py:993-1013; OpenCV turns INTER_AREA into INTER_LINEAR there).  OpenCV's published algorithm is restated --
 * initUndistortRectifyMap / the inverted homography evaluated in double, source coordinates quantised to 1/32 pixel,
 * fixed-point bilinear blend (2^15 weights, round half up) for uint8 and the float32 weight table for float32 frames,
 * BORDER_CONSTANT 0 -- NOT pinned against a real cv2 (absent here).  camera_matrix 3x3 row-major or NULL (no
 * undistortion step); dist_coeffs k1 k2 p1 p2 [k3 [k4 k5 k6]] (n_dist 0, 4, 5 or 8); M the 3x3 source-to-destination
 * homography of cv2.getPerspectiveTransform (pyorc/cv.py:769-795).  Frames uint8 or float32; the output has the dtype
 * of the input, (T, dst_h, dst_w), like the reference (pyorc/project.py:112). */
typedef struct lspiv_remap lspiv_remap;
int lspiv_project_cv_create(int64_t src_h, int64_t src_w, int64_t dst_h, int64_t dst_w, const double* camera_matrix,
                            const double* dist_coeffs, int n_dist, const double* M, lspiv_remap** handle);
int lspiv_project_cv_frames(lspiv_remap* handle, const void* frames, int dtype, int64_t T, void* out);
int lspiv_project_cv_frames_dev(lspiv_remap* handle, const void* d_frames, int dtype, int64_t T, void* d_out, void* stream);
int lspiv_project_cv_destroy(lspiv_remap* handle);

/* N2 -- element-wise pre-processing filters the reference writes in plain numpy (bit-reproducible):
 *   lspiv_time_diff   Frames.time_diff (pyorc/api/frames.py:409-436): out (T-1,H,W) float32 = f32(frame t+1) - f32(frame t),
 *                     values <= thres and NaN -> 0, |.| if use_abs
 *   lspiv_minmax      Frames.minmax (:344-362) on float32 frames: maximum(minimum(x, hi), lo), NaN propagates
 *   lspiv_reduce_rolling  Frames.reduce_rolling (:381-407) on uint8 frames, float64 arithmetic: trailing rolling mean over
 *                     `samples` frames removed, clipped at 0, per-frame (x * 255 / max) -> uint8, 0 where the rolling mean is 0;
 *                     frames without a complete window and frames whose maximum is 0 come out 0 (NaN.astype(uint8) on x86)
 *   lspiv_time_range  Frames.range (:364-379): out (H,W) in the frames' dtype = max over time - min over time (NaN skipped
 *                     for float frames, an all-NaN pixel stays NaN)
 *   lspiv_normalize   Frames.normalize (:279-306) on uint8 frames: float32 mean of frames [::round(T/samples)] removed,
 *                     per-frame ((x - min) / (max - min) * 255) -> uint8
 *   lspiv_gaussian_blur / lspiv_edge_detect   Frames.smooth (:438-467) / Frames.edge_detect (:308-342): pyorc calls
 *                     cv2.GaussianBlur(float32, (k, k), 0) (pyorc/cv.py:142-183); OpenCV's published algorithm is
 *                     restated (coefficient tables for k <= 7, separable, BORDER_REFLECT_101): NOT bit-pinned, ~1e-6.
 *                     out (T,H,W) float32; ksize odd, 1..31; edge = blur(ksize_2) - blur(ksize_1).
 * Host pointers; *_dev take device pointers. */
int lspiv_gaussian_blur(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize, float* out);
int lspiv_gaussian_blur_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize, float* d_out,
                            void* stream);
int lspiv_edge_detect(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize_1, int ksize_2, float* out);
int lspiv_edge_detect_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize_1, int ksize_2,
                          float* d_out, void* stream);
/* Frames.edge_detect followed by Frames.minmax (the reference's recipe order, examples/ngwerere/ngwerere.yml:6-11; pyorc/api/frames.py:308-362)
 * in one pass: np.maximum(np.minimum(x, hi), lo) applied as the band-filtered value is stored -- the bits of lspiv_edge_detect_dev +
 * lspiv_minmax_dev without the second trip over the float32 stack.  -INFINITY / INFINITY switch a limit off. */
int lspiv_edge_detect_clip_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, int ksize_1, int ksize_2, float lo,
                               float hi, float* d_out, void* stream);
int lspiv_time_diff(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, float thres, int use_abs, float* out);
int lspiv_time_diff_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, float thres, int use_abs,
                        float* d_out, void* stream);
int lspiv_time_range(const void* frames, int dtype, int64_t T, int64_t H, int64_t W, void* out);
int lspiv_time_range_dev(const void* d_frames, int dtype, int64_t T, int64_t H, int64_t W, void* d_out, void* stream);
int lspiv_minmax(const float* frames, int64_t n, float lo, float hi, float* out);
int lspiv_minmax_dev(const float* d_frames, int64_t n, float lo, float hi, float* d_out, void* stream);
int lspiv_normalize(const uint8_t* frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* out);
int lspiv_normalize_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* d_out, void* stream);
/* The two halves of lspiv_normalize_dev, for a stack that arrives in pieces (pyorc_amd/pipeline.py streams the camera frames
 * in while earlier ones are processed): the float32 mean plane (H,W) of frames [::round(T/samples)] of a T-frame stack of
 * which only those frames need to be resident yet, and the per-frame stretch of any run of frames against a given mean plane.
 * mean + apply over the whole stack == lspiv_normalize_dev bit for bit (the stretch uses per-frame statistics only). */
int lspiv_normalize_mean_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, int samples, float* d_mean, void* stream);
int lspiv_normalize_apply_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, const float* d_mean, uint8_t* d_out,
                              void* stream);
int lspiv_reduce_rolling(const uint8_t* frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* out);
int lspiv_reduce_rolling_dev(const uint8_t* d_frames, int64_t T, int64_t H, int64_t W, int samples, uint8_t* d_out, void* stream);

/* N3 -- post-PIV masks, ds.velocimetry.mask.* (pyorc/api/mask.py:147-403), on the result block
 * fields = [v_x | v_y | corr | s2n], each (T, R, C) float32 (lspiv_piv_pairs_dev's d_out after
 * lspiv_scale_velocity_dev).  mask: uint8, 1 = keep; (T, R, C), or (R, C) for the time-reducing kinds.
 * xarray's expressions are restated through the numpy calls they dispatch to (oracle/mask_oracle.py; unpinned
 * against a real xarray).  Reference quirks kept: stack_window's y range excludes +wdw (helpers.py:672-679),
 * variance's np.maximum(mean, 1e30), rolling's NaN edges.
 *   kind                     params (doubles)                                         mask shape
 *   LSPIV_MASK_MINMAX      0 s_min, s_max                                 :147-160    (T,R,C)
 *   LSPIV_MASK_ANGLE       1 angle_expected, angle_tolerance              :162-185    (T,R,C)
 *   LSPIV_MASK_COUNT       2 tolerance                                    :187-201    (R,C)
 *   LSPIV_MASK_CORR        3 tolerance                                    :203-213    (T,R,C)
 *   LSPIV_MASK_S2N         4 tolerance                                    :215-225    (T,R,C)
 *   LSPIV_MASK_OUTLIERS    5 tolerance, mode (0 "or", 1 "and")            :227-252    (T,R,C)
 *   LSPIV_MASK_VARIANCE    6 tolerance, mode                              :254-284    (R,C)
 *   LSPIV_MASK_ROLLING     7 wdw, tolerance                               :286-303    (T,R,C)
 *   LSPIV_MASK_WINDOW_NAN  8 tolerance, x_min, x_max, y_min, y_max        :305-340    (T,R,C)
 *   LSPIV_MASK_WINDOW_MEAN 9 tolerance, mode, x_min, x_max, y_min, y_max  :342-383    (T,R,C)
 * lspiv_mask_apply    ds[var].where(mask) on all four variables (:132-145); mask_has_time 0 for an (R,C) mask.
 * lspiv_time_mean     ds.mean(dim="time") -- the reduce_time=True pre-step (:51-52); out (4, R, C).
 * lspiv_window_replace  fillna with the neighbourhood nanmean, iter times (:385-403), in place.
 * lspiv_scale_velocity_dev  px/frame -> m/s in place on the first two planes: (u * res / dt).astype(float32),
 *                     pyorc/velocimetry/ffpiv.py:418-419; dt: HOST array, one entry per time step. */
#define LSPIV_MASK_MINMAX 0
#define LSPIV_MASK_ANGLE 1
#define LSPIV_MASK_COUNT 2
#define LSPIV_MASK_CORR 3
#define LSPIV_MASK_S2N 4
#define LSPIV_MASK_OUTLIERS 5
#define LSPIV_MASK_VARIANCE 6
#define LSPIV_MASK_ROLLING 7
#define LSPIV_MASK_WINDOW_NAN 8
#define LSPIV_MASK_WINDOW_MEAN 9
int lspiv_mask(const float* fields, int64_t T, int64_t R, int64_t C, int kind, const double* params, int n_params,
               uint8_t* mask);
int lspiv_mask_dev(const float* d_fields, int64_t T, int64_t R, int64_t C, int kind, const double* params, int n_params,
                   uint8_t* d_mask, void* stream);
int lspiv_mask_apply(float* fields, int64_t T, int64_t R, int64_t C, const uint8_t* mask, int mask_has_time);
int lspiv_mask_apply_dev(float* d_fields, int64_t T, int64_t R, int64_t C, const uint8_t* d_mask, int mask_has_time,
                         void* stream);
int lspiv_time_mean(const float* fields, int64_t T, int64_t R, int64_t C, float* out);
int lspiv_time_mean_dev(const float* d_fields, int64_t T, int64_t R, int64_t C, float* d_out, void* stream);
int lspiv_window_replace(float* fields, int64_t T, int64_t R, int64_t C, int x_min, int x_max, int y_min, int y_max,
                         int iter);
int lspiv_window_replace_dev(float* d_fields, int64_t T, int64_t R, int64_t C, int x_min, int x_max, int y_min,
                             int y_max, int iter, void* stream);
int lspiv_scale_velocity_dev(float* d_fields, int64_t T, int64_t n_vec, double res_x, double res_y, const double* dt,
                             void* stream);

/* N4 -- on-disk packing of the result variables (pyorc/const.py:80-83: int16, scale_factor 0.01,
 * _FillValue -9999; arithmetic of xarray's encoder: float32 x / float32 scale, NaN -> fill, round half even). */
int lspiv_pack_int16(const float* values, int64_t n, float scale, int fill, int16_t* packed);
int lspiv_pack_int16_dev(const float* d_values, int64_t n, float scale, int fill, int16_t* d_packed, void* stream);

/* ---------------------------------------------------------------- multi-GPU exchange ------ */
/* One process per GPU.  The reference is a single process: nothing is replaced here; frame
 * pairs are independent (docs/user-guide/velocimetry/index.rst:12-13) and the time axis is already cut into chunks
 * with a one-frame halo (pyorc/velocimetry/ffpiv.py:140), so rank r owns a contiguous block of pairs and the only
 * exchange is ONE all-gather of the packed (4, t, y, x) result block -- or, in ensemble mode, one sum all-reduce of
 * corr_sum / corr_count (ffpiv.py:361-363) before lspiv_ensemble_finish.
 *   transport LSPIV_COMM_RCCL  RCCL over xGMI; librccl.so is loaded on first use; the communicator binds to the
 *                              calling thread's current device (lspiv_set_device first);
 *   transport LSPIV_COMM_SHM   POSIX shared memory on the node, staged through host memory: plumbing tests only
 *                              (RCCL refuses two ranks on one GPU and needs a GPU; this one needs neither).
 * Rendezvous: rank 0 calls lspiv_comm_unique_id and hands the LSPIV_COMM_ID_BYTES to the other ranks (file, pipe,
 * environment); every rank then calls lspiv_comm_init.  "_dev" collectives take device pointers and are asynchronous
 * on `stream` with RCCL; the others take host pointers and block.  count = elements PER RANK; recv of an all-gather
 * holds world * count elements in rank order.  dtype LSPIV_F32 / LSPIV_F64; op LSPIV_COMM_SUM / LSPIV_COMM_MAX. */
#define LSPIV_COMM_RCCL 0
#define LSPIV_COMM_SHM 1
#define LSPIV_COMM_SUM 0
#define LSPIV_COMM_MAX 1
#define LSPIV_COMM_ID_BYTES 128
typedef struct lspiv_comm lspiv_comm;
int lspiv_comm_unique_id(int transport, void* id /* LSPIV_COMM_ID_BYTES */);
int lspiv_comm_init(int rank, int world, const void* id, int transport, lspiv_comm** comm);
/* backend_ranks: the rank count the transport itself reports (ncclCommCount / attached shm ranks) */
int lspiv_comm_info(lspiv_comm* comm, int* rank, int* world, int* transport, int* backend_ranks);
int lspiv_comm_allgather_dev(lspiv_comm* comm, const void* d_send, void* d_recv, int64_t count, int dtype, void* stream);
int lspiv_comm_allreduce_dev(lspiv_comm* comm, const void* d_send, void* d_recv, int64_t count, int dtype, int op,
                             void* stream);
int lspiv_comm_allgather(lspiv_comm* comm, const void* send, void* recv, int64_t count, int dtype);
int lspiv_comm_allreduce(lspiv_comm* comm, const void* send, void* recv, int64_t count, int dtype, int op);
int lspiv_comm_barrier(lspiv_comm* comm);
int lspiv_comm_destroy(lspiv_comm* comm);

/* ---------------------------------------------------------------- device-resident helpers  */
/* For hosts that keep stacks in HBM (bench.py, one-process-per-GPU shards).                 */
int lspiv_dev_malloc(void** d_ptr, size_t bytes);
int lspiv_dev_free(void* d_ptr);
/* pinned host memory: a uint8 / float32 stack that lives in it is DMA'd in place by lspiv_piv_pairs (no staging copy). */
int lspiv_host_alloc(void** h_ptr, size_t bytes);
int lspiv_host_free(void* h_ptr);
int lspiv_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int lspiv_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int lspiv_memset_dev(void* d_ptr, int value, size_t bytes);
/* HIP events recorded on the library's launch stream (bench.py's live kernel timing). */
int lspiv_event_create(void** ev);
int lspiv_event_record(void* ev);
int lspiv_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises ev_stop */
int lspiv_event_destroy(void* ev);
/* extra HIP streams for hosts that overlap the result exchange with the next launch (bench.py --gpus N): every "_dev"
 * entry point takes such a handle; events recorded on one stream can be waited for on another. */
int lspiv_stream_create(void** stream);
/* the same with a scheduling priority: > 0 the device's highest, < 0 its lowest, 0 = lspiv_stream_create.  pyorc_amd.shard puts the
 * result exchange on a HIGH-priority stream: RCCL's few workgroups must find a compute unit while a PIV kernel of ~80 000
 * workgroups is draining through all of them on the other stream. */
int lspiv_stream_create_priority(void** stream, int priority);
int lspiv_stream_destroy(void* stream);
/* a stream the CALLER created (hipStreamCreate) and passed to "_dev" entry points: free what the library keeps for it (the
 * rescue lists and their counters); NULL = the library's own stream.  lspiv_stream_destroy does this for its own streams. */
int lspiv_stream_release(void* stream);
int lspiv_stream_synchronize(void* stream);             /* NULL = the library's launch stream */
int lspiv_event_record_on(void* ev, void* stream);      /* NULL = the library's launch stream */
int lspiv_stream_wait_event(void* stream, void* ev);    /* NULL = the library's launch stream */

/* Synthetic particle-image stack rendered directly into HBM (SURVEY.md section 8d workload; test and
 * bench utility, not on the PIV path): N_p = density*H*W Gaussian particles (sigma 1.2 px)
 * advected by u = 3 + 2 sin(2 pi y/H), v = 1.5 cos(2 pi x/W) px/frame.  d_frames (T,H,W) uint8. */
int lspiv_synth_particles_dev(void* d_frames, int64_t T, int64_t H, int64_t W, uint64_t seed, float density);
/* test hook, host only: the float64 -> float32 conversion the host entry points apply while staging (csrc/host_stage.cpp), with
 * the DC-offset rule of the "narrow_offset" option at threshold min_abs (-1: plain conversion); offsets (nullable) receives the
 * offset taken off every frame; returns the number of staging threads. */
int lspiv_debug_narrow(const double* frames, int64_t frame_elems, int64_t n_frames, int min_abs, float* out, double* offsets);

/* Test hook: `count` independent length-n complex FFTs with the kernels' own register transforms (fft_regs.h), numpy
 * conventions (forward exp(-2 pi i jk/n), inverse exp(+...), both unnormalised); in / out: host arrays of count*n
 * interleaved (re, im) float pairs; n = 8, 16, 32, 64 or any even length 6..62 (the sizes lspiv_kernel_kind maps to 6 / 8). */
int lspiv_debug_fft(int n, int inverse, const float* in, float* out, int64_t count);
/* Test hook (host only): how the time-walking kernels cut a chunk of n_pairs pairs that starts at absolute pair index
 * pair_offset into segments of seg_len pairs anchored at multiples of seg_len: pairs in the first segment, segment count. */
int lspiv_debug_segments(int64_t n_pairs, int64_t pair_offset, int seg_len, int64_t* seg_first, int64_t* n_seg);

/* Measurement (round 5): with lspiv_set_option("time_kernel", 1) every launch records HIP events on its own stream right before and
 * right after its PIV kernel(s) -- not around the rescue kernels that follow.  lspiv_kernel_times returns the durations [ms] of the
 * last *n <= min(cap, 16) launches of the calling thread's device, oldest first, and empties the ring: the dominant kernel's time
 * inside the launch as a caller issues it, which is what `rocprofv3 --kernel-trace` reports for that kernel (bench.py:
 * roofline.achieved). */
int lspiv_kernel_times(float* ms, int cap, int* n);

/* Measurement / tests (round 6): lspiv_trace(1) makes the library record HIP event pairs -- on the streams the work runs on --
 * around the spans below, lspiv_trace(0) stops and forgets them; lspiv_trace_read returns up to `cap` of the *n recorded spans of the
 * calling thread's device, in recording order, as milliseconds since lspiv_trace(1) (it synchronises the device first).
 *   LSPIV_TRACE_PIV_HOST      one host-pointer PIV call (lspiv_piv_pairs[_at], lspiv_piv_velocity_at): from before its first kernel
 *                             to after its last download;
 *   LSPIV_TRACE_PROJECT_HOST  the projection kernel(s) of one host-pointer projection call (lspiv_project_frames[_u8],
 *                             lspiv_project_cv_frames), between its upload and its download.
 * A projection span that starts inside a PIV span shows what a wall clock cannot: the two calls, issued by two host threads on one
 * device, overlapped on the GPU (they share neither lock, nor stream, nor workspaces since round 6). */
#define LSPIV_TRACE_PIV_HOST 0
#define LSPIV_TRACE_PROJECT_HOST 1
int lspiv_trace(int enable);
int lspiv_trace_read(int64_t cap, int32_t* kind, double* start_ms, double* end_ms, int64_t* n);

/* Test hook (round 6): the orthoprojection kernel for plans with group means computes float(sum) / float(count) as q = RN(s y),
 * q' = RN(q + (s - q c) y) with y = RN(1 / c) instead of the division sequence; *mismatches = the number of (s, c), 0 <= s <= 255 c,
 * 1 <= c <= 255 -- every sum of c uint8 samples --, for which that differs from s / c on the device (must be 0). */
int lspiv_debug_project_division(int* mismatches);

/* Test hook (host only, no HIP call): hold one of device `device`'s locks -- 0 the host-pointer PIV entry points' workspaces, 1 a launch
 * and its rescue kernels, 2 the rescue lists, 3 / 4 the two slots of the host-pointer projection entry points -- for `milliseconds`.  The locks are per device (round 5; process-wide before): two
 * threads holding the same lock of two devices overlap, of one device queue.  tests/test_host.py times exactly that. */
int lspiv_debug_hold_lock(int device, int which, int milliseconds);

#ifdef __cplusplus
}
#endif
#endif /* LSPIV_H */

/* sofima_amd.h -- C ABI of libsofima_amd.so (MI355X / gfx950).
 *
 * The library is the drop-in device back end for the two compute cores of
 * SOFIMA.  The reference has no FFI: its "operator API" is the Python
 * signatures of flow_field.py / mesh.py.  Each entry point below names the
 * reference function (file:line under /root/reference) whose device work it
 * replaces; sofima_amd/flow_field.py and sofima_amd/mesh.py bind them through
 * ctypes with exactly the reference's Python signatures.
 *
 * Conventions
 *   - plain C types only; every buffer is CALLER-OWNED device memory (HIP
 *     pointers, e.g. torch.Tensor.data_ptr()); the library never frees or
 *     keeps caller memory, and scratch space is passed in as `workspace`.  The
 *     only device memory it owns: the twiddle tables of the hand-written FFT
 *     (8 bytes per sample of a transform length, per device), cached for the
 *     process;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream).  Entry points do not synchronise unless stated;
 *   - return 0 on success, a negative SFM_ERR_* otherwise; the message is in
 *     sfm_last_error() (thread-local).  No C++ exception crosses the boundary;
 *   - image / mask / surface arrays are row-major [z,]y,x; mesh arrays are
 *     [C, batch, z, y, x] with C = 2|3 vector components in x,y[,z] order;
 *   - shapes, sizes and starts are given in [z]yx order padded to 3 entries:
 *     for 2-D data entry 0 is 1 (sizes) or 0 (starts).
 */
#ifndef SOFIMA_AMD_H_
#define SOFIMA_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_ABI_VERSION 8   /* 4: SfmProfile.mfma_issued; sfm_mesh_relax_banded; 5: SfmWarpDesc.coord_map_f64;
                                6: SfmBandedDesc.host_halo / host_allgather / host_user;
                                7: sfm_ndimage_warp; 8: SfmProfile.tiles_abandoned */

#define SFM_OK 0
#define SFM_ERR_INVALID (-1)     /* bad argument / unsupported combination */
#define SFM_ERR_HIP (-2)         /* a HIP runtime call failed              */
#define SFM_ERR_WORKSPACE (-3)   /* workspace missing or too small         */
#define SFM_ERR_NO_DEVICE (-4)   /* no gfx950 device visible               */

#define SFM_DTYPE_U8 0
#define SFM_DTYPE_F32 1
#define SFM_DTYPE_U16 2          /* warping only                               */
#define SFM_DTYPE_I32 3          /* warping only: contiguous segment ids       */

/* Which correlation kernel family to use. AUTO picks MFMA_I8 whenever the
 * inputs qualify (uint8 images, 2-D, no masks in the correlation) and DIRECT
 * otherwise. */
#define SFM_XCORR_AUTO 0
#define SFM_XCORR_DIRECT 1    /* f32 shift-by-shift kernel, any dtype/dim/mask */
#define SFM_XCORR_MFMA_I8 2   /* int8 MFMA Toeplitz kernel                   */
#define SFM_XCORR_FFT 3       /* zero-padded rFFT form (3-D / large patches)  */

int sfm_version(void);
const char* sfm_last_error(void);
/* Number of visible HIP devices (0 and SFM_OK when none). */
int sfm_device_count(int* count);

/* Behaviour switches.  Every switch selects between kernels / schedules that
 * return the SAME results (the tests pin the variants against each other);
 * they exist for A/B measurements and for the tests themselves.  A switch set
 * here wins over the environment variable of the same name, which is only the
 * default; value NULL removes the explicit setting again.
 * sfm_get_option copies the value in effect (returns 1 and "" when unset).
 *   SFM_MFMA_PRUNE=0      correlation kernel computes every surface tile
 *   SFM_MFMA_LAZY=0       flow path stores every computed surface tile (default:
 *                         only the tiles the peak kernels can read)
 *   SFM_MFMA_EARLY=n      lazy path: least number of row groups between two tests that
 *                         abandon a provably cold tile inside its row loop, or narrow
 *                         it (default 1; each test schedules the next; 0: no tests)
 *   SFM_MFMA_NARROW=n     lazy path: widest in-flight narrowing of a row loop (outer column
 *                         tiles dropped once proved cold; default: down to the four
 *                         central tiles; 0: never)
 *   SFM_MFMA_WIDEN=1      lazy path: the store requests a patch starts with are widened
 *                         by one row tile (fewer recomputed tiles, more finished ones)
 *   SFM_MFMA_LAZYG=0      the prep kernel writes the whole correction table (default, pruned
 *                         flow launches: the finishing tiles build their 16 rows; the prep pass
 *                         then keeps no patch in LDS)
 *   SFM_MFMA_XCD=1        one patch queue per XCD (measured: no gain) instead of a flat one
 *   SFM_MFMA_PIPE=1       160-wide flow launches as a cross-patch pipeline: one workgroup of
 *                         eight waves per CU, two patch slots in LDS, the tile queue running
 *                         across the patch boundary (built in round 6, measured 3-4 % SLOWER
 *                         than two four-wave workgroups: profiles/r06_xcorr_phase_ticks.txt)
 *   SFM_MFMA_PIPE_ADMIT=n pipeline: tiles of a patch handed out before its first tile is done
 *                         (default 2)
 *   (measurement-only, honoured by a library built with -DSFM_MEASUREMENT_SWITCHES only --
 *   sfm_get_option("SFM_BUILD_MEASUREMENT_SWITCHES") = "1"; the production build ignores them:)
 *   SFM_MFMA_PROBE=0      no seed probe in front of the pruning
 *   SFM_MFMA_TOUCH_ALL=1  lazy path: pull a patch's whole correction table into L2
 *   SFM_MFMA_EXACT=0      run-time instead of compile-time column geometry
 *   SFM_MFMA_QUEUE=0      static instead of dynamic patch queue
 *   SFM_MFMA_PRIO=n       wave priority experiment (0..3)
 *   SFM_MFMA_MAX_WG_PER_CU=n   occupancy cap of the correlation kernel
 *   SFM_MASKED_FAST=0     masked patches always take all eight passes
 *   SFM_MASKED_DEADROWS=0 no overlap-rule skips in the masked assembly
 *   SFM_MASKED_GROUPS=n   masked matrix-core path: reference batches (`group` patches each)
 *                         processed per round of a call (default 8).  The workspace of a masked
 *                         call -- sfm_xcorr_workspace_bytes -- is sized for n groups (about 4 GB
 *                         per 1024 patches of 160^2): a caller with a tight memory budget and a
 *                         small `group` sets a smaller n BEFORE asking for the size
 *   SFM_PHASE_XCD=0       masked assembly without the XCD-aware tile order
 *   SFM_MESH_PERSISTENT=0 / SFM_MESH_SPECULATE=0 / SFM_MESH_TILED=0 /
 *   SFM_MESH_SMALL=0 / SFM_MESH_FUSE_TARGET=0
 *                         fall back to the simpler integrator
 *   SFM_MESH_PERSIST3D=1  volumetric montage (native target mesh): every step of a chunk in ONE
 *                         launch with grid barriers (built in round 6: bit-identical, 2 x SLOWER
 *                         than the four launches per step it replaces, which stay the default)
 *   SFM_MESH_MARCH3D=0|1  default-link volumes (elastic_mesh_3d): the z-march integrator that
 *                         evaluates every spring once -- never / for every volume (default:
 *                         from 1.5 * 10^6 nodes on).  Forces bit-identical to the per-node kernel.
 *   SFM_MESH_MARCH3D_T=256|512|1024 / SFM_MESH_MARCH3D_ZC=n
 *                         its workgroup size / planes per workgroup (default: planned per mesh)
 *   SFM_MESH_PACK=0       tiled in-plane step: a workgroup per tile also in a narrow last
 *                         tile column (default: its tile rows share workgroups)
 *   SFM_MESH_TILE=16|32   tile edge of the persistent integrator
 *   SFM_MESH_XCD=0|1      tiled in-plane step: one contiguous run of tiles per
 *                         XCD never / always (default: from 2048 tiles on)   */
int sfm_set_option(const char* name, const char* value);
int sfm_get_option(const char* name, char* value, size_t capacity);

/* Kernel timing hooks (used by bench.py for the roofline figures; no
 * reference counterpart).  While enabled, hipEvents on the launch stream
 * bracket every launch of the dominant kernels: the patch-correlation kernel
 * (kind 0) and the mesh integrate kernel (kind 1).  sfm_profile_read waits for
 * the recorded events, returns the accumulated kernel time and launch count
 * per kind since the last read, and resets the counters. */
typedef struct SfmProfile {
  double kernel_ms[2];
  int64_t launches[2];
  double clock_mhz[2];          /* sustained shader clock inside the kernels of
                                   that kind (s_memtime / s_memrealtime of one
                                   workgroup); 0 when not sampled             */
  int64_t tiles_skipped[2];     /* kind 0: surface row tiles the exact pruning of
                                   the fused-peaks kernel skipped / looked at   */
  int64_t tiles_drawn[2];
  int64_t col_tiles_skipped[2]; /* column tiles left out of the computed row tiles */
  int64_t mfma_issued[2];       /* kind 0: matrix instructions (16 x 16 x 64 int8 =
                                   32768 operations each) the kernel issued      */
  int64_t tiles_abandoned[2];   /* kind 0: row tiles given up inside their row loop
                                   (proved cold by the rows still to come)       */
} SfmProfile;
int sfm_profile_enable(int on);
int sfm_profile_read(SfmProfile* out);

/* ------------------------------------------------------------------------
 * Patch cross-correlation + peak statistics for ONE batch.
 * Replaces flow_field.batched_xcorr_peaks (flow_field.py:374-441) =
 * _batched_xcorr (:278-371) + masked_xcorr(use_jax=True) (:36-156) +
 * _batched_peaks (:205-275) + _peak_stats (:178-202).
 * ---------------------------------------------------------------------- */
typedef struct SfmXcorrDesc {
  int32_t ndim;                 /* 2 or 3                                    */
  int32_t dtype;                /* SFM_DTYPE_* of pre/post image             */
  const void* pre_image;        /* device, [z,]y,x                           */
  const void* post_image;
  const uint8_t* pre_mask;      /* device bool bytes (1 = invalid) or NULL   */
  const uint8_t* post_mask;
  int32_t pre_shape[3];         /* [z]yx, entry 0 = 1 for 2-D                */
  int32_t post_shape[3];
  int32_t pre_mask_shape[3];    /* masks may differ in shape from the images */
  int32_t post_mask_shape[3];
  int32_t patch[3];             /* pre patch size  P                         */
  int32_t post_patch[3];        /* post patch size Q (<= P)                  */
  const int32_t* pre_starts;    /* device [batch, ndim] [z]yx, row-major     */
  const int32_t* post_starts;   /* device [batch, ndim]                      */
  int32_t batch;                /* rows in the starts arrays (incl. padding) */
  int32_t group;                /* rows per reference batch, see below; 0 = batch */
  int32_t use_mean;             /* 0: subtract per-patch mean; 1: use `mean` */
  float mean;
  int32_t min_distance;         /* max-filter half width (reference: 2)      */
  float threshold_rel;          /* reference: 0.5                            */
  int32_t peak_radius[3];       /* [z]yx sharpness window radius             */
  int32_t method;               /* SFM_XCORR_*                               */
  void* workspace;              /* device scratch                            */
  size_t workspace_bytes;
  void* stream;
} SfmXcorrDesc;

/* Scratch bytes sfm_xcorr_peaks / sfm_xcorr_surface need for this desc. */
size_t sfm_xcorr_workspace_bytes(const SfmXcorrDesc* desc);

/* peaks: device float [batch, ndim + 2] = x, y[, z], sharpness, ratio.
 * The batch-coupled behaviours of the reference (second-peak suppression set,
 * batch-global masked-NCC tolerances) are computed over every run of `group`
 * consecutive rows separately (the last run may be shorter): one call can
 * carry many reference batches (flow_field.py:610-699) in one launch and
 * still returns what the reference returns batch by batch. */
int sfm_xcorr_peaks(const SfmXcorrDesc* desc, float* peaks);

/* surface: device float [batch, *(P + Q - 1)]; the array masked_xcorr returns
 * for the mean-subtracted patches (flow_field.py:361-371). */
int sfm_xcorr_surface(const SfmXcorrDesc* desc, float* surface);

/* Masked-pixel count of every patch of the sampling grid.  Replaces
 * flow_field._integral_image (flow_field.py:159-175) + the box query of the
 * summed-area table (flow_field.py:575-589) used for patch selection:
 * counts[o] = number of non-zero mask bytes in the patch of size `patch` at
 * o * step, for o in [0, (shape - patch) / step] per axis. */
typedef struct SfmMaskCountDesc {
  int32_t ndim;
  int32_t shape[3];             /* mask [z]yx                                */
  int32_t patch[3];
  int32_t step[3];
  const uint8_t* mask;          /* device bool bytes                         */
  void* stream;
} SfmMaskCountDesc;

/* counts: device int32 [prod((shape - patch) / step + 1)], row-major. */
int sfm_mask_patch_counts(const SfmMaskCountDesc* desc, int32_t* counts);

/* ------------------------------------------------------------------------
 * The host loop of flow_field() on the device.
 * sfm_flow_starts replaces the per-batch start arithmetic (flow_field.py:
 * 620-680): post start = grid position * step; pre start = post start -
 * (patch - post_patch) / 2 clipped at 0; optional integer shifts looked up in
 * the targeting fields (nearest cell, nan_to_num, truncation) and kept inside
 * the images; final clip at 0.  sfm_flow_scatter replaces the per-patch result
 * scatter (:701-709), undoing the targeting shifts.
 * All shapes / steps are [z]yx padded to 3 entries in front.
 * ---------------------------------------------------------------------- */
typedef struct SfmFlowStartsDesc {
  int32_t ndim;
  int32_t n;                    /* patches (rows of `positions`)             */
  int32_t step[3];
  int32_t patch[3];
  int32_t post_patch[3];
  int32_t pre_shape[3];         /* image shapes                              */
  int32_t post_shape[3];
  const int32_t* positions;     /* device [n, ndim] grid positions, [z]yx    */
  const float* pre_targeting_field;   /* device [ndim, *shape] (x, y[, z]) or NULL */
  const float* post_targeting_field;
  int32_t pre_targeting_shape[3];
  int32_t post_targeting_shape[3];
  int32_t pre_targeting_step[3];
  int32_t post_targeting_step[3];
  int32_t* pre_starts;          /* out, device [n, ndim]                     */
  int32_t* post_starts;
  int32_t* pre_offsets;         /* out, device [n, ndim]; required with the  */
  int32_t* post_offsets;        /* corresponding targeting field             */
  void* stream;
} SfmFlowStartsDesc;

int sfm_flow_starts(const SfmFlowStartsDesc* desc);

typedef struct SfmFlowScatterDesc {
  int32_t ndim;
  int32_t n;
  int32_t grid[3];              /* output grid, [z]yx                        */
  const float* peaks;           /* device [n, ndim + 2]                      */
  const int32_t* positions;     /* device [n, ndim]                          */
  const int32_t* pre_offsets;   /* device [n, ndim] or NULL                  */
  const int32_t* post_offsets;
  float* out;                   /* device [ndim + 2, *grid], pre-filled (NaN) */
  void* stream;
} SfmFlowScatterDesc;

int sfm_flow_scatter(const SfmFlowScatterDesc* desc);

/* Stand-alone peak statistics, replaces _batched_peaks (flow_field.py:205-275)
 * for a caller-provided batch of surfaces. */
typedef struct SfmPeaksDesc {
  int32_t ndim;
  int32_t batch;
  int32_t shape[3];             /* surface [z]yx                             */
  float center_offset[3];       /* [z]yx                                     */
  int32_t min_distance;
  float threshold_rel;
  int32_t peak_radius[3];
  const float* surface;         /* device [batch, *shape]                    */
  void* workspace;
  size_t workspace_bytes;
  void* stream;
} SfmPeaksDesc;

size_t sfm_peaks_workspace_bytes(const SfmPeaksDesc* desc);
int sfm_peaks(const SfmPeaksDesc* desc, float* peaks);

/* ------------------------------------------------------------------------
 * Coordinate-map composition.
 * Replaces map_utils.compose_maps_fast (map_utils.py:616-734): bilinear /
 * trilinear resampling of map2 (absolute) at the targets of map1, result in
 * map1's relative format.  Maps are float [C, z, y, x], C = 2 (every z
 * section independently in-plane) or 3 (volumetric).
 * ---------------------------------------------------------------------- */
#define SFM_INTERP_NEAREST 0   /* out-of-range corner indices are clamped      */
#define SFM_INTERP_CONSTANT 1  /* out-of-range corners read as NaN             */

typedef struct SfmComposeDesc {
  int32_t ncomp;                /* 2 or 3                                    */
  int32_t mode;                 /* SFM_INTERP_*                              */
  int32_t shape1[3];            /* z, y, x of map1 (and of the output)       */
  int32_t shape2[3];            /* z, y, x of map2                           */
  float start1[3];              /* zyx origin of map1 (z ignored for C = 2)  */
  float start2[3];
  float stride1[3];             /* zyx node spacing of map1                  */
  float stride2[3];
  const float* map1;            /* device [C, *shape1]                       */
  const float* map2;            /* device [C, *shape2]                       */
  void* stream;
} SfmComposeDesc;

int sfm_compose_maps(const SfmComposeDesc* desc, float* out);

/* ------------------------------------------------------------------------
 * Flow-field clean-up, the step between flow estimation and mesh relaxation.
 * Replaces flow_utils.clean_flow (flow_utils.py:37-78): invalidates (NaN)
 * vectors with low peak sharpness / peak ratio, too large components, or a
 * too large deviation from the 3 x 3 (x 3) median of the nan_to_num'ed field
 * (scipy.ndimage.median_filter, mode "reflect").
 * ---------------------------------------------------------------------- */
typedef struct SfmCleanFlowDesc {
  int32_t dim;                  /* 2 or 3 spatial dimensions                 */
  int32_t channels;             /* dim .. dim + 2 input channels             */
  int32_t shape[3];             /* z, y, x                                   */
  float min_peak_ratio;
  float min_peak_sharpness;
  float max_magnitude;          /* <= 0: not applied                         */
  float max_deviation;          /* <= 0: not applied                         */
  const float* flow;            /* device [channels, z, y, x]                */
  void* stream;
} SfmCleanFlowDesc;

/* out: device float [dim, z, y, x]. */
int sfm_clean_flow(const SfmCleanFlowDesc* desc, float* out);

/* ------------------------------------------------------------------------
 * Fold / stretch detection on a relaxed mesh, the step after relaxation.
 * Replaces map_utils.mask_irregular (map_utils.py:737-786): a node is bad when
 * the distance to its +x (+y) neighbour leaves [frac, max_frac] * stride; the
 * bad set is dilated `dilation_iters` times with the full 3 x 3 structure and
 * the map is NaN'ed there in place.
 * ---------------------------------------------------------------------- */
typedef struct SfmMaskIrregularDesc {
  int32_t shape[2];             /* y, x                                      */
  float stride[2];              /* x, y (the reference's order)              */
  float frac;
  float max_frac;
  int32_t dilation_iters;
  void* stream;
} SfmMaskIrregularDesc;

/* coord_map: device float [2, y, x], modified in place; bad: device uint8
 * [y, x] (1 = masked). */
int sfm_mask_irregular(const SfmMaskIrregularDesc* desc, float* coord_map,
                       uint8_t* bad);

/* ------------------------------------------------------------------------
 * Warping one image section by an inverse coordinate map.
 * Replaces the per-section body of warp.warp_subvolume (warp.py:145-167):
 * linear (extrapolating) interpolation of the map nodes to every output
 * pixel, cv2.convertMaps to 1/32-pixel fixed point, cv2.remap with a zero
 * border -- fused, one pass over the output.
 * ---------------------------------------------------------------------- */
#define SFM_WARP_NEAREST 0
#define SFM_WARP_TABLE 1         /* taps weighted by `weights` (linear, cubic,
                                    lanczos4 differ only in the table)        */
typedef struct SfmWarpDesc {
  int32_t dtype;                /* SFM_DTYPE_* of image and out              */
  int32_t interpolation;        /* SFM_WARP_*                                */
  int32_t ksize;                /* taps per axis: 2, 4 or 8                  */
  int32_t image_shape[2];       /* y, x                                      */
  int32_t map_shape[2];         /* y, x nodes (>= 2 x 2)                     */
  int32_t out_shape[2];         /* y, x                                      */
  double map_origin[2];         /* y, x of map node (0, 0) in output pixels  */
  double stride;                /* output pixels per map node                */
  const void* image;            /* device [y, x]                             */
  const void* coord_map;        /* device [2, y, x]: ABSOLUTE source (x, y)
                                   coordinates in image pixels, float32 (or
                                   float64 when coord_map_f64 is set: the
                                   reference interpolates in the map's dtype,
                                   warp.py:125-150)                          */
  const void* weights;          /* device [32 * 32, ksize * ksize]: int16 with
                                   15 fractional bits for u8 images, float
                                   otherwise; phase = 32 * fy + fx            */
  void* out;                    /* device [y, x]                             */
  void* stream;
  int32_t coord_map_f64;        /* 1: coord_map holds doubles                */
} SfmWarpDesc;

int sfm_warp_section(const SfmWarpDesc* desc);

/* ------------------------------------------------------------------------
 * warp.ndimage_warp (warp.py:189-335), the SciPy rendering path, 2-D and 3-D,
 * as ONE kernel: per output voxel the node-space position (c - offset) / stride,
 * the order-1 interpolation of the absolute source map (scipy.ndimage.
 * map_coordinates, mode "constant", cval 0: a position outside the node grid
 * gives the dense coordinate 0, like the reference), and the order-0 / order-1
 * sampling of the image at the dense coordinates, all in double like SciPy and
 * in SciPy's operation order (weights 1 - t and 1 - (1 - t), value x weights in
 * axis order, taps added in row-major order), so integer images come out bit
 * for bit and float images to the last bit.  The work-box decomposition of the
 * reference (work_size / overlap) does not change the result for these orders
 * and has no counterpart here.  Axes are z, y, x; 2-D inputs use y, x of the
 * last two entries with shape[0] = 1.
 * ---------------------------------------------------------------------- */
typedef struct SfmNdWarpDesc {
  int32_t ndim;                 /* 2 or 3                                      */
  int32_t dtype;                /* SFM_DTYPE_U8 | U16 | F32 of image and out   */
  int32_t order;                /* 0 nearest, 1 linear                         */
  int32_t image_shape[3];       /* z, y, x                                     */
  int32_t map_shape[3];         /* z, y, x nodes                               */
  int32_t out_shape[3];         /* z, y, x                                     */
  double stride[3];             /* z, y, x output voxels per map node          */
  double offset[3];             /* z, y, x of map node 0 relative to out voxel 0 */
  const void* image;            /* device [z, y, x]                            */
  const double* src_map;        /* device [ndim, z, y, x] float64, channels x,
                                   y[, z]: absolute source coordinates in image
                                   voxels (warp.py:250-265)                    */
  void* out;                    /* device [z, y, x]                            */
  void* stream;
} SfmNdWarpDesc;

int sfm_ndimage_warp(const SfmNdWarpDesc* desc);

/* ------------------------------------------------------------------------
 * Dynamic-range mask of an overlap strip, the step in front of the
 * whole-overlap correlation of a tile pair.
 * Replaces the mask construction of stitch_rigid._estimate_offset
 * (stitch_rigid.py:47-60): out = (maximum_filter(img, size) -
 * minimum_filter(img, size)) < range_limit, OR-ed with `extra_mask`
 * (scipy.ndimage filters: mode "reflect", window [i - size/2, i - size/2 +
 * size - 1]; uint8 images subtract in uint8).
 * ---------------------------------------------------------------------- */
typedef struct SfmRangeMaskDesc {
  int32_t dtype;                /* SFM_DTYPE_* of the image                  */
  int32_t shape[2];             /* y, x                                      */
  int32_t filter_size;
  double range_limit;
  const void* image;            /* device [y, x]                             */
  const uint8_t* extra_mask;    /* device bool bytes [y, x] or NULL          */
  void* stream;
} SfmRangeMaskDesc;

/* out: device uint8 [y, x] (1 = masked). */
int sfm_range_mask(const SfmRangeMaskDesc* desc, uint8_t* out);

/* ------------------------------------------------------------------------
 * Target mesh of an elastic tile montage.
 * Replaces stitch_elastic.compute_target_mesh (stitch_elastic.py:624-676,
 * with _update_mesh :573-620 and _apply_flow :456-570) vmapped over all
 * tiles: out[:, t] = positions in the meshes of tile t's neighbours that the
 * flow fields say its nodes correspond to; NaN where no neighbour overlaps.
 * ---------------------------------------------------------------------- */
typedef struct SfmTargetMeshDesc {
  int32_t ncomp;                /* 2 (in-plane montage) or 3 (volumetric)    */
  int32_t n_tiles;
  int32_t mesh_shape[3];        /* z, y, x of one tile mesh (z = 1 for 2)    */
  int32_t fx_shape[3];          /* z, y, x of one flow array in fx           */
  int32_t fy_shape[3];
  int32_t n_fx;                 /* flow arrays in fx (its dimension 1)       */
  int32_t n_fy;
  int32_t nbor_fields;          /* 8, or 11 with the z fields (ncomp 3)      */
  float stride[3];              /* zyx stride of flow and mesh               */
  const int32_t* nbors;         /* device [n_tiles, 4, nbor_fields]          */
  const float* fx;              /* device [ncomp, n_fx, *fx_shape]           */
  const float* fy;              /* device [ncomp, n_fy, *fy_shape]           */
  int32_t n_eval;               /* 0: all tiles; > 0: only the first n_eval
                                   rows of `nbors` are evaluated (the meshes in
                                   x still number n_tiles)                   */
} SfmTargetMeshDesc;

/* x: device float [ncomp, n_tiles, *mesh_shape]; out: [ncomp, n_eval or
 * n_tiles, *mesh_shape]. */
int sfm_target_mesh(const SfmTargetMeshDesc* desc, const float* x, float* out,
                    void* stream);

/* ------------------------------------------------------------------------
 * Spring mesh.
 * ---------------------------------------------------------------------- */
#define SFM_MESH_MAX_LINKS 13

/* Which mesh_force the integrator evaluates (the `mesh_force` argument of
 * mesh.relax_mesh / velocity_verlet, mesh.py:372-382). */
#define SFM_FORCE_SPRINGS 0    /* inplane_force / elastic_mesh_3d link stencils  */
#define SFM_FORCE_TILE_MESH 1  /* stitch_rigid.elastic_tile_mesh (stitch_rigid.py:
                                  330-388) for ncomp 2, elastic_tile_mesh_3d
                                  (:391-473) for ncomp 3: every node is a tile,
                                  cx / cy are the desired offsets to the +x / +y
                                  tile, force = displacement mismatch          */
#define SFM_FORCE_EXTERNAL 2   /* produced by the caller through force_cb        */

/* SFM_FORCE_EXTERNAL: called on the host before every force evaluation (once
 * for the initial a = F(x), then once per step after the position update).
 * It must enqueue, on SfmMeshDesc.stream, work that leaves mesh_force(x) in
 * SfmMeshDesc.ext_force.  Non-zero return aborts the chunk with SFM_ERR_INVALID. */
typedef int (*SfmForceCallback)(void* user);

typedef struct SfmMeshDesc {
  int32_t ncomp;                /* 2: inplane_force (mesh.py:42-169)
                                   3: elastic_mesh_3d (mesh.py:192-279)      */
  int32_t shape[4];             /* batch, z, y, x  (ncomp 2: batch = 1 and z
                                   slices are independent)                   */
  /* Configuration scalars are doubles so that host-side constant folding
     matches the reference's Python-float arithmetic before rounding to f32. */
  double stride[3];             /* x, y[, z] node spacing                    */
  double k;                     /* intra-mesh spring constant                */
  double k0;                    /* spring constant to `prev`                 */
  int32_t prefer_orig_order;
  int32_t n_links;              /* ncomp 3 only; 0 = the 13 default links    */
  int32_t links[SFM_MESH_MAX_LINKS][3];  /* xyz directions, |v| <= 1         */
  /* integrator (mesh.IntegrationConfig, mesh.py:282-338) */
  double dt;
  double gamma;
  int32_t num_iters;
  int32_t fire;
  double f_alpha, f_inc, f_dec, alpha0;  /* alpha0 = config.alpha            */
  int32_t n_min;
  double dt_max;                /* in units of dt, like the reference        */
  double final_cap;
  double cap_scale;
  int32_t cap_upscale_every;
  int32_t remove_drift;         /* 0 no; 1 subtract the global mean of x, v;
                                   2 per-x-column means: what the reference
                                   does for 5-D states (mesh.py:496-497)   */
  /* state, device float [ncomp, batch, z, y, x] */
  float* x;
  float* v;
  float* a;
  const float* prev;            /* or NULL                                   */
  void* workspace;
  size_t workspace_bytes;
  void* stream;
  /* Optional native prev_fn (mesh.py:429-430): when set, `prev` must be NULL
     and prev = target_mesh(x) is re-evaluated inside every force evaluation. */
  const SfmTargetMeshDesc* target;
  /* Force model; zero-initialised descriptors get the spring stencils. */
  int32_t force_kind;           /* SFM_FORCE_*                               */
  const float* cx;              /* TILE_MESH: device [ncomp, batch*z, y, x]  */
  const float* cy;
  float* ext_force;             /* EXTERNAL: device, same shape as x         */
  SfmForceCallback force_cb;
  void* force_user;
  /* Generic prev_fn (mesh.py:429-430, any callable): when prev_cb is set, `prev`
     and `target` must be NULL; the library calls it on the host in front of
     every force evaluation (initial a = F(x), then after each position update)
     and it must enqueue, on `stream`, work that leaves prev_fn(x) in ext_prev
     (device, same shape as x; NaN = no spring).  Non-zero return aborts. */
  float* ext_prev;
  SfmForceCallback prev_cb;
  void* prev_user;
} SfmMeshDesc;

/* FIRE scalars carried between chunks (mesh.py:449, :513, :589). */
typedef struct SfmFireState {
  float dt;
  float alpha;
  int32_t n_pos;
  float cap;
} SfmFireState;

typedef struct SfmChunkStats {
  float e_kin;                  /* sum |v|^2   (mesh.py:584-585)             */
  float v_max;                  /* max |v|     (mesh.py:586)                 */
} SfmChunkStats;

size_t sfm_mesh_workspace_bytes(const SfmMeshDesc* desc);

/* out = mesh_force(x): device float, same shape as x.  Uses desc->x only. */
int sfm_mesh_force(const SfmMeshDesc* desc, float* out);

/* One call of mesh.velocity_verlet (mesh.py:371-521): a = F(x); num_iters
 * damped-Verlet or FIRE steps on x, v, a in place; `fire` is in/out (n_pos
 * restarts at 0 like the reference); stats are copied back to the host, which
 * synchronises the stream. */
int sfm_mesh_relax_chunk(const SfmMeshDesc* desc, SfmFireState* fire,
                         SfmChunkStats* stats);

/* ------------------------------------------------------------------------
 * One mesh across GPUs (one process per GPU).  The reference has no
 * multi-device code; these entry points split the step of
 * sfm_mesh_relax_chunk (mesh.velocity_verlet, mesh.py:436-499) at its two
 * exchange points so that bands of rows of ONE mesh can live on different GPUs:
 * the 1-node halo of the 8 / 26-neighbour stencil (mesh.py:106-167, :230-277)
 * and the whole-array reductions of FIRE (power :455, drift means :494-497,
 * e_kin / v_max mesh.py:584-586).
 *
 * A band holds rows [own_y0, own_y1) of the local array plus the halo rows
 * next to them.  Per step the caller
 *   1. brings (x, v, a) of the halo rows from the neighbour bands and gathers
 *      every band's `my_sums` row into `sums` (sfm_comm_* below, or any other
 *      transport),
 *   2. sfm_mesh_shard_advance  (FIRE scalars from `sums`, reduced in rank order
 *      on every band -> identical everywhere; pending gate / drift; position
 *      update of owned and halo rows),
 *   3. sfm_mesh_shard_integrate (force, velocity update, partial sums of the
 *      owned rows -> my_sums).
 * Only SfmMeshDesc.prev (no prev_fn), spring / tile-mesh forces and
 * remove_drift 0 | 1 are supported.  Everything is enqueued on desc->stream;
 * only sfm_mesh_shard_finish synchronises.
 * ---------------------------------------------------------------------- */
typedef struct SfmMeshShard {
  int32_t own_y0, own_y1;       /* owned rows of the local [.., y, x] arrays  */
  int64_t global_nodes;         /* nodes of the whole mesh (drift mean)       */
  int32_t n_ranks;              /* rows of `sums`                             */
  float* sums;                  /* device [n_ranks, 8], gathered by the caller */
  float* my_sums;               /* device [8]: power, sum x[3], sum v[3], 0   */
  int32_t phase;                /* library state: steps advanced so far       */
  float cap0;                   /* library state: force cap without FIRE      */
} SfmMeshShard;

/* a = F(x) + pull on the local rows; FIRE scalars of the chunk from `fire`. */
int sfm_mesh_shard_begin(const SfmMeshDesc* desc, SfmMeshShard* shard,
                         const SfmFireState* fire);
int sfm_mesh_shard_advance(const SfmMeshDesc* desc, SfmMeshShard* shard);
int sfm_mesh_shard_integrate(const SfmMeshDesc* desc, SfmMeshShard* shard);
/* Pending gate / drift of the last step (needs the last `sums`); `fire` gets
 * the chunk's final scalars, `stats` e_kin and v_max of the OWNED rows. */
int sfm_mesh_shard_finish(const SfmMeshDesc* desc, SfmMeshShard* shard,
                          SfmFireState* fire, SfmChunkStats* stats);

/* The same split step with the loop INSIDE the library: one call runs
 * desc.num_iters steps of every local band, moving the edge rows between local
 * bands by device copies and between ranks through `comm` (grouped RCCL
 * send / recv), all-gathering the bands' partial sums once per step, with the
 * exchange of a step's edge rows overlapped with the integration of its
 * interior rows (second stream).  Band g = rank * n_local + i owns rows
 * [own_y0, own_y1) of bands[i]'s arrays; shards[i].global_nodes must be set,
 * sums / my_sums / n_ranks are filled in by the library.  In-plane spring
 * meshes need SfmMeshDesc.workspace as for sfm_mesh_relax_chunk (it holds the
 * second state set).  `fire` in/out and `stats` are those of the WHOLE mesh
 * (e_kin added up in band order, v_max the maximum), identical on every rank.
 * Synchronises desc.stream once, at the end of the chunk. */
#define SFM_BANDED_LOOPBACK 1   /* exchanges between LOCAL bands travel through
                                   comm as self send / recv too (exercises the
                                   RCCL path on one GPU)                       */
#define SFM_BANDED_NO_OVERLAP 2 /* one launch per band and step, exchange after it */
/* Host-staged transport (twin of force_cb / prev_cb): used between ranks when
 * `comm` is NULL and n_ranks > 1, so that the inter-rank branch of the loop --
 * peer indexing, packed edge rows, the in-place all-gather layout with several
 * bands per rank, the edge-first split on the second stream -- also runs where
 * RCCL cannot connect the ranks (several processes sharing one GPU, CPU-side
 * process groups).  The library synchronises the exchange stream, copies the
 * packed rows to host memory, calls back, and copies the received rows to the
 * device; every pointer the callbacks see is HOST memory.  Argument meaning as
 * sfm_comm_halo_exchange / sfm_comm_allgather below (peer = -1: no neighbour on
 * that side; `recv` of the all-gather holds n_ranks * count floats, rank r's
 * part at r * count, and the callee fills every part).  Return 0, or non-zero
 * to abort the chunk (SFM_ERR_INVALID, like force_cb). */
typedef int (*SfmHostHaloFn)(void* user, int peer_lo, const float* send_lo, float* recv_lo,
                             int peer_hi, const float* send_hi, float* recv_hi, size_t count);
typedef int (*SfmHostAllgatherFn)(void* user, const float* send, float* recv, size_t count);
struct SfmComm;
typedef struct SfmBandedDesc {
  int32_t n_local;              /* bands of this rank, consecutive, in y order  */
  const SfmMeshDesc* bands;     /* [n_local]                                   */
  SfmMeshShard* shards;         /* [n_local]                                   */
  struct SfmComm* comm;         /* NULL: this process holds the whole mesh      */
  int32_t rank;
  int32_t n_ranks;              /* every rank holds n_local bands               */
  int32_t flags;                /* SFM_BANDED_*                                 */
  void* comm_stream;            /* exchange stream; NULL: bands[0].stream       */
  void* scratch;                /* device, sfm_mesh_banded_scratch_bytes        */
  size_t scratch_bytes;
  SfmHostHaloFn host_halo;      /* both or neither; ignored when comm is given   */
  SfmHostAllgatherFn host_allgather;
  void* host_user;
} SfmBandedDesc;
size_t sfm_mesh_banded_scratch_bytes(const SfmBandedDesc* desc);
int sfm_mesh_relax_banded(const SfmBandedDesc* desc, SfmFireState* fire,
                          SfmChunkStats* stats);

/* RCCL transport of those exchanges (resolved at run time; SFM_ERR_NO_DEVICE
 * when no librccl can be loaded).  One communicator per process on the
 * current device; rank 0 creates the id and hands its SFM_COMM_ID_BYTES bytes
 * to the other ranks out of band. */
#define SFM_COMM_ID_BYTES 128
#define SFM_REDUCE_SUM 0
#define SFM_REDUCE_MAX 1
typedef struct SfmComm SfmComm;
int sfm_comm_unique_id(void* id128);
int sfm_comm_init(SfmComm** comm, const void* id128, int rank, int n_ranks);
int sfm_comm_destroy(SfmComm* comm);
/* Sends `count` floats to each existing neighbour and receives as many, as one
 * grouped send/recv (peer = -1: no neighbour on that side). */
int sfm_comm_halo_exchange(SfmComm* comm, int peer_lo, const float* send_lo,
                           float* recv_lo, int peer_hi, const float* send_hi,
                           float* recv_hi, size_t count, void* stream);
/* recv[r * count .. (r + 1) * count) = `send` of rank r. */
int sfm_comm_allgather(SfmComm* comm, const float* send, float* recv, size_t count,
                       void* stream);
/* In-place all-reduce of a few scalars (chunk statistics). */
int sfm_comm_allreduce_scalars(SfmComm* comm, float* inout, size_t count, int op,
                               void* stream);

/* Test hook (no reference counterpart; tests/test_gpu_fft_own.py): batched 1-D complex FFT of
 * `n_pencils` strided pencils through the FFT form's pencil kernel -- pencil p at in + p,
 * its n_in samples at stride n_pencils (float2 each), zero-extended to length n; out
 * [n, n_pencils].  n: any length the hand-written transforms take (radices 2, 3, 5). */
int sfm_debug_fft1d(const void* in, void* out, int n, int n_in, int n_pencils,
                    int inverse, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* SOFIMA_AMD_H_ */

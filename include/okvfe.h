/*
 * okvfe.h -- C ABI of the MI355X (gfx950) vision front-end for OKVIS2.
 *
 * libokvfe.so replaces, for the front-end hot path only, the arithmetic that
 * smartroboticslab/okvis2 reaches through three C++ seams (the reference has
 * no C/FFI boundary of its own; citations are into the reference tree):
 *
 *   (1) cv::FeatureDetector::detect(image, keypoints)
 *         okvis_cv/include/okvis/implementation/Frame.hpp:152, object built at
 *         okvis_frontend/src/Frontend.cpp:2406-2409
 *         (brisk::ScaleSpaceFeatureDetector<HarrisScoreCalculator>(
 *              uniformityRadius, octaves, absoluteThreshold, maxNumKpt))
 *       cv::DescriptorExtractor::compute(image, keypoints, descriptors)
 *         okvis_cv/include/okvis/implementation/Frame.hpp:167, object built at
 *         okvis_frontend/src/Frontend.cpp:2410-2412, configured by
 *         setCameraProperties / setExtractionDirection (Frontend.cpp:239-251)
 *   (2) okvis::Frontend::detectAndDescribe (Frontend.cpp:221-269) and the
 *       brute-force loops of matchStereo (Frontend.cpp:2016-2076),
 *       matchMotionStereo (:1812-1905) and verifyRecognisedPlace (:330-355)
 *   (3) brisk::Hamming::PopcntofXORed(a, b, 3)
 *         (Frontend.cpp:341,1580,1661,1846,2024; FBrisk.cpp:66)
 *
 * Conventions: plain pointers and sizes, caller-allocated outputs, integer
 * status returns, no exceptions across the ABI.  A context is bound to one
 * GPU and is single-threaded (the reference holds one detector/extractor per
 * camera under one mutex per camera, Frontend.cpp:226,2405-2413); different
 * contexts are independent.  "_device" entry points take HIP device pointers
 * and a hipStream_t (as void*) and never synchronise the host: per-call host
 * parameters (gravity, poses) ride through a ring of pinned slots with one
 * asynchronous copy per call.  Stream argument: NULL = the context's own
 * non-blocking stream; OKVFE_STREAM_LEGACY_DEFAULT = the HIP legacy default
 * (null) stream -- the stream torch.cuda.default_stream() denotes, whose raw
 * handle is 0 and therefore cannot be passed as itself; any other value = that
 * hipStream_t.  Work enqueued on one stream is ordered with work on another
 * only by the caller (events), exactly as for any HIP library.  The
 * host-buffer entry points stage through pinned memory and return only when
 * the outputs are written.
 *
 * There is NO CPU fallback: every compute entry point fails with
 * OKVFE_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef OKVFE_H_
#define OKVFE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKVFE_ABI_VERSION 7
#define OKVFE_STREAM_LEGACY_DEFAULT ((void*)(uintptr_t)1) /* = hipStreamLegacy */
#define OKVFE_DESC_BYTES 48 /* okvis_frontend/include/DBoW2/FBrisk.hpp:35 */

typedef enum okvfe_status {
  OKVFE_OK = 0,
  OKVFE_ERR_INVALID_ARGUMENT = 1,
  OKVFE_ERR_NO_DEVICE = 2,
  OKVFE_ERR_OUT_OF_MEMORY = 3,
  OKVFE_ERR_UNSUPPORTED = 4, /* e.g. octaves > 4 */
  OKVFE_ERR_CAPACITY = 5,    /* a caller- or context-sized buffer was too small */
  OKVFE_ERR_DEVICE = 6,      /* HIP runtime error; see okvfe_last_error */
  OKVFE_ERR_NOT_READY = 7    /* e.g. camera-aware extraction without okvfe_set_camera */
} okvfe_status;

/* Layout-compatible with cv::KeyPoint as the reference consumes it
 * (okvis_cv/include/okvis/implementation/Frame.hpp:253-273). */
typedef struct okvfe_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} okvfe_keypoint;

typedef enum okvfe_distortion {
  OKVFE_DIST_NONE = 0,
  OKVFE_DIST_RADTAN = 1,     /* okvis::cameras::RadialTangentialDistortion */
  OKVFE_DIST_EQUIDISTANT = 2 /* okvis::cameras::EquidistantDistortion */
} okvfe_distortion;

/* okvis::cameras::PinholeCamera<DISTORTION_T> intrinsics
 * (okvis_cv/include/okvis/cameras/PinholeCamera.hpp). */
typedef struct okvfe_camera {
  int32_t width, height;
  double fu, fv, cu, cv;
  int32_t distortion; /* okvfe_distortion */
  double d[4];        /* k1 k2 p1 p2 | k1 k2 k3 k4 */
} okvfe_camera;

/* T_WC as rotation (row-major) and translation: p_W = C p_C + r. */
typedef struct okvfe_pose {
  double C[9];
  double r[3];
} okvfe_pose;

/* Detector / extractor / matcher parameters = okvis::FrontendParameters
 * (okvis_common/include/okvis/Parameters.hpp:123-133) plus sizes. */
typedef struct okvfe_config {
  int32_t abi_version;        /* OKVFE_ABI_VERSION */
  int32_t device;             /* HIP device ordinal */
  int32_t width, height;      /* image size, fixed per context */
  int32_t max_batch;          /* images per batch call (>= 1) */
  int32_t num_cameras;        /* camera slots for camera-aware extraction (>= 1) */
  float uniformity_radius;    /* detection_threshold: uniformity radius in px */
  int32_t octaves;            /* 0 = single scale (every shipped config); 1..4 = scale space of
                               * 2*octaves layers (the reference's own smoke test uses 2,
                               * okvis_cv/test/TestFrame.cpp:75-77): every layer may deliver
                               * max_keypoints, so rows per image = 2*octaves*max_keypoints */
  int32_t absolute_threshold; /* Harris noise floor, >= 1 */
  int32_t max_keypoints;      /* max_num_keypoints */
  int32_t rotation_invariant; /* Frontend.cpp:142 default true */
  int32_t scale_invariant;    /* Frontend.cpp:143 default false.  true: the published BRISK scale
                               * ladder -- the pattern of keypoint k is the base pattern scaled to
                               * index okvfe_scale_index(k.size) of 64 (the fixed-scale extractor is
                               * index 17 of the same ladder) */
  int32_t match_threshold;    /* matching_threshold (Hamming bits, strict <) */
  int32_t max_candidates;     /* per-image NMS candidate capacity; 0 = worst case */
  int32_t score_type;         /* OKVFE_SCORE_HARRIS (0): brisk::HarrisScoreCalculator, the x86
                               * reference path and every shipped configuration (Frontend.cpp:2406);
                               * OKVFE_SCORE_AGAST_9_16 (1): the AGAST 9-16 corner score of
                               * brisk::BriskFeatureDetector, the reference's ARM branch
                               * (okvis_cv/test/TestFrame.cpp:71-72), absolute_threshold = its
                               * threshold (34 there); the rest of the detector is shared */
  float box_scale;            /* ABI 7.  Smoothing width of the built-in BRISK2 pattern: every sample's box half-side
                               * = the published one (sigma = 1.3 x the ring's sample spacing) x box_scale; 0 and 1.0 =
                               * the published boxes.  The one parameter of the descriptor that no file of the reference
                               * tree pins (the boxes live in the un-vendored brisk submodule; the vocabulary's statistics
                               * sit best at 1.7 - 2.0 x, tools/pattern/README.md): a named knob instead of an
                               * okvfe_get_pattern / okvfe_set_pattern edit.  Range (0.25, 2.5]; up to 2.05 stays on the
                               * fast descriptor kernels (21 x 21 / 10 x 10 row slots from 1.03 x on).  okvfe_set_pattern
                               * afterwards replaces the whole pattern, this scaling included. */
} okvfe_config;
#define OKVFE_SCORE_HARRIS 0
#define OKVFE_SCORE_AGAST_9_16 1
/* 2: the published BRISK scale-space detector = brisk::BriskFeatureDetector(threshold, octaves)
 * (okvis_cv/test/TestFrame.cpp:71-72): absolute_threshold = the AGAST threshold, octaves as there;
 * AGAST 9-16 scores on the octaves c_i and intra-octaves d_i, the FAST 5-8 score of c0 as the layer
 * below the first octave, maxima over the 3 x 3 neighbourhood and the +-1 px patches of the layers
 * above and below, 2-D sub-pixel fit, and a parabola over the three layers' scores for the CONTINUOUS
 * scale (keypoint.size = 12 x scale, response = the parabola's maximum, octave = layer index; layer nodes
 * 3/4, 1, 3/2 on octaves, 2/3, 1, 4/3 on intra-octaves, 2/3, 1, 3/2 with the result in [0.7, 1.5] on c0).  No
 * uniformity stage (uniformity_radius is ignored); the strongest max_keypoints maxima per layer are
 * kept, (score desc, y, x).  With octaves == 0 it is OKVFE_SCORE_AGAST_9_16.
 * CAVEATS (the brisk sources are not in the reference tree; tools/ref_compare is the check):
 *  - the keypoint POSITION is the 2-D sub-pixel fit on the keypoint's own layer, scaled to image coordinates; the
 *    published detector additionally re-interpolates the position between the fits of the layers above and below
 *    along the fitted scale -- NOT restated here, so x / y may differ from a brisk build by a fraction of a pixel
 *    times the layer's scale while size / response / octave follow the parabola described above;
 *  - the c0 node 2/3 (and the [0.7, 1.5] clamp) is taken from the coefficient matrix quoted in the published
 *    refine1D_2's comment, not from verified code: if the executable coefficients there fit nodes 1/2, 1, 3/2,
 *    sizes and responses of layer-0 keypoints differ. */
#define OKVFE_SCORE_BRISK_SCALESPACE 2

typedef struct okvfe_ctx okvfe_ctx;

/* ---- lifetime ------------------------------------------------------------ */
okvfe_status okvfe_create(const okvfe_config* cfg, okvfe_ctx** out);
void okvfe_destroy(okvfe_ctx* ctx);
/* Message of the last failing call on this context ("" if none). ctx may be
 * NULL to read the message of a failed okvfe_create on this thread. */
const char* okvfe_last_error(const okvfe_ctx* ctx);
int32_t okvfe_abi_version(void);

/* ---- camera-aware extraction setup --------------------------------------- */
/* = BriskDescriptorExtractor::setCameraProperties(rays, imageJacobians, fu)
 * (Frontend.cpp:239-242): host maps, H*W*3 and H*W*6 floats; copied. */
okvfe_status okvfe_set_camera_maps(okvfe_ctx* ctx, int32_t cam, const float* rays_hw3,
                                   const float* jacobians_hw6, float fu);
/* Builds the same maps from intrinsics on the host
 * (= PinholeCamera::initialiseCameraAwarenessMaps, PinholeCamera.hpp:180-208)
 * and uploads them; also stores the intrinsics for back-projection. */
okvfe_status okvfe_set_camera(okvfe_ctx* ctx, int32_t cam, const okvfe_camera* camera);
/* Host helper: fills caller buffers with the awareness maps of a camera. */
okvfe_status okvfe_build_awareness_maps(const okvfe_camera* camera, float* rays_hw3,
                                        float* jacobians_hw6);

/* Host helper: field-of-view overlap of `camera` as seen by `other`
 * (= NCameraSystem::computeOverlaps, okvis_cv/src/NCameraSystem.cpp:48-119; decides which
 * camera pairs matchStereo visits, Frontend.cpp:1998).  R_other_cam = rotation part of
 * T_Cother_C, row-major.  mask_hw (H*W of `camera`, 1 = visible) may be NULL. */
okvfe_status okvfe_camera_overlap(const okvfe_camera* camera, const okvfe_camera* other,
                                  const double R_other_cam[9], uint8_t* mask_hw,
                                  int32_t* has_overlap);

/* ---- detect + describe, host buffers (cv::Feature2D-shaped) -------------- */
/* One image: detect(), compute() and Frame::computeBackProjections in one
 * call.  gravity_C = extraction direction (gravity in the camera frame,
 * Frontend.cpp:247-251); NULL or cam < 0 selects the non-camera-aware mode.
 * keypoints/descriptors: capacity `cap` rows; backproj (cap*3 doubles) and
 * backproj_valid (cap bytes) may be NULL. */
okvfe_status okvfe_detect_describe(okvfe_ctx* ctx, const uint8_t* image, size_t stride,
                                   int32_t cam, const float gravity_C[3],
                                   okvfe_keypoint* keypoints, uint8_t* descriptors,
                                   double* backproj, uint8_t* backproj_valid, int32_t cap,
                                   int32_t* n_out);
/* detect() only (no descriptor-stage removal): keypoints in acceptance order. */
okvfe_status okvfe_detect(okvfe_ctx* ctx, const uint8_t* image, size_t stride,
                          okvfe_keypoint* keypoints, int32_t cap, int32_t* n_out);

/* cv::FeatureDetector::detect when the SAME image goes to cv::DescriptorExtractor::compute right after
 * it, as okvis::Frame::detect() / Frame::describe() do (okvis_cv/include/okvis/implementation/
 * Frame.hpp:152,167; the extraction direction is set before both, Frontend.cpp:246-251): returns what
 * okvfe_detect returns, but runs the whole detect + describe chain for (cam, gravity_C) on the one
 * upload and keeps the result.  An okvfe_compute that then asks for exactly this -- same image pointer,
 * stride and pixel content, same cam / gravity_C, the keypoints unchanged -- is answered from the kept
 * result without touching the GPU; any other okvfe_compute runs as usual.  One image upload and one
 * synchronisation per frame instead of two. */
okvfe_status okvfe_detect_ahead(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                                const float gravity_C[3], okvfe_keypoint* keypoints, int32_t cap, int32_t* n_out);

/* compute() only = cv::DescriptorExtractor::compute(image, keypoints, descriptors)
 * (Frame.hpp:167): describes the n_in caller keypoints (in/out: the extractor
 * removes keypoints too close to the rim, order preserved) and back-projects
 * the survivors.  n_in <= max_keypoints. */
okvfe_status okvfe_compute(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                           const float gravity_C[3], okvfe_keypoint* keypoints, int32_t n_in,
                           uint8_t* descriptors, double* backproj, uint8_t* backproj_valid,
                           int32_t* n_out);

/* ---- detect + describe, device-resident batches -------------------------- */
/* images_dev: n_images contiguous H*W u8 images in HBM.  cam_ids /
 * gravity_C (n_images*3) are HOST arrays (NULL = not camera aware).
 * Results stay in the context's device buffers (okvfe_get_device_outputs). */
okvfe_status okvfe_detect_describe_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                                int32_t n_images, const int32_t* cam_ids,
                                                const float* gravity_C, void* stream);

/* Host-fed form of the batch call: images_host = n_images contiguous H*W u8 images in HOST
 * memory, as frames arrive in the reference (cv::Mat handed to the multiframe,
 * okvis_multisensor_processing/src/ThreadedSlam.cpp:247-265).  The context owns two device image
 * buffers and a copy stream: the PCIe copy of this batch is enqueued at once and the kernels wait
 * for it through an event, so with pinned host memory (hipHostMalloc / cudaHostRegister; a pageable
 * source makes the runtime stage the copy synchronously) the call returns immediately and the copy
 * of batch k+1 runs under the kernels of batch k.  images_host must stay valid until the work
 * enqueued on `stream` by this call has completed.  Results as okvfe_detect_describe_batch_device. */
okvfe_status okvfe_detect_describe_batch_host(okvfe_ctx* ctx, const uint8_t* images_host,
                                              int32_t n_images, const int32_t* cam_ids,
                                              const float* gravity_C, void* stream);

typedef struct okvfe_device_outputs {
  int32_t max_keypoints;         /* row capacity per image (= max_keypoints * layers) */
  const int32_t* counts;         /* [max_batch] kept keypoints per image */
  const okvfe_keypoint* keypoints; /* [max_batch][max_keypoints] */
  const uint8_t* descriptors;    /* [max_batch][max_keypoints][48] */
  const double* backproj;        /* [max_batch][max_keypoints][3] */
  const uint8_t* backproj_valid; /* [max_batch][max_keypoints] */
  const int32_t* scores;         /* [max_batch][H][score_pitch] score maps of the last detect call (layer 0);
                                  * pixel (x, y) at y * score_pitch + okvfe_score_column(ctx, x).  NULL when
                                  * that call wrote no map: see okvfe_set_keep_score_map */
  const int32_t* detect_counts;  /* [max_batch] keypoints before descriptor-stage removal */
  const int32_t* candidate_counts; /* [max_batch] NMS maxima found (may exceed capacity) */
  int32_t score_pitch;           /* ints per score-map row (>= W: the fused score+NMS kernel pads rows so that
                                  * every wave stores whole 128-byte lines) */
  int32_t score_strips;          /* 0 / 1: dense rows; >= 2: column x sits at okvfe_score_column(ctx, x) */
} okvfe_device_outputs;
okvfe_status okvfe_get_device_outputs(okvfe_ctx* ctx, okvfe_device_outputs* out);
/* Column of pixel x within a row of okvfe_device_outputs.scores (x itself for dense maps).  For a
 * dense copy of the score map use okvfe_harris_score_device. */
int32_t okvfe_score_column(const okvfe_ctx* ctx, int32_t x);
/* Single-scale Harris detection keeps NO score map by default (ABI 5): HarrisScoreCalculator's map
 * (behind cv::FeatureDetector::detect, Frame.hpp:152) is only ever read at the maxima, so the fused
 * score + NMS kernel writes the candidates alone -- one of its five bytes per pixel -- and the selection
 * recomputes the nine sub-pixel scores of the keypoints it keeps from the image, bit-identical.  keep = 1
 * makes the following detect calls of this context write the map again (okvfe_device_outputs.scores);
 * scale-space and AGAST contexts always keep theirs. */
okvfe_status okvfe_set_keep_score_map(okvfe_ctx* ctx, int32_t keep);
/* Lanes inside one call (ABI 7).  okvfe_detect_describe_batch_device / _host cut a batch into `lanes` slices of whole
 * stereo pairs and run each slice's kernel chain on a stream of the context's own, joined back onto the caller's
 * stream before the call returns (still without a host synchronisation): the vector-ALU-bound score kernel of one
 * slice runs under the LDS- and latency-bound selection / descriptor kernels of another -- what a caller used to get
 * only by splitting the batch over several contexts and streams (the camera-parallel shape of
 * okvis_multisensor_processing/src/ThreadedSlam.cpp:434-448, inside the call).  Results are those of the unsplit
 * call, byte for byte (every kernel works per image).  lanes = 0: the library's choice -- at present not to cut: with
 * ONE caller stream every call ends in a join, the slices run in phase and measure 1-5 % slower than the unsplit call,
 * while several contexts on several streams (lanes that drift out of phase across calls) gain 8 %: DESIGN.md (e); 1: off;
 * 2..8: that many.  Single-scale contexts only; others ignore it.
 * NEGATIVE, -2 .. -8: PIPELINED lanes.  The call does not join its lanes onto the caller's stream at all, and an
 * okvfe_match_stereo_batch_device that follows matches each slice's pairs on that slice's lane stream (every pair inside
 * one slice, the slices' pairs in contiguous runs -- what a batch of stereo pairs (2i, 2i + 1) is).  Lane l then starts
 * the NEXT call's score kernel behind its own previous work instead of behind everybody's: the lanes drift out of phase
 * and stay there, which is what makes several contexts on several streams faster than one.  The price is the contract:
 * after such a call returns, work the CALLER queues on its stream is NOT ordered behind the results.  Any other entry
 * point of the context joins first (stream-taking ones make their stream wait, host-side readers synchronise), and
 * okvfe_lanes_join(ctx, stream) does it explicitly; the caller must join before it overwrites the input images or reads
 * okvfe_device_outputs / the match rows with kernels of its own.  Results are the unsplit call's, byte for byte. */
okvfe_status okvfe_set_internal_lanes(okvfe_ctx* ctx, int32_t lanes);
/* Makes `stream` (NULL: the context's own) wait for every pipelined lane of the context; no-op when none is pending. */
okvfe_status okvfe_lanes_join(okvfe_ctx* ctx, void* stream);

/* Order of every 3-term FP64 sum in the matchers' gate chain (dot products, norms, C * v and C^T * v:
 * stereo_triangulation.cpp:62-76, Frontend.cpp:2027-2073 evaluate them through Eigen):
 *   OKVFE_SUM3_EIGEN_TREE (default, ABI 6)  x0 + (x1 + x2) -- Eigen's unrolled non-vectorised reduction of a
 *                                           fixed-size Vector3d (Redux.h, split at Length / 2), which is what a
 *                                           stock build takes because Vector3d is not packet-aligned;
 *   OKVFE_SUM3_LEFT_TO_RIGHT                (x0 + x1) + x2 -- the order ABI <= 5 used.
 * Affects hp_W / quality / back-projection-derived values by <= 1 ulp per sum (decisions rarely flip, bytes do).
 * Neither order can be confirmed against the reference in this tree (no Eigen); tools/ref_compare decides it
 * on a machine with the reference built.  The setting lives on the context's DEVICE: it applies to every
 * context of this process on that device, and the call synchronises the device. */
#define OKVFE_SUM3_LEFT_TO_RIGHT 0
#define OKVFE_SUM3_EIGEN_TREE 1
okvfe_status okvfe_set_fp64_reduction(okvfe_ctx* ctx, int32_t order);

/* Several contexts fed in turn from several host threads / streams on ONE GPU (a camera per context,
 * ThreadedSlam.cpp:434-448): mode 1 runs the score kernels of all contexts of the process on a device
 * one after the other in enqueue order while everything downstream of them overlaps freely, so the
 * VALU-bound score kernel of one context runs beside the latency-bound selection / matching of
 * another instead of beside another score kernel; mode 2 also chains the descriptor kernels; 0 (the
 * default) = off.  Process-wide, takes effect for calls enqueued afterwards.  (The library reads no
 * environment variable.) */
okvfe_status okvfe_set_heavy_kernel_chaining(int32_t mode);

/* Scale index of the scale-invariant extractor (brisk::BriskDescriptorExtractor(rotInv, scaleInv =
 * true), Frontend.cpp:2410-2412) for a keypoint of diameter `size`, published BRISK:
 * max(int(64 / lb(30) * lb(size / (0.6 * 12)) + 0.5), 0), at most 63.  Host-only, no context. */
int32_t okvfe_scale_index(float keypoint_size);

/* NMS candidate capacity check of the last batch (synchronises; one small copy): an image whose
 * score map had more maxima than the context's candidate capacity (okvfe_config.max_candidates)
 * keeps NO keypoints -- which maxima an overflowing list drops would depend on the order of the
 * atomics -- and makes this call fail with OKVFE_ERR_CAPACITY (*first_overflowed = its index, -1
 * if none; may be NULL).  Device-resident pipelines (batch detect -> match / gather) call this
 * once per batch or once per sequence, as their budget allows. */
okvfe_status okvfe_check_capacity(okvfe_ctx* ctx, int32_t n_images, int32_t* first_overflowed);

/* Copies image `index` of the last batch to host buffers (synchronises). */
okvfe_status okvfe_download_image_result(okvfe_ctx* ctx, int32_t index, okvfe_keypoint* keypoints,
                                         uint8_t* descriptors, double* backproj,
                                         uint8_t* backproj_valid, int32_t cap, int32_t* n_out);

/* The batch call in two halves, = Frame::detect / Frame::describe of the reference
 * (okvis_cv/include/okvis/implementation/Frame.hpp:140-154, 160-175): okvfe_detect_batch_device
 * leaves the detected keypoints in the context, okvfe_describe_batch_device (same images, same
 * n_images) extracts descriptors, compacts and back-projects.  Splitting lets a caller with
 * several contexts / streams enqueue detect for all of them before describe for all of them.
 *
 * okvfe_set_heavy_kernel_chaining(1) makes the score kernels of all contexts of the process (per
 * device) run one after the other, in enqueue order, through events; 2 chains the describe
 * kernels as well.  Contexts fed in turn from different streams then run out of phase, so the
 * latency-bound stages of one overlap the throughput-bound ones of another (bench.py --lanes). */
okvfe_status okvfe_detect_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                       int32_t n_images, void* stream);
okvfe_status okvfe_describe_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                         int32_t n_images, const int32_t* cam_ids,
                                         const float* gravity_C, void* stream);

/* ---- single stages on device buffers (parity tests, profiling) ----------- */
/* K1: score maps of the configured score_type (Harris unless the context was created with
 * OKVFE_SCORE_AGAST_9_16), n_images * H * W int32. */
okvfe_status okvfe_harris_score_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                       int32_t n_images, int32_t* scores_dev, void* stream);
/* Diagnostic: the byte mover of the fused score + NMS kernel -- the same loads and the same stores
 * on the context's score-map layout with no arithmetic in between (the score maps receive pixel
 * bytes; no candidates are produced).  Its duration is that kernel's own memory floor; bench.py
 * times it beside the kernel (roofline.byte_mover_ms).  OKVFE_ERR_UNSUPPORTED where the fused kernel
 * does not apply (AGAST score types, widths that are no multiple of 4, octaves > 0). */
okvfe_status okvfe_harris_byte_mover_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                            int32_t n_images, void* stream);

/* ---- stage profiling ----------------------------------------------------- */
/* When enabled, every stage launch of the batch entry points is bracketed by a
 * HIP event pair on the launch stream (no synchronisation is added).
 * okvfe_profile_read synchronises and returns, per stage, the summed elapsed
 * milliseconds and the number of launches since okvfe_profile_enable. */
typedef enum okvfe_stage {
  OKVFE_STAGE_HARRIS = 0,   /* K1 score map */
  OKVFE_STAGE_NMS = 1,      /* K2 */
  OKVFE_STAGE_SORT = 2,     /* K3 sort */
  OKVFE_STAGE_SELECT = 3,   /* K3 uniformity + K4 sub-pixel */
  OKVFE_STAGE_MAP = 4,      /* map matchers on device blocks: matchToMap / uninitialised / verifyRecognisedPlace */
  OKVFE_STAGE_DESCRIBE = 5, /* K6 */
  OKVFE_STAGE_COMPACT = 6,  /* compaction + back-projection */
  OKVFE_STAGE_MATCH = 7,    /* K7 gated stereo match */
  OKVFE_STAGE_COUNT = 8
} okvfe_stage;
/* enable: 0 = off, 1 = every stage, or an OR of OKVFE_PROFILE_STAGE(stage) to time only some
 * stages (each timed launch costs two event records, i.e. two barrier packets on the stream). */
#define OKVFE_PROFILE_STAGE(stage) (1 << (8 + (stage)))
okvfe_status okvfe_profile_enable(okvfe_ctx* ctx, int32_t enable);
okvfe_status okvfe_profile_read(okvfe_ctx* ctx, double total_ms[OKVFE_STAGE_COUNT],
                                int32_t launches[OKVFE_STAGE_COUNT]);

/* ---- matching ------------------------------------------------------------ */
typedef struct okvfe_stereo_match {
  int32_t k1;            /* index in image 1, -1 = no match */
  int32_t dist;          /* Hamming distance of the match (match_threshold if none) */
  int32_t initialisable; /* !isParallel */
  int32_t pad;
  double hp_W[4];        /* triangulated homogeneous point, world frame */
} okvfe_stereo_match;

/* One (im0, im1) pair of the last batch, = the k0 x k1 loop of
 * Frontend::matchStereo (Frontend.cpp:2016-2076). */
typedef struct okvfe_stereo_pair {
  int32_t image0, image1; /* indices into the last batch */
  okvfe_pose T_WC0, T_WC1;
  double f0, f1;          /* 0.5*(fu+fv) of each camera */
} okvfe_stereo_pair;

/* pairs: HOST array.  matches_dev: [n_pairs][max_keypoints] device rows. */
okvfe_status okvfe_match_stereo_batch_device(okvfe_ctx* ctx, const okvfe_stereo_pair* pairs,
                                             int32_t n_pairs, okvfe_stereo_match* matches_dev,
                                             void* stream);
/* Host-buffer form on explicit descriptor sets (no context batch needed). */
okvfe_status okvfe_match_stereo(okvfe_ctx* ctx, const uint8_t* desc0, const okvfe_keypoint* kp0,
                                const double* backproj0, const uint8_t* valid0, int32_t n0,
                                const uint8_t* desc1, const okvfe_keypoint* kp1,
                                const double* backproj1, const uint8_t* valid1, int32_t n1,
                                const okvfe_pose* T_WC0, const okvfe_pose* T_WC1, double f0,
                                double f1, okvfe_stereo_match* matches /* n0 */);

typedef struct okvfe_motion_match {
  int32_t k1;            /* index in the current frame, -1 = no match */
  int32_t dist;
  int32_t initialisable; /* !isParallel */
  int32_t accepted;      /* winner re-projects within 4 px (Frontend.cpp:1897-1905) */
  double cos_quality;    /* quality = acos(cos_quality) (Frontend.cpp:1887-1889) */
  double hp_W[4];
} okvfe_motion_match;

/* = the k0 x k1 loop of Frontend::matchMotionStereo for one (older frame, current frame, camera)
 * triple (Frontend.cpp:1789-1905).  Frame 0 = older frame.  skip0[k0] != 0: k0 is not matched
 * (landmark already initialised / already observed, :1814-1841); matched1[k1] != 0: current
 * keypoint already carries a landmark and is left out (:1795-1798).  Either may be NULL.
 * Host buffers; `camera` is the shared camera of both frames. */
okvfe_status okvfe_match_motion_stereo(okvfe_ctx* ctx, const okvfe_camera* camera,
                                       const uint8_t* desc0, const okvfe_keypoint* kp0,
                                       const double* backproj0, const uint8_t* valid0,
                                       const uint8_t* skip0, int32_t n0, const uint8_t* desc1,
                                       const okvfe_keypoint* kp1, const double* backproj1,
                                       const uint8_t* valid1, const uint8_t* matched1, int32_t n1,
                                       const okvfe_pose* T_WC0, const okvfe_pose* T_WC1,
                                       okvfe_motion_match* matches /* n0 */);

/* = Frontend::matchToMapByThread for the 3-D landmarks (Frontend.cpp:1552-1589), all keypoints
 * in one call.  Landmarks in the caller's order (ascending LandmarkId in the reference);
 * landmark l owns pool rows desc_begin[l] .. desc_begin[l+1]-1 (<= 3 descriptors each,
 * :1220-1222) and the projection (2 doubles).  use[k] == 0 skips keypoint k (:1541-1547).
 * reprojection_threshold: 20 px with IMU, 150 without (:1530).  Outputs per keypoint: landmark
 * INDEX (-1 = none) and distance (match_threshold if none). */
okvfe_status okvfe_match_to_map(okvfe_ctx* ctx, const uint8_t* desc, const okvfe_keypoint* kps,
                                const uint8_t* use, int32_t n_kps, const double* projections_l2,
                                const int32_t* desc_begin /* n_landmarks + 1 */,
                                int32_t n_landmarks, const uint8_t* pool,
                                double reprojection_threshold, int32_t* best_landmark,
                                int32_t* best_dist);

/* ---- matchToMap from the raw landmark table (no host-side pooling) ---------------------------
 * = Frontend::matchToMap up to and including its first matcher pass (Frontend.cpp:1219-1411):
 * every landmark is projected into the current camera (FoV check, :1232-1256), its observations
 * are scored and pooled by the three-slot "best views" buffer exactly as written at :1262-1354,
 * and the 3-D landmarks are matched against the frame (matchToMapByThread, :1552-1589) -- all on
 * the device, the pooled set never visits the host.
 * Landmarks in ascending LandmarkId order (the order of the reference's std::map); landmark l owns
 * observations obs_begin[l] .. obs_begin[l+1]-1, listed in the reference's iteration order
 * (observations.rbegin() -> rend()). */
typedef struct okvfe_landmark_table {
  int32_t n_landmarks, n_observations, n_poses;
  const double* hp_W;         /* n_landmarks x 4 homogeneous points (MapPoint::point) */
  const double* quality;      /* n_landmarks (MapPoint::quality) */
  const int32_t* obs_begin;   /* n_landmarks + 1 */
  const int32_t* obs_pose;    /* n_observations: index into poses */
  const uint8_t* obs_desc;    /* n_observations x 48: keypointDescriptor of the observing keypoint */
  const double* obs_backproj; /* n_observations x 3: its cached back-projection (not normalised) */
  const okvfe_pose* poses;    /* n_poses: T_WC of every observing (frame, camera) */
} okvfe_landmark_table;

/* What the pooling left per landmark (caller-allocated, n_landmarks rows; the struct pointer may
 * be NULL): status 0 = not matched against (outside the FoV or no pooled view), 1 = 3-D, 2 = not
 * 3-D yet (LandmarkToMatch::is3d); n_desc = pooled descriptors (0..2); obs_rows[3l + r] =
 * observation whose descriptor is pooled row r (-1 = none; row 2 may be written but is cropped as
 * in the reference); projection (2); e_W / r_W: 2 x 3 doubles, observing unit ray and camera
 * centre per pooled row (LandmarkToMatch::e_W / r_W) -- together the inputs of
 * okvfe_match_to_map_uninitialised for the status-2 landmarks. */
typedef struct okvfe_landmark_pool {
  int32_t* status;
  int32_t* n_desc;
  int32_t* obs_rows;
  double* projection;
  double* e_W;
  double* r_W;
} okvfe_landmark_pool;

/* cam = camera slot with intrinsics (okvfe_set_camera); exclusive != 0 =
 * loopClosureLandmarksToUseExclusively (view-point / scale pruning off, :1293-1303).  Outputs per
 * keypoint as okvfe_match_to_map: landmark index (into the table, -1 = none) and distance. */
okvfe_status okvfe_match_to_map_landmarks(okvfe_ctx* ctx, int32_t cam, const okvfe_landmark_table* table,
                                          const okvfe_pose* T_WC1, double reprojection_threshold,
                                          int32_t exclusive, const uint8_t* desc,
                                          const okvfe_keypoint* kps, const uint8_t* use, int32_t n_kps,
                                          okvfe_landmark_pool* pool_out, int32_t* best_landmark,
                                          int32_t* best_dist);

/* = Frontend::matchToMapByThreadUnitialised (Frontend.cpp:1616-1719), all keypoints in one call:
 * landmarks that are not 3-D yet.  Pool row d carries its observing unit ray e0_W[d] and camera
 * centre r0_W[d] (3 doubles each, LandmarkToMatch::e_W / r_W).  backproj = the current frame's
 * cached back-projections (normalised inside, :1625); use[k] != 0 as computed at :1621-1635;
 * previous_landmark[k] = index of the landmark keypoint k already carries or -1 (:1701-1704).
 * Outputs: landmark index / distance per keypoint, hps_W (4 doubles, written when hp_set[k]:
 * only non-parallel triangulations are stored, :1708-1710) and the count of already-correct
 * matches (ctrs, :1702). */
okvfe_status okvfe_match_to_map_uninitialised(okvfe_ctx* ctx, const uint8_t* desc,
                                              const double* backproj, const uint8_t* use,
                                              const int32_t* previous_landmark, int32_t n_kps,
                                              const int32_t* desc_begin, int32_t n_landmarks,
                                              const uint8_t* pool, const double* e0_W,
                                              const double* r0_W, const okvfe_pose* T_WC1,
                                              double focal_length, int32_t* best_landmark,
                                              int32_t* best_dist, double* hps_W, uint8_t* hp_set,
                                              int32_t* already_matched);

typedef struct okvfe_candidate {
  int32_t i, j, dist;
} okvfe_candidate;
/* All (i, j) with popcnt(A[i]^B[j]) < threshold, ordered by (i, j); host buffers.
 * n_out receives the total found even when it exceeds cap (then OKVFE_ERR_CAPACITY). */
okvfe_status okvfe_hamming_candidates(okvfe_ctx* ctx, const uint8_t* A, int32_t nA,
                                      const uint8_t* B, int32_t nB, int32_t threshold,
                                      okvfe_candidate* out, int32_t cap, int32_t* n_out);
/* Per row of A the first-lowest j with minimal distance < threshold
 * (= the loop of verifyRecognisedPlace, Frontend.cpp:337-346); -1 if none. */
okvfe_status okvfe_hamming_argmin(okvfe_ctx* ctx, const uint8_t* A, int32_t nA, const uint8_t* B,
                                  int32_t nB, uint32_t threshold, int32_t* best_j,
                                  uint32_t* best_dist);

/* = the descriptor matching of Frontend::verifyRecognisedPlace for ALL old landmarks against one
 * camera of the current frame in one launch (Frontend.cpp:330-355): landmark l owns rows
 * desc_begin[l] .. desc_begin[l+1]-1 of landmark_desc (its descriptors in the insertion order of
 * :318-326); per landmark the running minimum over (descriptor, k) with strict <.  k_min[l] = 0 and
 * dist_min[l] = match_threshold when nothing is below the threshold ("distMin < threshold" fails). */
okvfe_status okvfe_verify_place_match(okvfe_ctx* ctx, const uint8_t* landmark_desc,
                                      const int32_t* desc_begin /* n_landmarks + 1 */,
                                      int32_t n_landmarks, const uint8_t* frame_desc, int32_t n_kps,
                                      int32_t* k_min, uint32_t* dist_min);

/* = DBoW2::TemplatedVocabulary<FBrisk::TDescriptor, FBrisk>::transform for n features (the
 * quantisation behind dBow_->database.add / query, Frontend.cpp:756-766): descend from the root
 * (node 0), at every level the child with the smallest FBrisk::distance (FBrisk.cpp:64-67; first
 * child on ties), down to a leaf.  Tree in arrays: node i's children are
 * child_index[child_begin[i] .. child_begin[i+1]); node_word[i] = word id of a leaf, < 0 for an
 * inner node; node descriptors n_nodes x 48 (the root's row is unused).  Nodes must be numbered
 * parents-first (as in DBoW2 files).  Outputs per feature: word id and (optional) leaf node. */
okvfe_status okvfe_fbrisk_transform(okvfe_ctx* ctx, const uint8_t* descriptors, int32_t n,
                                    const uint8_t* node_descriptors, int32_t n_nodes,
                                    const int32_t* child_begin, const int32_t* child_index,
                                    const int32_t* node_word, int32_t* word_ids, int32_t* leaf_nodes);

/* = the weighting / normalisation half of DBoW2::TemplatedVocabulary::transform(features, BowVector)
 * (behind dBow_->database.add / query, Frontend.cpp:756-766), for the word ids okvfe_fbrisk_transform
 * returned: TF_IDF / TF (weighting 0 / 1) sum the word's stored weight per occurrence in feature
 * order, IDF / BINARY (2 / 3) keep the first; words of weight <= 0 are skipped; with
 * normalise_l1 != 0 (the L1 scoring of the shipped vocabulary) the vector is divided by its L1 norm
 * (summed in ascending word order), otherwise TF_IDF / TF divide by the number of distinct words.
 * Host helper (a few thousand features at most).  Output: ascending word ids + values; n_out is the
 * number of distinct words even when it exceeds cap (then OKVFE_ERR_CAPACITY). */
okvfe_status okvfe_bow_vector(const int32_t* word_ids, int32_t n_features, const double* word_weight,
                              int32_t n_words, int32_t weighting, int32_t normalise_l1,
                              int32_t* ids_out, double* values_out, int32_t cap, int32_t* n_out);
/* = DBoW2::TemplatedDatabase::query with L1 scoring (queryL1) against ALL stored entries in one
 * launch: entry e owns db_ids / db_values [db_begin[e], db_begin[e+1]) (ascending word ids, the
 * BowVector it was added with).  Per entry, over the common words in ascending word order:
 * value += |q - d| - |q| - |d|, score = -value / 2 (1 = identical, 0 = nothing in common) -- the
 * same additions in the same order as the inverted-file walk of the reference, so the doubles are
 * bit-identical.  scores[e] = -1 for entries without a common word (DBoW2 does not list them).
 * The reference then sorts the listed entries by id (Frontend.cpp:760-765): this array is already
 * in id order. */
okvfe_status okvfe_bow_query_l1(okvfe_ctx* ctx, const int32_t* db_begin /* n_entries + 1 */,
                                const int32_t* db_ids, const double* db_values, int32_t n_entries,
                                const int32_t* q_ids, const double* q_values, int32_t n_q,
                                double* scores /* n_entries */);

/* = brisk::Hamming::PopcntofXORed(a, b, n128); host, no context. */
uint32_t okvfe_popcnt_xor(const uint8_t* a, const uint8_t* b, int32_t n128);

/* ---- data formats either side of the path (host, no context) ------------- */
/* Text records of OKVIS2 map files, one line per keypoint:
 *   "FRAME:KEYPOINT <stateId> <cameraIdx> <x> <y> <size> BRISK2 <96 hex digits>\n"
 * (okvis::Component::save, okvis_ceres/src/Component.cpp:448-460, precision 17 from :409).
 * out == NULL queries the size; *written always receives the bytes needed. */
okvfe_status okvfe_format_keypoint_lines(uint64_t state_id, uint64_t camera_idx,
                                         const okvfe_keypoint* keypoints,
                                         const uint8_t* descriptors, int32_t n, char* out,
                                         size_t cap, size_t* written);
/* Reads one block of such lines (same stateId / cameraIdx; stops at the first other line), as
 * okvis::Component::load does before MultiFrame::resetKeypoints / resetDescriptors
 * (Component.cpp:235-266).  Descriptor kinds other than BRISK2 -> OKVFE_ERR_UNSUPPORTED. */
okvfe_status okvfe_parse_keypoint_lines(const char* text, size_t len, uint64_t* state_id,
                                        uint64_t* camera_idx, okvfe_keypoint* keypoints,
                                        uint8_t* descriptors, int32_t cap, int32_t* n_out,
                                        size_t* consumed);
/* = DBoW2::FBrisk::meanValue (okvis_frontend/src/FBrisk.cpp:25-58): bitwise majority of n
 * 48-byte descriptors (bit set iff more than n/2 descriptors have it). */
okvfe_status okvfe_fbrisk_mean(const uint8_t* descriptors, int32_t n, uint8_t* mean48);

/* ---- cross-camera gather block (multi-GPU) -------------- */
/* Fixed-size per-image record for the RCCL all-gather: {count, keypoints,
 * descriptors, back-projections, valid flags}; size depends only on
 * max_keypoints. */
size_t okvfe_gather_block_bytes(const okvfe_ctx* ctx);
/* Packs images first_index .. first_index+n-1 of the last batch into n contiguous blocks
 * (stride okvfe_gather_block_bytes) with one kernel. */
okvfe_status okvfe_pack_gather_blocks_device(okvfe_ctx* ctx, int32_t first_index, int32_t n,
                                             void* blocks_dev, void* stream);
/* Matches frame f of blocks0 with frame f of blocks1 for f < n_frames in one launch (both arrays
 * contiguous with the block stride); matches_dev: [n_frames][max_keypoints]. */
okvfe_status okvfe_match_stereo_blocks_batch_device(okvfe_ctx* ctx, const void* blocks0_dev,
                                                    const void* blocks1_dev, int32_t n_frames,
                                                    const okvfe_pose* T_WC0,
                                                    const okvfe_pose* T_WC1, double f0, double f1,
                                                    okvfe_stereo_match* matches_dev, void* stream);
/* Packs image `index` of the last batch into block_dev (device). */
okvfe_status okvfe_pack_gather_block_device(okvfe_ctx* ctx, int32_t index, void* block_dev,
                                            void* stream);
/* Device-resident variant of okvfe_match_motion_stereo: block0 = older frame, block1 = current
 * frame, both gather blocks of camera slot `cam` (intrinsics from okvfe_set_camera; frame size =
 * the context's).  skip0_dev / matched1_dev: device flag arrays [max_keypoints] or NULL.
 * matches_dev: [max_keypoints] okvfe_motion_match, rows >= the block's keypoint count untouched.
 * Nothing crosses PCIe but the two poses. */
okvfe_status okvfe_match_motion_stereo_blocks_device(okvfe_ctx* ctx, int32_t cam,
                                                     const void* block0_dev,
                                                     const void* block1_dev,
                                                     const uint8_t* skip0_dev,
                                                     const uint8_t* matched1_dev,
                                                     const okvfe_pose* T_WC0,
                                                     const okvfe_pose* T_WC1,
                                                     okvfe_motion_match* matches_dev, void* stream);
/* Matches two gathered blocks (device), as okvfe_match_stereo_batch_device. */
okvfe_status okvfe_match_stereo_blocks_device(okvfe_ctx* ctx, const void* block0_dev,
                                              const void* block1_dev, const okvfe_pose* T_WC0,
                                              const okvfe_pose* T_WC1, double f0, double f1,
                                              okvfe_stereo_match* matches_dev, void* stream);

/* ---- sampling pattern as data ------------------------------------------------ */
/* The extractor's sampling pattern is DATA.  The pattern the library builds at okvfe_create carries
 * BRISK2's pair table and bit order as recovered from the 819 real BRISK2 descriptors of the
 * reference's vocabulary (66 sample points, 384 live bits; tools/pattern/README.md, INTEGRATION.md
 * section 0) on the published BRISK ring radii and smoothing widths, which the reference tree cannot
 * confirm (its `brisk` submodule is empty).  A pattern dumped from a real brisk build -- sample
 * offsets, smoothing half-widths, the short pairs IN BIT ORDER, the long pairs of the orientation
 * estimate -- is installed with okvfe_set_pattern and replaces it without touching a kernel.
 * Limits of the kernels: <= 72 sample points, <= 384 short pairs (bit b of the 48-byte row =
 * value[short_i[b]] > value[short_j[b]]; unused bits stay 0), <= 1100 long pairs, border >= the farthest
 * sample + its half-width + 1.  okvfe_set_pattern synchronises the context.
 * Which descriptor kernel serves a pattern (okvfe_pattern_kernel_class): a wave has 64 lanes, and lane l samples
 * point (n_points - 64) + l; the FIRST n_points - 64 points of a pattern with more than 64 are the extra samples
 * (evaluated beside the wave), so put the smallest boxes first -- the built-in pattern has the centre and one
 * hexagon point there.
 *   class 0: extra samples' half-width <= 2.0 (5 x 5 row slots), all others <= 4.75 (11 x 11): the fast kernels;
 *   class 1: <= 4.25 / <= 9.75 (10 x 10 / 21 x 21 row slots): the WIDE instantiations of the same kernels;
 *   class 2: wider still, or any half-width below 0.5 (bilinear point samples): the all-modes kernel with plain
 *            box loops -- correct for every pattern, several times slower. */
#define OKVFE_PATTERN_POINTS 72
#define OKVFE_PATTERN_SHORT_PAIRS 384
#define OKVFE_PATTERN_LONG_PAIRS 1100
typedef struct okvfe_pattern {
  int32_t n_points;
  float px[OKVFE_PATTERN_POINTS], py[OKVFE_PATTERN_POINTS]; /* offsets from the keypoint, upright, px */
  float sigma_half[OKVFE_PATTERN_POINTS];                   /* half side of the smoothing box */
  int32_t n_short;
  uint8_t short_i[OKVFE_PATTERN_SHORT_PAIRS], short_j[OKVFE_PATTERN_SHORT_PAIRS];
  int32_t n_long;
  uint8_t long_i[OKVFE_PATTERN_LONG_PAIRS], long_j[OKVFE_PATTERN_LONG_PAIRS];
  int32_t long_wdx[OKVFE_PATTERN_LONG_PAIRS], long_wdy[OKVFE_PATTERN_LONG_PAIRS]; /* round(2048 d / |d|^2) */
  int32_t border; /* keypoints closer than this to the image rim are removed */
} okvfe_pattern;
okvfe_status okvfe_get_pattern(const okvfe_ctx* ctx, okvfe_pattern* out);
okvfe_status okvfe_set_pattern(okvfe_ctx* ctx, const okvfe_pattern* pattern);
/* 0 / 1 / 2 as above for the pattern installed now (after okvfe_create's box_scale or okvfe_set_pattern); -1: ctx NULL */
int32_t okvfe_pattern_kernel_class(const okvfe_ctx* ctx);

/* ---- device-resident, batched map matchers ---------------------------------- */
/* The map-side loops of the front-end on data that never leaves the GPU: frame f of the batch is
 * gather block f (okvfe_pack_gather_blocks_device: keypoints, descriptors, back-projections and the
 * keypoint count of one image), the pooled landmark set lives in device memory, ONE launch serves
 * all frames, nothing synchronises the host.  Row capacity of every per-keypoint array = the
 * context's okvfe_device_outputs.max_keypoints (K); rows >= a frame's keypoint count are untouched.
 * The pooled set is what okvfe_match_to_map / okvfe_match_to_map_uninitialised take, as device
 * pointers (e.g. uploaded once per keyframe with okvfe_copy_to_device). */
typedef struct okvfe_map_device {
  int32_t n_landmarks;        /* L */
  const int32_t* desc_begin;  /* device, L + 1: landmark l owns pool rows desc_begin[l] .. desc_begin[l+1]-1 */
  const uint8_t* pool;        /* device, desc_begin[L] x 48 pooled descriptors */
  const double* projections;  /* device, n_frames x L x 2 (frame-major): 3-D landmarks projected into frame f
                               * (Frontend.cpp:1232-1256); only okvfe_match_to_map_blocks_device reads it */
  const double* e0_W;         /* device, desc_begin[L] x 3: observing unit ray per pool row (LandmarkToMatch::e_W) */
  const double* r0_W;         /* device, desc_begin[L] x 3: observing camera centre per pool row */
} okvfe_map_device;
/* = Frontend::matchToMapByThread (Frontend.cpp:1552-1589) for n_frames frames.  use_dev: device
 * n_frames x K flags or NULL (every keypoint).  Outputs (device, n_frames x K): landmark index
 * (-1 = none) and distance (match_threshold if none).  The call keeps the frames' keypoint order in a workspace
 * of its own per stream (ABI 6), so calls on different streams of one context may be in flight together, like
 * every other entry point of the batch API (the host side of a context is still one thread at a time). */
okvfe_status okvfe_match_to_map_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                              const uint8_t* use_dev, const okvfe_map_device* map,
                                              double reprojection_threshold, int32_t* best_landmark_dev,
                                              int32_t* best_dist_dev, void* stream);
/* = Frontend::matchToMapByThreadUnitialised (Frontend.cpp:1616-1719) for n_frames frames.
 * T_WC1: HOST array of n_frames poses (the one thing that crosses PCIe: 96 bytes per frame, through
 * the context's pinned parameter ring); previous_landmark_dev: device n_frames x K or NULL (-1);
 * outputs device: best_landmark / best_dist n_frames x K, hps_W n_frames x K x 4, hp_set
 * n_frames x K, already_matched n_frames (zeroed by this call). */
okvfe_status okvfe_match_to_map_uninitialised_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                                            const uint8_t* use_dev, const int32_t* previous_landmark_dev,
                                                            const okvfe_map_device* map, const okvfe_pose* T_WC1,
                                                            double focal_length, int32_t* best_landmark_dev,
                                                            int32_t* best_dist_dev, double* hps_W_dev, uint8_t* hp_set_dev,
                                                            int32_t* already_matched_dev, void* stream);
/* = the descriptor matching of Frontend::verifyRecognisedPlace (Frontend.cpp:330-355) of all L old
 * landmarks against n_frames frames.  Outputs device n_frames x L as okvfe_verify_place_match. */
okvfe_status okvfe_verify_place_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                              const okvfe_map_device* map, int32_t* k_min_dev,
                                              uint32_t* dist_min_dev, void* stream);

/* ---- cross-camera gather collective (RCCL over xGMI) ---------------------- */
/* The one exchange step of the path (okvis_frontend/src/Frontend.cpp:1990-2026 needs the keypoints
 * of BOTH cameras of a pair; with one camera per GPU they live on different ranks): an all-gather of
 * the ranks' gather blocks, issued from C on the caller's stream.  The library loads RCCL at first
 * use (dlopen of librccl.so.1: the copy already mapped by the process -- e.g. torch's -- is the
 * one found); a process that never calls these needs no RCCL.
 *   okvfe_comm_unique_id   ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank
 *                          (any side channel: MPI, a socket, torch.distributed.broadcast, a file);
 *   okvfe_comm_create      ncclCommInitRank on `device` (collective over all ranks).  world == 1 with
 *                          id == NULL makes a local communicator that never touches RCCL;
 *   okvfe_comm_wrap        adopts an ncclComm_t the caller already owns (not destroyed by
 *                          okvfe_comm_destroy);
 *   okvfe_gather_blocks    ncclAllGather(send, recv, bytes_per_rank, ncclUint8) on `stream`
 *                          (a hipStream_t; NULL = the HIP null stream): recv_dev holds world x
 *                          bytes_per_rank bytes, rank-major.  Asynchronous: ordered by the stream. */
typedef struct okvfe_comm okvfe_comm;
#define OKVFE_COMM_ID_BYTES 128
okvfe_status okvfe_comm_unique_id(uint8_t id[OKVFE_COMM_ID_BYTES]);
okvfe_status okvfe_comm_create(const uint8_t* id /* OKVFE_COMM_ID_BYTES or NULL */, int32_t world,
                               int32_t rank, int32_t device, okvfe_comm** out);
okvfe_status okvfe_comm_wrap(void* nccl_comm, int32_t world, int32_t rank, okvfe_comm** out);
void okvfe_comm_destroy(okvfe_comm* comm);
int32_t okvfe_comm_world(const okvfe_comm* comm);
int32_t okvfe_comm_rank(const okvfe_comm* comm);
const char* okvfe_comm_last_error(void);
okvfe_status okvfe_gather_blocks(okvfe_comm* comm, const void* send_dev, void* recv_dev,
                                 size_t bytes_per_rank, void* stream);

/* ---- device utilities for hosts without HIP headers ----------------------- */
/* The C++ mirror (okvis2_amd/host/) is dependency-free; these let it own streams and device
 * buffers (gather blocks, match rows).  Plain hipMalloc / hipStream / hipMemcpyAsync underneath. */
okvfe_status okvfe_device_alloc(int32_t device, size_t bytes, void** out_dev);
void okvfe_device_free(void* dev);
okvfe_status okvfe_stream_create(int32_t device, void** out_stream);
void okvfe_stream_destroy(void* stream);
okvfe_status okvfe_stream_synchronize(void* stream);
okvfe_status okvfe_copy_to_device(void* dst_dev, const void* src_host, size_t bytes, void* stream);
okvfe_status okvfe_copy_to_host(void* dst_host, const void* src_dev, size_t bytes, void* stream);
okvfe_status okvfe_device_fill(void* dst_dev, int32_t byte_value, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OKVFE_H_ */

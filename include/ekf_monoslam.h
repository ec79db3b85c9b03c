/* ekf_monoslam.h -- C ABI of the MI355X-native EKF-MonoSLAM predict/update core.
 *
 * Drop-in boundary for the math methods of the reference's `class VSlamFilter`
 * (mono-slam/src/vslamRansac.hpp:94-141 of engyasin/EKF-MonoSLAM_for_3D-reconstruction).
 * The reference has no FFI today (it is a plain C++ class subclassed by RosVSLAM,
 * RosVSLAMRansac.hpp:19-38); every entry point below names the reference method or
 * source range it replaces.  "vR.cpp" = mono-slam/src/vslamRansac.cpp.
 *
 * Conventions
 *  - every function returns an `int` status (EKF_OK = 0) unless it documents a count;
 *    `ekf_last_error` gives the message of the last failure on that handle.
 *  - the filter owns all device memory; every pointer crossing the ABI is a caller-owned
 *    HOST buffer unless the parameter is named `d_*` (device pointer, resident in HBM).
 *  - scalars are `float` for an EKF_F32 filter and `double` for an EKF_F64 filter;
 *    such buffers are declared `void*`.
 *  - matrices crossing the ABI are COLUMN-MAJOR (Eigen's default, so a
 *    `VSlamFilter`-shaped C++ wrapper can `Eigen::Map` them directly).
 *  - state layout (vR.hpp:12-25, vR.cpp:163-164, 483-486):
 *      mu = [ r(0:3) | q=(w,x,y,z)(3:7) | v(7:10) | omega(10:13) | map_scale(13) | features... ]
 *    inverse-depth feature = [x y z theta phi rho] (6), XYZ feature = [X Y Z] (3),
 *    contiguous in insertion order.  camera_dim = 14 reproduces the reference
 *    (#define STATE_DIM 14, vR.cpp:22); 13 drops the map-scale element.
 *  - a handle is not thread-safe (the reference filter is single-threaded, node.cpp:865).
 *  - there is no CPU fallback: without a HIP device `ekf_create` fails.
 */
#ifndef EKF_MONOSLAM_H_
#define EKF_MONOSLAM_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 3): + ekf_export_points_table, ekf_get_feature_ids, ekf_set_feature_meta, ekf_num_archived,
 * EKF_OPT_FUSED_LAUNCHES; the sharded filter accepts the whole update flow; - ekf_debug_flow_trace */
/* 4 (round 4): + EKF_OPT_W_RECOMPUTE (default on), ekf_get_chunk_plan; the map getters are collective on a sharded filter */
/* 5 (round 5): no new entry points; EKF_OPT_SPLIT_BF16 is ON by default (large maps: covariance downdate on the bf16 matrix
 * pipe at fp32 accuracy) and applies to the sharded step too; the sharded step runs the sequential form (EKF_OPT_W_RECOMPUTE);
 * the gathered diagonal blocks of the map getters stay valid between two filter steps */
/* 6 (round 6): + ekf_launch_kinds / ekf_launch_kind_name / ekf_launch_count (which launch structure an update took) */
#define EKF_ABI_VERSION 6

typedef struct ekf_filter ekf_filter;

enum ekf_dtype { EKF_F32 = 0, EKF_F64 = 1 };

enum ekf_status {
  EKF_OK = 0,
  EKF_ERR_ARG = 1,         /* bad argument                                   */
  EKF_ERR_CAPACITY = 2,    /* capacity_features exceeded                     */
  EKF_ERR_DEVICE = 3,      /* HIP runtime failure                            */
  EKF_ERR_STATE = 4,       /* call order (e.g. update before predict)        */
  EKF_ERR_NUMERIC = 5,     /* innovation covariance not positive definite    */
  EKF_ERR_UNSUPPORTED = 6
};

/* ConfigVSLAM (ConfigVSLAM.h:23-48) + camConfig (camModel.hpp:9-11) + frame size.
 * fx, fy, u0, v0 are the values AFTER the reference's division by `scale`
 * (ConfigVSLAM.cpp:87-103); image_width/height are frame.size() after the resize
 * of captureNewFrame (vR.cpp:236). */
typedef struct ekf_config {
  float sigma_vx, sigma_vy, sigma_vz;
  float sigma_wx, sigma_wy, sigma_wz;
  float rho_0, sigma_rho_0;
  int window_size, sigma_pixel, kernel_size, sigma_size, scale;
  float T_camera;
  int nInitFeatures, min_features, max_features, forsePlane;
  float fx, fy, u0, v0, k1, k2, k3, p1, p2;
  int image_width, image_height;
} ekf_config;

enum ekf_option {
  /* 0: in-place strip kernel (touches 26 n elements); 1: streaming out-of-place
   * Sigma' = F Sigma F^T + Q (reads n^2, writes n^2 -- the formulation the reference's
   * `.eval()` at vR.cpp:477 has, and the one the HBM roofline of P-propagate is quoted on). */
  EKF_OPT_PROPAGATE_STREAMING = 0,
  /* 1 (default): hand-written MFMA kernels for the dense contractions (fp32: v_mfma_f32_32x32x2_f32,
   * fp64: v_mfma_f64_16x16x4_f64); 0: plain VALU tiles. */
  EKF_OPT_USE_MFMA = 1,
  /* profiling level: 0 off, 1 HIP events around the dominant kernels (downdate, streaming propagate),
   * 2 around every kernel, 3 as 1 but only in every 8th update since the last ekf_profile_reset (each pair of
   * events costs ~6 microseconds of queue time: 5 % of a step at N = 200). */
  EKF_OPT_PROFILE = 2,
  /* Chunked factorisation: 0 = one chunk, one stream (plain blocked Cholesky + one solve + one downdate; for A/B runs:
   * the solve then goes through the explicit inverse of the WHOLE factor, and Sigma after an update is an order of
   * magnitude further from the fp64 result than on the default path at 2M >= 2000 -- 1.2e-4 against 1.2e-5 of
   * max|Sigma|, tools/acc_check_sizes.py);
   * 1 = the default three column chunks, the solve / downdate / W re-evaluation (or update) of every chunk but the last on a
   * second, CU-masked stream beside the serial chain; k >= 2 = k equal chunks;
   * -1 (default): as 1 when the chain has at least 8 block steps (m >= 1024), else as 0. */
  EKF_OPT_PIPELINE = 3,
  /* 1 (default since round 5): the covariance downdate Sigma -= V_g V_g^T of large maps (at least 23 tile rows of 128: N >= ~480
   * inverse-depth features on a 256-CU device) runs on the bf16 matrix pipe AT FP32 ACCURACY: each fp32 operand is split
   * exactly into three bf16 values, a = a1 + a2 + a3, and six of the nine bf16 products -- all but a2 b3, a3 b2, a3 b3, which
   * together are <= 2^-24 |a||b| in the worst case and 2^-28 |a||b| on average, the size of the fp32 product's own rounding
   * (2^-24 worst, 2^-25.5 mean) -- are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (csrc/ekf_syrk6.hpp).  Every other contraction, every
   * accumulation, the state and the covariance stay fp32.  Measured against the fp64 oracle the result is as close as the
   * fp32 instruction's (tests/test_gpu_parity.py::test_split_bf16_downdate_is_fp32_accurate runs BOTH arithmetics -- the
   * launch counters prove which kernel each filter ran -- and ::test_n1000_exact_fp32_downdate_matches_fp64_oracle /
   * ::test_n1000_default_pipeline_matches_fp64_oracle hold each at N = 1000 against the fp32-oracle yardstick); it is not bit-equal
   * to it.  Sigma stays exactly symmetric, and a rank of a sharded filter computes bit-identical rows (every element pair is
   * one sum, whoever computes it).  0: every contraction on v_mfma_f32_32x32x2_f32 (the arithmetic of rounds 1-4).  fp32
   * filters only; smaller maps and fp64 filters are not affected. */
  EKF_OPT_SPLIT_BF16 = 4,
  /* 0 (default: the reference's model -- the map is static, features carry no process noise).  v > 0: every
   * predict adds v x 1e-12 to the variance of every feature state (Sigma[i][i], i >= camera_dim): the "stabilising
   * noise" of EKF-SLAM practice, for callers who want it.  The fp32 filter does not need it to stay positive: with
   * every feature measured in every frame Sigma stays positive to rounding over every run followed so far (N = 200:
   * 12000 frames, 1000: 3000, 4000: 1200; profiles/r2_drift_after_fix.txt).  Any dtype. */
  EKF_OPT_FEATURE_NOISE = 5,
  /* 1 (default): launch-bound sequences go out as fused launches where that changes no result beyond rounding --
   * camera step + strip congruence + per-feature h / H of ekf_predict as one launch (bit-identical to the three), and,
   * when the innovation fits one 128-column block (2 M + 3 <= 128: the reference's operating point of <= 35 features),
   * gain solve + state update as one launch without the panel step (same sums in another order: fp32 rounding).
   * 0: one launch per kernel (what the per-kernel profile of EKF_OPT_PROFILE = 2 times). */
  EKF_OPT_FUSED_LAUNCHES = 6,
  /* Chunked factorisation, fp32 MFMA path (round 4).  1 (default): after the downdate of column chunk g the columns of
   * W = Sigma H^T of the NEXT chunk are re-evaluated from the downdated Sigma -- the sequential form of the update:
   * W'_h = (Sigma - sum_{g<h} V_g V_g^T) H_h^T, algebraically what the right-looking GEMM update
   * W_h -= V_g L_hg^T produces, for 26 n w_h flop and one read of those columns of Sigma instead of 2 n w_g w_h flop
   * (15 of the 99 GFLOP of a step at N = M = 1000; 15 % at N = 4000); only the innovation row is still updated
   * right-looking (inside the downdate launch).  0: the right-looking W update of rounds 1-3.  The sharded step runs the
   * same sequential form since round 5 (a rank re-evaluates ITS rows of W from its downdated rows of Sigma).  Same result
   * up to fp32 rounding. */
  EKF_OPT_W_RECOMPUTE = 7
};

/* Fills `cfg` with the reference defaults (ConfigVSLAM.cpp:27-47, camModel.hpp:22-31). */
void ekf_config_default(ekf_config* cfg);

int ekf_abi_version(void);

/* VSlamFilter::VSlamFilter (vR.cpp:142-223): mu0, Sigma0, Vmax, Vmax_n.  `device` is the HIP
 * device ordinal.  capacity_features bounds numOfFeatures() for the life of the handle. */
int ekf_create(const ekf_config* cfg, int camera_dim, int capacity_features, int dtype,
               int device, ekf_filter** out);
void ekf_destroy(ekf_filter* f);
/* Message of the last failure (f may be NULL: last failure of ekf_create). */
const char* ekf_last_error(const ekf_filter* f);

/* captureNewFrame's dT (vR.cpp:226-233) and getDt (vR.cpp:247). */
int ekf_set_dt(ekf_filter* f, double dT);
double ekf_get_dt(const ekf_filter* f);

/* Launch on a caller-provided hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int ekf_set_stream(ekf_filter* f, void* hip_stream);
int ekf_set_option(ekf_filter* f, int option, int value);
int ekf_synchronize(ekf_filter* f);

/* VSlamFilter::addFeature (vR.cpp:309-371).  Returns 1 = added, 0 = pixel outside the image
 * margin (vR.cpp:314), negative = -ekf_status. */
int ekf_add_feature(ekf_filter* f, double u, double v);
/* VSlamFilter::removeFeature (vR.cpp:373-421): splices the feature out of mu / Sigma and
 * shifts position_in_state of later features. */
int ekf_remove_feature(ekf_filter* f, int index);
/* Batched removal, one compaction pass; same result as removing the listed indices in
 * descending order (the order of the loop at vR.cpp:1296-1299). */
int ekf_remove_features(ekf_filter* f, const int* indices, int count);

/* VSlamFilter::predict (vR.cpp:451-603) without the image blur: covariance propagation
 * Sigma <- F Sigma F^T + Q, Predict_State, and for every feature h, the compact
 * measurement Jacobian (2x7 camera block + 2x6 / 2x3 feature block), the visibility test
 * (vR.cpp:529) and the rho <= 0 removal flag (vR.cpp:517-521), and the 2x2 diagonal block
 * of St = H Sigma H^T + sigma_px^2 I that `Patch::findMatch` is gated with (vR.cpp:875).
 * t_ctl / r_ctl: 3 scalars each (may be NULL = 0); vcontrol selects Vmax vs Vmax_n. */
int ekf_predict(ekf_filter* f, const void* t_ctl, const void* r_ctl, int vcontrol);
/* The motion Jacobian Ft (the 13x13 block System_model_jacobian fills, vR.cpp:454, 1492-1507) and the process
 * noise Q = Ft[:,7:13] (V / dT^2) Ft[:,7:13]^T (vR.cpp:463-475) of the last ekf_predict, 13 x 13 column-major each;
 * either pointer may be NULL. */
int ekf_get_motion_jacobian(ekf_filter* f, void* Ft, void* Q);
/* Re-evaluates h / H / flags / 2x2 blocks at the current state (the recomputation at
 * vR.cpp:1080-1117 before the high-innovation update). */
int ekf_measure(ekf_filter* f);

/* Per-feature outputs of the last predict/measure.  Any pointer may be NULL.
 * h: 2 per feature; visible / remove_flag: 1 byte per feature; S2x2: 4 per feature,
 * column-major 2x2; Hc: 14 per feature (2x7 column-major); Hf: 12 per feature (2x6
 * column-major, last 3 columns zero for an XYZ feature). */
int ekf_get_predictions(ekf_filter* f, void* h, unsigned char* visible,
                        unsigned char* remove_flag, void* S2x2, void* Hc, void* Hf);

/* The EKF update block (vR.cpp:1245-1284; the same block at 1053-1061 and 663-676):
 * St = H Sigma H^T + R, Kt = Sigma H^T St^-1, mu += Kt (z - h), Sigma <- (I - Kt H) Sigma,
 * normalizeQuaternion (vR.cpp:1625-1642).  `indices` (strictly ascending feature indices, M of them: the order of
 * position_in_z, vR.cpp:589)
 * is the measured set, z holds 2 pixels per listed feature.  plane_constraint != 0 appends
 * the forsePlane pseudo-measurement (vR.cpp:1250-1263, 1272).  M = 0 and no plane: no-op.
 * A factorisation of St that meets a non-positive pivot is reported as EKF_ERR_NUMERIC by the next
 * synchronising call (ekf_synchronize, any getter).  (Through most of round 2 an fp32 map whose features were ALL
 * measured in EVERY frame ended there after a few hundred frames; the cause -- partial sums rounded at the magnitude
 * of Sigma -- was fixed, see EKF_OPT_FEATURE_NOISE above and DESIGN.md section 8: the fp32 covariance now stays
 * positive over every run followed so far.) */
int ekf_update(ekf_filter* f, const void* z, const int* indices, int M, int plane_constraint);
/* Same with z (2 M scalars) and indices (M ints) already resident in DEVICE memory.  Contract:
 *  - both buffers are read IN PLACE and ASYNCHRONOUSLY by the kernels of the queued step: they must stay allocated
 *    and unmodified until the step has run (ekf_synchronize, any getter, or an event the caller records on the
 *    filter's stream after this call);
 *  - the host cannot validate them.  The first kernel of the step checks every index (inside [0, N), strictly
 *    ascending); a bad list raises a device flag that the next synchronising call reports as EKF_ERR_ARG.  Until
 *    then every kernel clamps the indices it uses into [0, N), so nothing is read out of bounds, but the state
 *    after such a step is meaningless. */
int ekf_update_device(ekf_filter* f, const void* d_z, const int* d_indices, int M,
                      int plane_constraint);

/* computeEllipsoidParameters (vR.cpp:1368-1382) for every feature, from the 2x2 St blocks of the last
 * predict/measure: out holds 3 ints per feature (semi-axis of the smaller eigenvalue, of the larger one,
 * angle in degrees of the minor-axis eigenvector with non-negative x component). */
int ekf_get_search_ellipses(ekf_filter* f, int sigma_size, int* out);

/* 1-point RANSAC hypothesis evaluation (vR.cpp:986-1034), every measured feature as a hypothesis, in
 * one pass on the device: counts[k] = number of listed features within `threshold` pixels of their
 * measurement after the single-feature update with feature indices[k] (the reference uses
 * threshold = 2 * sigma_pixel, :968).  best = the hypothesis with most inliers (lowest k on ties;
 * the reference keeps the flags of the LAST hypothesis it happened to draw, :1022 -- its stopping rule
 * makes that one of the best; returning the best is the deterministic equivalent), inliers_of_best:
 * 1 byte per listed feature.  Any output pointer may be NULL.  Needs ekf_predict / ekf_measure. */
int ekf_ransac_1point(ekf_filter* f, const void* z, const int* indices, int M, double threshold,
                      int* counts, unsigned char* inliers_of_best, int* best);

/* High-innovation rescue (vR.cpp:1066-1117), after the low-innovation update: for the listed features
 * (matched but not low-innovation inliers) h / H are re-evaluated with the feature entries of the CURRENT
 * (updated) state and the camera pose `cam_before` = mu[0:7] of the state BEFORE that update (the
 * reference's choice, :1069-1072), S_hi = H Sigma H^T without measurement noise (:1113), and
 * is_hi[k] = ((h - z)^T S_hi^-1 (h - z) <= chi2_threshold) (the reference uses 1, :1066).  The listed
 * features keep the new h / H, so a following ekf_update over the rescued ones uses them (:1119-1130). */
int ekf_rescue_high_innovation(ekf_filter* f, const void* cam_before, const void* z, const int* indices,
                               int M, double chi2_threshold, unsigned char* is_hi);

/* The RANSAC branch of VSlamFilter::update in ONE call (vR.cpp:964-1130 + 1245-1284), for the matched features
 * `indices` (strictly ascending, M of them) with pixels z:
 *   1. every 1-point hypothesis is evaluated on the device (as ekf_ransac_1point, threshold `ransac_threshold`;
 *      the reference uses 2 * sigma_pixel, :968);
 *   2. seed == 0: the low-innovation set is that of the BEST hypothesis (most inliers, lowest index on ties) -- the
 *      deterministic form.  seed != 0: the reference's own loop is replayed on those counts -- hypotheses drawn
 *      without replacement with glibc's srand(seed) / rand() sequence (the reference seeds with time(NULL), :970),
 *      nhyp adapted as (int)(log(1 - 0.99) / log(1 - best_count / M)) (:1030), and the low-innovation set is that of
 *      the LAST hypothesis drawn (the reference overwrites the flags on every draw, :1022);
 *   3. EKF update with the low-innovation inliers, no plane rows (:1036-1064);
 *   4. the remaining matched features are re-linearised at (camera pose before step 3, features after it) and
 *      gated by (h - z)^T S_hi^-1 (h - z) <= chi2_threshold with S_hi = H Sigma H^T (:1066-1117; the reference uses 1);
 *   5. second EKF update with the rescued features and, if plane_constraint, the forsePlane rows (:1245-1284).
 * Outputs (any may be NULL): is_low_innovation / is_high_innovation, one byte per listed feature;
 * hypotheses_drawn = draws of the replayed loop (M in the deterministic form).  Needs ekf_predict. */
int ekf_update_two_stage(ekf_filter* f, const void* z, const int* indices, int M, int plane_constraint,
                         unsigned int seed, double ransac_threshold, double chi2_threshold,
                         unsigned char* is_low_innovation, unsigned char* is_high_innovation, int* hypotheses_drawn);

/* ---- image side: what sits between predict() and the EKF update in the reference ----
 * captureNewFrame's image (vR.cpp:234-245) AFTER the node's resize / grayscale: 8-bit, single channel,
 * image_width x image_height of the config; `stride` = bytes per row.  The frame is copied to the device.
 * While a frame is set, ekf_add_feature also captures the feature's window_size^2 template at
 * ((int)(u - w/2), (int)(v - w/2)) (Patch::Patch in addFeature, vR.cpp:318) and removals keep the
 * templates aligned with their features (vR.cpp:1296-1299). */
int ekf_set_frame(ekf_filter* f, const unsigned char* gray, int width, int height, int stride);
/* Patch::patch of feature `index`: window_size^2 bytes, row-major (test injection / inspection).
 * matching != 0 reads Patch::matching_patch (the blurred copy or the last matched window). */
int ekf_set_patch(ekf_filter* f, int index, const unsigned char* pixels);
int ekf_get_patch(ekf_filter* f, int index, int matching, unsigned char* out);
/* hi_out_blurred of every feature (vR.cpp:546, 575): the prediction at the pose r + v T_camera dT,
 * q (x) quat(w T_camera dT), 2 scalars per feature.  ekf_predict computes it, together with
 * Patch::blur (Patch.cpp:50-57, libblur.cpp:17-79: line kernel, filter2D with BORDER_REFLECT_101) of every
 * visible feature's template, whenever templates are present. */
int ekf_get_blur_predictions(ekf_filter* f, void* hb);
/* Patch::findMatch (Patch.cpp:215-293) for every visible feature, on the device: NCC
 * (computeCorrelation, Patch.cpp:295-329) of the matching template against every window whose centre lies
 * in the clamped sigma_size box and inside the Mahalanobis ellipse of the feature's 2x2 St block; first
 * maximum in scan order; found[i] = (score >= threshold) (the reference's patch_matching_threshold is 0.8,
 * Patch.cpp:14).  z: 2 scalars per feature (the matched centre, or -1 -1), score: the best NCC (-1 if no
 * candidate).  A found feature's matching template becomes the matched window (Patch.cpp:286).
 * Any of z / found / score may be NULL. */
int ekf_find_matches(ekf_filter* f, double threshold, void* z, unsigned char* found, float* score);

/* Full St for a measured set (vR.cpp:598): out is m x m column-major, m = 2M (+3). */
int ekf_innovation_covariance(ekf_filter* f, const int* indices, int M, int plane_constraint,
                              void* S_out);
/* Kt of the last update (n x m column-major, n = state dim before normalisation). */
int ekf_get_gain(ekf_filter* f, void* K_out);
int ekf_last_measurement_rows(const ekf_filter* f);

/* VSlamFilter::convert2XYZ_ifLinear (vR.cpp:741-772): returns 1 converted, 0 not, <0 error. */
int ekf_convert_xyz_if_linear(ekf_filter* f, int index);
/* VSlamFilter::convert2XYZ_ifLinearAll (vR.cpp:776-780): returns the number converted. */
int ekf_convert_xyz_if_linear_all(ekf_filter* f);

/* numOfFeatures (vR.cpp:127-129) and mu.size(). */
int ekf_num_features(const ekf_filter* f);
int ekf_state_dim(const ekf_filter* f);
/* Patch::position_in_state / Patch::coding (0 = inverse depth, 1 = XYZ) per feature. */
int ekf_get_feature_layout(const ekf_filter* f, int* position_in_state, int* coding);

/* getState (vR.cpp:135-140) generalised to any segment; set_* inject test inputs. */
int ekf_get_state(ekf_filter* f, void* out, int offset, int count);
int ekf_set_state(ekf_filter* f, const void* in, int offset, int count);
/* getSigma (vR.cpp:131-133) generalised to any block (RosVSLAM reads Sigma.block directly,
 * RosVSLAMRansac.cpp:171-183). Column-major rows x cols. */
int ekf_get_sigma_block(ekf_filter* f, void* out, int r0, int c0, int rows, int cols);
int ekf_set_sigma_block(ekf_filter* f, const void* in, int r0, int c0, int rows, int cols);
/* Diagnostics only (no reference counterpart; tools/determinism_probe_sharded.py): a block of the workspace of the LAST
 * update, row-major rows x cols in the filter's dtype.  which = 0: W = Sigma H^T as the triangular solves read it,
 * 1: V = W L^-T (the factor of the covariance downdate).  Rows: state rows, then the padding; columns: 2 x list slot.
 * which = 2 (round 6; a filter created with EKF_CHAIN_TRACE=1 in the environment): the task trace of the persistent chain
 * kernel of the last update -- `rows` records of 8 32-bit words from record r0 on (cols = 8; r0 = -1 starts at the header,
 * whose word 0 counts the records): type | workgroup << 8 | critical << 24, block step, row block, column block, and the
 * 100 MHz wall clock at draw / dependencies met / computed / published (tools/chain_trace.py).
 * which = 3 (round 6; EKF_SMALL_STAMPS=1): the phase stamps of the one-launch update of a small map (k_update_small_onelaunch):
 * 16 64-bit words of the 100 MHz clock as rows = 16, cols = 2 32-bit halves, r0 = c0 = 0 (tools/small_stamps.py). */
int ekf_peek_workspace(ekf_filter* f, int which, void* out, int r0, int c0, int rows, int cols);
/* Covariance_Parameter (vR.cpp:841-866): trace of Sigma[0:7,0:7]. */
int ekf_covariance_parameter(ekf_filter* f, double* out);
/* Invariants of the device-resident covariance, evaluated on the device (no n^2 copy): max |Sigma| outside the live
 * n x n inside the padded buffer (the tile kernels rely on exact zeros there), max |Sigma[i][j] - Sigma[j][i]| and
 * max |Sigma[i][j]| over the live block (the reference never symmetrises, vR.cpp:1279; this implementation keeps
 * Sigma exactly symmetric).  Any pointer may be NULL.  Test / debug entry: no counterpart in the reference. */
int ekf_check_invariants(ekf_filter* f, double* max_abs_outside_live, double* max_asymmetry, double* max_abs_live);
/* inverseDepth2XyzWorld mode 1 + Jf Sigma_ff Jf^T (vR.cpp:690-738,
 * RosVSLAMRansac.cpp:177-183): world point (3) and 3x3 covariance (column-major). */
int ekf_feature_xyz(ekf_filter* f, int index, void* xyz, void* cov3x3);

/* RosVSLAM::getPointsFeatures (RosVSLAMRansac.cpp:340-418), the table behind `points.txt`: one row of
 * 12 scalars per feature, ROW-MAJOR N x 12: [X Y Z] * map_scale (mu[13]; 1 for camera_dim 13), then the
 * 3x3 covariance block row by row.  The reference fills the rows of XYZ features only; with
 * convert_inverse_depth != 0 the inverse-depth rows carry inverseDepth2XyzWorld(f) and Jf Sigma Jf^T. */
int ekf_export_points(ekf_filter* f, void* out, int convert_inverse_depth);
/* The same table in the reference's own layout (RosVSLAMRansac.cpp:340-418): rows indexed by Patch::real_index --
 * (real_index of the LAST live feature) + 1 rows (:350-352), ROW-MAJOR rows x 12 --, live XYZ features at their rows,
 * live inverse-depth rows zero, then the patches ARCHIVED at removal written over their rows (:406-414): a removed
 * XYZ feature with n_find > 5 leaves XYZ_pos and the 3x3 block of Sigma of the moment of its removal behind
 * (deleted_patches, vR.cpp:394-404; captured on the device by ekf_remove_feature(s)).  The archive is emptied by the
 * call that finds more than 7000 entries in it (:396-404).  *rows receives the row count; out == NULL only queries it;
 * max_rows < rows is EKF_ERR_ARG.  An archived patch whose real_index lies beyond the table is skipped (the reference
 * writes out of bounds there). */
int ekf_export_points_table(ekf_filter* f, void* out, int max_rows, int* rows);
/* Patch::real_index (vR.cpp:148, 318-319: 1, 2, ... in creation order, never reused) and Patch::n_find (Patch.cpp:86,
 * 145: 1 at creation, +1 per ekf_update_two_stage in which the feature is a low- or high-innovation inlier) per live
 * feature; either pointer may be NULL.  ekf_update (the bare update block) does not touch n_find: a caller that
 * composes its own flow keeps it with ekf_set_feature_meta (a negative value leaves that field alone). */
int ekf_get_feature_ids(const ekf_filter* f, int* real_index, int* n_find);
int ekf_set_feature_meta(ekf_filter* f, int index, int real_index, int n_find);
/* deleted_patches.size() */
int ekf_num_archived(const ekf_filter* f);

/* Per-kernel HIP-event timing (EKF_OPT_PROFILE).  Kernel ids are dense in
 * [0, ekf_profile_kernels()). */
int ekf_profile_kernels(void);
const char* ekf_profile_kernel_name(int kernel_id);
int ekf_profile_read(ekf_filter* f, int kernel_id, double* total_ms, long long* launches);
int ekf_profile_reset(ekf_filter* f);
/* Algorithmic flop of the launches timed under `kernel_id` since the last reset (kept for "downdate_syrk" only:
 * n^2 x the real columns of the chunk (symmetric half).  Where the launch also carries a right-looking update of its
 * chunk -- exact-fp32 path, EKF_FUSE_WU -- that product is counted with it: under EKF_OPT_W_RECOMPUTE = 1 (the default)
 * only the innovation ROW is updated, 2 x 1 x (m - c1) x the chunk's columns; under = 0 the whole W, 2 (n + 1) (m - c1)
 * x those columns).  Where the other pieces of the sequential form are booked: the re-evaluation W'_h = Sigma' H_h^T under
 * "sigma_ht"; the innovation-row update under EKF_OPT_SPLIT_BF16 = 1 rides as the first workgroups of the downdate's
 * launch (its time is inside "downdate_syrk", its 2 (m - c1) x columns flop are not counted), a stand-alone one
 * (EKF_FUSE_WU = 0 on the exact-fp32 path, a rank that owns no rows) under "w_update"; the plane image of V_g under "misc". */
int ekf_profile_work(ekf_filter* f, int kernel_id, double* flop);
/* How the last ekf_update factored S (what the algorithmic flop of a step depends on): `block` = rows of a block step
 * (128 fp32 MFMA, 64 otherwise), ends[g] = block step at which column chunk g ends (the last one = m_pad / block),
 * `w_recompute` = 1 if the W columns of the later chunks were re-evaluated from the downdated Sigma (EKF_OPT_W_RECOMPUTE)
 * rather than updated right-looking.  Returns the number of chunks (0 before the first update; at most `max_chunks`
 * entries are written). */
int ekf_get_chunk_plan(ekf_filter* f, int* ends, int max_chunks, int* block, int* w_recompute);

/* Which launch structure the updates of this handle actually took: one host-side counter per kind, incremented at the
 * launch (always on, no device cost), cleared by ekf_profile_reset.  Diagnostics with no reference counterpart: the
 * bit-identity tests of the tuning knobs assert with it that a knob really changed the launches, and bench.py reports the
 * downdate kernel that ran instead of re-deriving the library's selection rule. */
enum ekf_launch_kind {
  EKF_LAUNCH_DOWNDATE_BF16X6 = 0,     /* k_syrk_bf16x6 (EKF_OPT_SPLIT_BF16 = 1, large maps)                              */
  EKF_LAUNCH_DOWNDATE_F32,            /* k_gemm_mfma<DOWNDATE> 128 x 128, plain tile list (also every VALU / fp64 downdate) */
  EKF_LAUNCH_DOWNDATE_F32_FUSED_WU,   /* ... carrying the chunk's right-looking update as its first tiles (EKF_FUSE_WU)  */
  EKF_LAUNCH_DOWNDATE_F32_HALF_TAIL,  /* ... with 64 x 128 half tiles at the end of its list (EKF_SPLIT_TAIL)            */
  EKF_LAUNCH_DOWNDATE_F32_T64,        /* 64 x 64 tiles (small maps)                                                      */
  EKF_LAUNCH_ROW_RIDER,               /* innovation-row update as the first workgroups of k_syrk_bf16x6                  */
  EKF_LAUNCH_ROW_GEMV,                /* ... as a stand-alone k_innov_row_update launch                                  */
  EKF_LAUNCH_ROW_TILE_GEMM,           /* ... through the 64 x 128 tile GEMM (EKF_ROW_GEMV = 0)                           */
  EKF_LAUNCH_W_UPDATE_GEMM,           /* right-looking update of all of W (EKF_OPT_W_RECOMPUTE = 0), stand-alone launch  */
  EKF_LAUNCH_W_RECOMPUTE,             /* W' = Sigma' H^T re-evaluation launches (EKF_OPT_W_RECOMPUTE = 1)                */
  EKF_LAUNCH_CHAIN_STEP,              /* launches of the per-block-step chain (diagonal factor, panel, trailing)          */
  EKF_LAUNCH_CHAIN_PERSISTENT,        /* one launch per column chunk: the look-ahead chain kernel (round 6)              */
  EKF_LAUNCH_SOLVE,                   /* triangular-solve launches on one wave group per tile                            */
  EKF_LAUNCH_SOLVE_TWO_GROUPS,        /* ... on two wave groups per tile (EKF_SOLVE_S2)                                  */
  EKF_LAUNCH_UPDATE_ONEBLOCK,         /* k_solve_state_oneblock (2 M + 3 <= 128)                                         */
  EKF_LAUNCH_UPDATE_ALLINONE,         /* k_update_oneblock_small (... and a small map)                                   */
  EKF_LAUNCH_CHAIN_TRAIL_DIAG,        /* k_trail_diag: the trailing update of a block step and the factor of the next as one launch (round 6) */
  EKF_LAUNCH_SPLIT_IMAGE,             /* k_split_image of V_g as a launch of its own (else: written by the solve's tiles) */
  EKF_LAUNCH_STATE_UPDATE_TAIL,       /* mu += V y done by the workgroups of the last k_syrk_bf16x6 launch that run out of tiles */
  EKF_LAUNCH_UPDATE_ONELAUNCH,        /* k_update_small_onelaunch: W, S, the factor and the whole rest of the update of a small map (n_pad <= 256) as ONE launch (round 6) */
  EKF_LAUNCH_CHAIN_DIST_GATHER,       /* sharded step, distributed chain: all-gathers of a block step's panel (round 6) */
  EKF_LAUNCH_CHAIN_STEP_FUSED,        /* k_chain_step_fused: factor, panel and trailing update of a block step as ONE launch (one-chunk maps, round 6) */
  EKF_LAUNCH_KINDS
};
int ekf_launch_kinds(void);
const char* ekf_launch_kind_name(int kind);
int ekf_launch_count(ekf_filter* f, int kind, long long* launches);

/* ---- multi-GPU: row-panel sharding, one process per GPU ----------------------------------------
 * Every rank holds the same filter (same calls in the same order on every rank: add / remove / convert / predict /
 * update) and OWNS a contiguous range of features: it keeps valid the rows of Sigma of those features (all columns)
 * plus a replica of the camera rows; mu is replicated.  One sharded step:
 *   predict   camera step + strips; h / H / flags of the OWN features
 *       -> all-gather of the per-feature records (32 scalars per feature)             "reassemble H"
 *   update    nu; W = Sigma H^T rows {camera, own}; rows of S of the own MEASURED features
 *       -> all-gather of the row panels of S                                           "reassemble S"
 *             Cholesky chain of S in column chunks -- replicated on every rank up to 39 block steps (N < 2500); from 40 on
 *             DISTRIBUTED (round 6): a rank keeps its own 128-row blocks (cyclic), every diagonal block and the inverse strip of
 *             the trailing matrix up to date, and per block step
 *       -> all-gather of the rank's blocks of the panel                                (64 KB per block; the step's column of L)
 *             (bit-identical to the replicated chain; EKF_SHARD_DIST_CHAIN=0 keeps it replicated, EKF_SHARD_DIST_MIN_BLOCKS
 *             moves the threshold); per chunk g, beside the chain on a second stream:
 *             V_g = W_g Z_gg for rows {camera tile, own panel, innovation row} (one queued launch)
 *       -> all-gather of the own rows of V_g                                           (n x 2M scalars per step in all)
 *             Sigma[own rows, :] -= V_g[own rows] V_g^T (+ the replicated camera tile; the canonical tiles of k_syrk_bf16x6 that
 *             touch an own block: bit-identical rows), then W' of the next chunk re-evaluated on rows {camera, own} from the
 *             downdated Sigma (round 5: the sequential form; the right-looking W update of rounds 1-4 only under
 *             EKF_OPT_W_RECOMPUTE = 0); after the last chunk mu += V y, normalisation.
 * Any measured subset (strictly ascending list, M <= N), inverse-depth and XYZ features, the plane rows and any N
 * (ranks may own different numbers of features, or none) are supported.  add_feature appends to the LAST rank's
 * range, remove / convert compact every rank's copy alike; when a rank owns more than 1.125 x the mean number of
 * rows the next predict re-partitions (ekf_shard_rebalance: all-gather of the row panels of Sigma, after which every
 * row is valid on every rank, then a fresh row-balanced partition).
 *
 * The library does no communication itself: every exchange is ONE call of the host's all-gather on device staging
 * buffers -- `world` equal slots, slot g = the `bytes_per_rank` bytes rank g passed as d_send, delivered to every
 * rank, enqueued on `hip_stream` (the library packs before and unpacks after on that same stream).  With
 * torch.distributed this is `all_gather_into_tensor` (backend "nccl" = RCCL over xGMI; sharded.py); a C++ node
 * passes a function that calls ncclAllGather(d_send, d_recv, bytes_per_rank, ncclChar, comm, stream).  Every rank
 * makes the same sequence of calls with the same sizes.  Return 0 on success. */
typedef int (*ekf_allgather_fn)(void* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);

typedef struct ekf_shard_info {
  int rank, world, N, state_dim;
  int f_begin, f_end;                     /* own feature range                                             */
  int row_begin, row_end;                 /* own state rows                                                 */
  int max_rows_any_rank;                  /* the largest panel (imbalance = this x world / (n - camera_dim)) */
  int rebalances;                         /* re-partitions since ekf_shard_configure                        */
} ekf_shard_info;

/* Switches the filter to sharded operation.  Call it at a point where every rank holds the same, fully valid
 * filter (e.g. right after the identical construction of the map).  From then on ekf_predict, ekf_update (host
 * z / indices), ekf_add_feature, ekf_remove_feature(s) and ekf_convert_xyz_if_linear(_all) run the sharded step;
 * getters of Sigma are valid for the camera rows and the own rows only (ekf_shard_rebalance makes all rows valid).
 * Round 3: the whole update() flow runs sharded as well -- ekf_update_device (the list is copied to the host once),
 * ekf_update_two_stage, ekf_ransac_1point (every rank evaluates the hypotheses on the listed features it owns, the
 * partial inlier counts are all-gathered), ekf_rescue_high_innovation and the 2x2 St blocks behind ekf_get_predictions /
 * ekf_get_search_ellipses (owner-computes, the blocks are all-gathered), ekf_innovation_covariance; under sharding their
 * measured lists must be strictly ascending.  Round 4: the image side (ekf_set_frame, ekf_set_patch, ekf_find_matches,
 * the predicted blur inside ekf_predict) works on a sharded filter too -- frame and templates are replicated (every
 * rank makes the same calls), the 2x2 blocks the search gates with are all-gathered, and every rank searches every
 * feature (at most 124 us at N = 1000, far below one exchange): the templates stay identical on every rank.
 * world = 1 needs no callback.
 *
 * COLLECTIVE CONTRACT.  On a sharded filter (world > 1) the following entry points run one or more all-gathers and are
 * therefore COLLECTIVE: every rank must call them, with the same arguments, in the same order relative to each other --
 * a rank that skips one, or calls them in another order, leaves the others waiting inside the collective:
 *   ekf_predict, ekf_update, ekf_update_device, ekf_shard_update, ekf_update_two_stage, ekf_shard_rebalance,
 *   ekf_remove_feature(s), ekf_convert_xyz_if_linear(_all) (linearity flags; the removal archive gathers the owners'
 *   3 x 3 blocks), ekf_innovation_covariance, ekf_ransac_1point, ekf_rescue_high_innovation,
 *   ekf_get_predictions WITH s2 and ekf_get_search_ellipses (the 2x2 St blocks; gathered once per predict, so the
 *   first of these calls after a predict is the collective one -- make the same calls on every rank),
 *   ekf_find_matches (the same 2x2 blocks),
 *   ekf_feature_xyz, ekf_export_points, ekf_export_points_table (round 4: a feature's covariance block is valid on
 *   its owner only; the owners' diagonal blocks are all-gathered first, so every rank returns the same table.  Round 5:
 *   the gathered blocks stay valid until the next call that changes mu, Sigma, the layout or the sharding (predict, the
 *   updates, add / remove / convert, the setters, re-balance, ekf_set_option), so a loop of N per-feature getters between
 *   two filter steps costs ONE all-gather, not N -- the first getter after such a call is the collective one, as for the
 *   2x2 blocks above: make the same getter calls on every rank, or none).
 * Local (no exchange): ekf_add_feature, ekf_get_state, ekf_get_sigma_block (valid for camera + own rows),
 * ekf_get_predictions without s2, ekf_covariance_parameter, ekf_shard_get_info, the setters, ekf_last_error. */
int ekf_shard_configure(ekf_filter* f, int rank, int world, ekf_allgather_fn allgather, void* ctx);
int ekf_shard_get_info(ekf_filter* f, ekf_shard_info* out);
/* ekf_update with z (2 M scalars) resident in device memory and the measured list on the host. */
int ekf_shard_update(ekf_filter* f, const void* d_z, const int* indices, int M, int plane_constraint);
/* All-gather of the row panels of Sigma + fresh partition (also done automatically, see above). */
int ekf_shard_rebalance(ekf_filter* f);

/* Raw device pointers for zero-copy plumbing (torch / RCCL): mu, the live Sigma buffer,
 * and its leading dimension (device storage is row-major, ld elements per row). */
void* ekf_device_mu(ekf_filter* f);
void* ekf_device_sigma(ekf_filter* f, int* ld);

#ifdef __cplusplus
}
#endif
#endif /* EKF_MONOSLAM_H_ */

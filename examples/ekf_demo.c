/* A plain-C client of the drop-in boundary (include/ekf_monoslam.h): no torch, no Python, no C++.
 * It walks the reference's frame loop (monoslam_ransac.cpp: imageCb -> captureNewFrame / predict / update):
 *   add features -> predict -> read predictions -> "match" (here: prediction + a fixed offset) -> update ->
 *   read the camera state and its 14 x 14 covariance -> remove a feature.
 *
 *   gcc -O2 -Iinclude examples/ekf_demo.c -o examples/ekf_demo \
 *       -Lekf-monoslam_for_3d-reconstruction_amd/lib -lekfslam_hip -Wl,-rpath,'$ORIGIN/../ekf-monoslam_for_3d-reconstruction_amd/lib' -lm
 *   ./examples/ekf_demo            # needs an MI355X; exits 0 and prints "ok" on success
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ekf_monoslam.h"

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != EKF_OK) {                                                                \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ekf_last_error(f));                 \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

int main(void) {
  ekf_config cfg;
  ekf_config_default(&cfg);
  /* mono-slam/conf/conf_kinect.cfg at scale 2 */
  cfg.sigma_vx = cfg.sigma_vy = cfg.sigma_vz = 0.03f;
  cfg.sigma_wx = cfg.sigma_wy = cfg.sigma_wz = 0.015f;
  cfg.rho_0 = 0.2f; cfg.sigma_rho_0 = 0.25f; cfg.window_size = 15; cfg.sigma_pixel = 2; cfg.scale = 2;
  cfg.fx = 268.8369f; cfg.fy = 267.1901f; cfg.u0 = 160.6130f; cfg.v0 = 124.8870f;
  cfg.k1 = 0.0395956f; cfg.k2 = -0.1113105f; cfg.k3 = 0.f; cfg.p1 = 0.00211989f; cfg.p2 = 0.00070924f;
  cfg.image_width = 320; cfg.image_height = 240;

  ekf_filter* f = NULL;
  if (ekf_abi_version() != EKF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  if (ekf_create(&cfg, 14, 64, EKF_F32, 0, &f) != EKF_OK) {
    fprintf(stderr, "ekf_create: %s\n", ekf_last_error(NULL));
    return 1;
  }
  CHECK(ekf_set_dt(f, 1.0 / 30.0));

  enum { N = 24 };
  int added = 0;
  for (int i = 0; i < N; ++i) {                       /* a 6 x 4 grid of pixels; the last one is outside the margin */
    double u = 40.0 + 48.0 * (i % 6), v = 40.0 + 50.0 * (i / 6);
    if (i == N - 1) u = 2.0;
    int rc = ekf_add_feature(f, u, v);
    if (rc < 0) { fprintf(stderr, "ekf_add_feature: %s\n", ekf_last_error(f)); return 1; }
    added += rc;
  }
  if (added != N - 1 || ekf_num_features(f) != N - 1 || ekf_state_dim(f) != 14 + 6 * (N - 1)) {
    fprintf(stderr, "unexpected map size %d\n", ekf_num_features(f));
    return 1;
  }

  float h[2 * N], S2[4 * N], z[2 * N];
  unsigned char visible[N], remove_flag[N];
  int idx[N];
  double trace_before = 0, trace_after = 0;
  for (int frame = 0; frame < 3; ++frame) {
    CHECK(ekf_predict(f, NULL, NULL, 0));
    CHECK(ekf_get_predictions(f, h, visible, remove_flag, S2, NULL, NULL));
    int M = 0;
    for (int i = 0; i < N - 1; ++i)
      if (visible[i]) {                               /* the matcher's answer: half a pixel off the prediction */
        idx[M] = i;
        z[2 * M] = h[2 * i] + 0.5f;
        z[2 * M + 1] = h[2 * i + 1] - 0.5f;
        ++M;
      }
    if (frame == 0) CHECK(ekf_covariance_parameter(f, &trace_before));
    CHECK(ekf_update(f, z, idx, M, 0));
    CHECK(ekf_synchronize(f));
  }
  CHECK(ekf_covariance_parameter(f, &trace_after));

  float mu[14], P[14 * 14];
  CHECK(ekf_get_state(f, mu, 0, 14));
  CHECK(ekf_get_sigma_block(f, P, 0, 0, 14, 14));
  double qn = sqrt((double)mu[3] * mu[3] + mu[4] * mu[4] + mu[5] * mu[5] + mu[6] * mu[6]);
  double asym = 0;
  for (int r = 0; r < 14; ++r)
    for (int c = 0; c < 14; ++c) asym = fmax(asym, fabs((double)P[c * 14 + r] - P[r * 14 + c]));
  CHECK(ekf_remove_feature(f, 3));
  if (ekf_num_features(f) != N - 2) { fprintf(stderr, "remove failed\n"); return 1; }
  printf("r = (%.5f %.5f %.5f)  |q| = %.7f  covariance parameter %.3e -> %.3e  asym %.1e\n", mu[0], mu[1], mu[2], qn,
         trace_before, trace_after, asym);
  ekf_destroy(f);
  if (!(fabs(qn - 1.0) < 1e-5) || !(asym < 1e-9) || !isfinite(trace_after)) { fprintf(stderr, "bad state\n"); return 1; }
  printf("ok\n");
  return 0;
}

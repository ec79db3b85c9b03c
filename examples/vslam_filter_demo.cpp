// The C++ mirror (include/vslam_filter_hip.hpp) driven the way the reference node drives VSlamFilter
// (monoslam_ransac.cpp imageCb: captureNewFrame -> predict -> update with its two stages -> convert2XYZ -> export).
//
//   g++ -std=c++14 -O2 -Iinclude examples/vslam_filter_demo.cpp -o examples/vslam_filter_demo
//       -Lekf-monoslam_for_3d-reconstruction_amd/lib -lekfslam_hip -Wl,-rpath,<libdir>
#include <cmath>
#include <cstdio>
#include <vector>

#include "vslam_filter_hip.hpp"

int main() {
  ekf_config cfg;
  ekf_config_default(&cfg);
  cfg.sigma_vx = cfg.sigma_vy = cfg.sigma_vz = 0.03f;
  cfg.sigma_wx = cfg.sigma_wy = cfg.sigma_wz = 0.015f;
  cfg.rho_0 = 0.2f; cfg.sigma_rho_0 = 0.25f; cfg.window_size = 15; cfg.sigma_pixel = 2; cfg.sigma_size = 2; cfg.scale = 2;
  cfg.fx = 268.8369f; cfg.fy = 267.1901f; cfg.u0 = 160.6130f; cfg.v0 = 124.8870f;
  cfg.k1 = 0.0395956f; cfg.k2 = -0.1113105f; cfg.k3 = 0.f; cfg.p1 = 0.00211989f; cfg.p2 = 0.00070924f;
  cfg.image_width = 320; cfg.image_height = 240;
  try {
    VSlamFilterHip slam(cfg, 64);
    // an 8-bit frame with some texture, so that the templates and the matcher have something to work on
    std::vector<unsigned char> frame(320 * 240);
    unsigned s = 12345u;
    for (auto& p : frame) { s = s * 1664525u + 1013904223u; p = (unsigned char)(s >> 24); }
    slam.setFrame(frame.data(), 320, 240, 320);
    for (int i = 0; i < 30; ++i) slam.addFeature(40.f + 48.f * (i % 6), 30.f + 40.f * (i / 6));
    const int N = slam.numOfFeatures();
    slam.captureNewFrame(1.0);
    slam.captureNewFrame(1.0 + 1.0 / 30.0);                  // dT = 1/30 s
    if (std::fabs(slam.getDt() - 1.0 / 30.0) > 1e-9) return 2;

    slam.predict();
    std::vector<float> h, S2, z, score;
    std::vector<unsigned char> visible, remove, found;
    slam.predictions(h, visible, remove, S2);
    std::vector<int> ell = slam.searchEllipses(cfg.sigma_size);   // drawPrediction's ellipses belong to the predictions
    slam.findMatches(z, found, score);                       // same frame: every template is found where it was cut
    std::vector<float> zm;
    std::vector<int> idx;
    for (int i = 0; i < N; ++i)
      if (found[i]) { idx.push_back(i); zm.push_back(z[2 * i]); zm.push_back(z[2 * i + 1]); }
    if ((int)idx.size() < N - 2) { std::printf("only %zu of %d matched\n", idx.size(), N); return 3; }

    // the reference's two-stage update: 1-point RANSAC -> low-innovation update -> rescue -> second update
    std::vector<int> counts;
    std::vector<unsigned char> inl;
    slam.ransac1Point(zm, idx, 2.0 * cfg.sigma_pixel, counts, inl);
    std::vector<float> z_lo, z_rest;
    std::vector<int> i_lo, i_rest;
    for (size_t k = 0; k < idx.size(); ++k) {
      auto& zz = inl[k] ? z_lo : z_rest;
      (inl[k] ? i_lo : i_rest).push_back(idx[k]);
      zz.push_back(zm[2 * k]); zz.push_back(zm[2 * k + 1]);
    }
    std::vector<float> cam = slam.getState();
    slam.update(z_lo, i_lo);
    if (!i_rest.empty()) {
      std::vector<unsigned char> hi = slam.rescueHighInnovation(cam.data(), z_rest, i_rest);
      std::vector<float> z_hi; std::vector<int> i_hi;
      for (size_t k = 0; k < i_rest.size(); ++k)
        if (hi[k]) { i_hi.push_back(i_rest[k]); z_hi.push_back(z_rest[2 * k]); z_hi.push_back(z_rest[2 * k + 1]); }
      if (!i_hi.empty()) slam.update(z_hi, i_hi);
    }
    slam.convert2XYZ_ifLinearAll();
    std::vector<float> pts = slam.getPointsFeatures();
    std::vector<float> st = slam.getState(), P = slam.getSigma();
    const double qn = std::sqrt((double)st[3] * st[3] + st[4] * st[4] + st[5] * st[5] + st[6] * st[6]);
    std::printf("N %d matched %zu low-innovation %zu  |q| %.7f  P00 %.3e  covariance parameter %.3e  points %zu x 12  ellipses %zu x 5\n",
                N, idx.size(), i_lo.size(), qn, P[0], slam.Covariance_Parameter(), pts.size() / 12, ell.size() / 5);
    if (std::fabs(qn - 1.0) > 1e-5 || !(P[0] > 0) || pts.size() != (size_t)12 * N) return 4;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf("ok\n");
  return 0;
}

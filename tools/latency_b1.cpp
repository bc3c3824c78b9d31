// Per-frame latency of the drop-in path: ONE sequence, ONE frame at a time, host pointers, PCIe included -- what a plugin's
// compute() delivers inside the unchanged pipeline (the reference times one frame at a time too: apps/app_benchmark.cpp:345-353).
//
// Two flavours over the same frames (written by bench.py: tools/latency_b1.py):
//   "adapters"  the C++ adapter classes of plugin/proslam_hip_plugin.hpp on array-of-structs clouds (the layout of the
//               reference's PointIntensityDescriptor clouds): setFixed / setMoving / compute() of the epipolar matcher, the stereo
//               adaptor's assembly loop (raw_data_preprocessor_stereo_projective.cpp:107-132), the triangulator and the aligner
//               with its circle finder.  The AoS -> SoA gather and the scatter back are inside the timed region.
//   "c_abi"     the same calls on flat arrays straight through include/proslam_hip.h (prs_stereo_match -> prs_triangulate ->
//               prs_pcf_set_fixed / set_moving -> prs_pcf_align): the boundary without the gather.
// The finder object lives across the frames of the run like the reference's (its radius / threshold schedule adapts).
// Output: one JSON line; the estimated poses of both flavours go to <frames file>.poses (float32 [2][n][16]) for the parity check.
//
//   g++ -std=c++17 -O2 -Iinclude -Iplugin tools/latency_b1.cpp -Lsrrg2_proslam_amd -lproslam_hip -o tools/bin/latency_b1
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "proslam_hip_plugin.hpp"

using namespace proslam_hip;
using Clock = std::chrono::steady_clock;

struct Frame {
  std::vector<float> uvl, uvr, xyz, X0;
  std::vector<uint8_t> dl, dr, dm;
  std::vector<uint32_t> nopt;
};

static double ms(Clock::time_point a, Clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}

template <typename T>
static void rd(std::ifstream& f, std::vector<T>& v, size_t n) {
  v.resize(n);
  f.read(reinterpret_cast<char*>(v.data()), (std::streamsize) (n * sizeof(T)));
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: latency_b1 <frames file> [warmup frames] [step_norm_exit]\n");
    return 2;
  }
  const int warm = argc > 2 ? std::atoi(argv[2]) : 4;
  const float step_norm_exit = argc > 3 ? (float) std::atof(argv[3]) : 0.0f;  // opt-in early exit (less work than the reference); 0 = off
  std::ifstream f(argv[1], std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", argv[1]);
    return 2;
  }
  int32_t head[4];
  f.read(reinterpret_cast<char*>(head), sizeof(head));
  const int n_frames = head[0], N = head[1], NM = head[2], n_par = head[3];
  std::vector<float> par;
  rd(f, par, (size_t) n_par);
  std::vector<Frame> frames((size_t) n_frames);
  for (Frame& fr : frames) {
    rd(f, fr.uvl, (size_t) N * 2);
    rd(f, fr.dl, (size_t) N * 32);
    rd(f, fr.uvr, (size_t) N * 2);
    rd(f, fr.dr, (size_t) N * 32);
    rd(f, fr.xyz, (size_t) NM * 3);
    rd(f, fr.dm, (size_t) NM * 32);
    rd(f, fr.nopt, (size_t) NM);
    rd(f, fr.X0, 16);
  }
  if (!f) {
    std::fprintf(stderr, "short frames file\n");
    return 2;
  }
  // parameter block (order fixed by tools/latency_b1.py): camera, matcher, triangulator, finder, aligner
  int k = 0;
  const float fx = par[k++], fy = par[k++], cx = par[k++], cy = par[k++], cols = par[k++], rows = par[k++], baseline_m = par[k++];
  const float range_min = par[k++], range_max = par[k++];
  const float m_dist = par[k++], m_ratio = par[k++], m_minratio = par[k++], m_disp = par[k++], m_thick = par[k++];
  const float t_mindisp = par[k++], t_inf = par[k++];
  const float f_maxd = par[k++], f_ratio = par[k++], f_minratio = par[k++], f_mind = par[k++], f_dstep = par[k++], f_maxr = par[k++], f_minr = par[k++],
              f_rstep = par[k++], f_minit = par[k++], f_eps = par[k++], f_every = par[k++];
  const float a_i0 = par[k++], a_i1 = par[k++], a_i2 = par[k++], a_chi = par[k++], a_idw = par[k++], a_damp = par[k++], a_maxit = par[k++],
              a_mininl = par[k++], a_mincorr = par[k++];
  const float b_lr_x = par[k++];  // K * t_left_in_right, x (the caller's value: a float product here would round differently from the host's double one)
  const float K9[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};

  ContextPtr ctx(new Context(0));
  std::vector<float> poses((size_t) 2 * n_frames * 16, 0.f);
  std::vector<double> t_adapt, t_abi, t_match, t_tri, t_align;

  // ---------------------------------------------------------------------------------------------- adapters (AoS clouds)
  {
    using Cloud3 = PointIntensityDescriptorVectorCloud<3>;
    using Cloud4 = PointIntensityDescriptorVectorCloud<4>;
    CorrespondenceFinderDescriptorBasedEpipolarHIP3D3D matcher(ctx);
    matcher.param_maximum_descriptor_distance.setValue(m_dist);
    matcher.param_maximum_distance_ratio_to_second_best.setValue(m_ratio);
    matcher.param_minimum_matching_ratio.setValue(m_minratio);
    matcher.param_maximum_disparity_pixels.setValue((size_t) m_disp);
    matcher.param_epipolar_line_thickness_pixels.setValue((size_t) m_thick);
    matcher.param_image_rows.setValue((size_t) rows);
    TriangulatorRigidStereoHIP tri(ctx);
    tri.param_projector.reset(new ProjectorPinholeHIP());
    tri.param_projector->setCameraMatrix(K9);
    tri.param_minimum_disparity_pixels.setValue(t_mindisp);
    tri.param_infinity_depth_meters.setValue(t_inf);
    tri.setBaselineRightInLeftMeters(baseline_m, 0, 0);
    using Finder = CorrespondenceFinderProjectiveCircleHIP<Cloud4, Cloud3>;
    AlignerProjectiveHIP<Finder> aligner(ctx);
    Finder& fd = *aligner.param_finder;
    fd.param_projector->setCameraMatrix(K9);
    fd.param_projector->param_canvas_cols.setValue((size_t) cols);
    fd.param_projector->param_canvas_rows.setValue((size_t) rows);
    fd.param_projector->param_range_min.setValue(range_min);
    fd.param_projector->param_range_max.setValue(range_max);
    fd.param_maximum_descriptor_distance.setValue(f_maxd);
    fd.param_maximum_distance_ratio_to_second_best.setValue(f_ratio);
    fd.param_minimum_matching_ratio.setValue(f_minratio);
    fd.param_minimum_descriptor_distance.setValue(f_mind);
    fd.param_descriptor_distance_step_size_pixels.setValue(f_dstep);
    fd.param_maximum_search_radius_pixels.setValue((size_t) f_maxr);
    fd.param_minimum_search_radius_pixels.setValue((size_t) f_minr);
    fd.param_search_radius_step_size_pixels.setValue((size_t) f_rstep);
    fd.param_minimum_number_of_iterations.setValue((size_t) f_minit);
    fd.param_maximum_estimate_change_norm_for_convergence.setValue(f_eps);
    fd.param_number_of_solver_iterations_per_projection.setValue((size_t) f_every);
    aligner.param_max_iterations.setValue((size_t) a_maxit);
    aligner.param_step_norm_exit.setValue(step_norm_exit);
    aligner.param_min_num_inliers.setValue((size_t) a_mininl);
    aligner.param_min_num_correspondences.setValue((size_t) a_mincorr);
    aligner.param_damping.setValue(a_damp);
    aligner.param_chi_threshold.setValue(a_chi);
    aligner.param_diagonal_info_matrix[0]        = a_i0;
    aligner.param_diagonal_info_matrix[1]        = a_i1;
    aligner.param_diagonal_info_matrix[2]        = a_i2;
    aligner.param_enable_inverse_depth_weighting = a_idw != 0.f;
    aligner.baseline_left_in_right_pixels[0]     = b_lr_x;

    // the clouds as the reference holds them: array of structs (built outside the timed region: they are the pipeline's data)
    std::vector<Cloud3> L((size_t) n_frames), R((size_t) n_frames), M((size_t) n_frames);
    for (int i = 0; i < n_frames; ++i) {
      const Frame& fr = frames[(size_t) i];
      L[i].resize((size_t) N);
      R[i].resize((size_t) N);
      M[i].resize((size_t) NM);
      for (int p = 0; p < N; ++p) {
        L[i][p].coords[0] = fr.uvl[2 * p], L[i][p].coords[1] = fr.uvl[2 * p + 1], L[i][p].coords[2] = 0.f;
        R[i][p].coords[0] = fr.uvr[2 * p], R[i][p].coords[1] = fr.uvr[2 * p + 1], R[i][p].coords[2] = 0.f;
        std::memcpy(L[i][p].descriptor_row, &fr.dl[(size_t) p * 32], 32);
        std::memcpy(R[i][p].descriptor_row, &fr.dr[(size_t) p * 32], 32);
      }
      for (int p = 0; p < NM; ++p) {
        std::memcpy(M[i][p].coords, &fr.xyz[(size_t) p * 3], 12);
        std::memcpy(M[i][p].descriptor_row, &fr.dm[(size_t) p * 32], 32);
        M[i][p].number_of_optimizations = fr.nopt[(size_t) p];
      }
    }
    CorrespondenceVector corr;
    Cloud4 meas;
    Cloud3 points;
    for (int i = -warm; i < n_frames; ++i) {
      const int u = i < 0 ? (i + warm) % n_frames : i;
      const auto t0 = Clock::now();
      matcher.setFixed(&L[u]);
      matcher.setMoving(&R[u]);
      matcher.setCorrespondences(&corr);
      matcher.compute();
      // the stereo adaptor's assembly (raw_data_preprocessor_stereo_projective.cpp:107-132)
      meas.clear();
      meas.reserve(corr.size());
      for (const Correspondence& c : corr) {
        const auto& l = L[u][(size_t) c.fixed_idx];
        const auto& r = R[u][(size_t) c.moving_idx];
        if (l.coords[0] - r.coords[0] < 0.f || l.coords[1] - r.coords[1] < 0.f) continue;
        PointIntensityDescriptor4f m4;
        m4.coords[0] = l.coords[0], m4.coords[1] = l.coords[1], m4.coords[2] = r.coords[0], m4.coords[3] = r.coords[1];
        std::memcpy(m4.descriptor_row, l.descriptor_row, 32);
        meas.push_back(m4);
      }
      const auto t1 = Clock::now();
      tri.setMoving(&meas);
      tri.setDest(&points);
      tri.compute();
      const auto t2 = Clock::now();
      aligner.setFixed(&meas);
      aligner.setMoving(&M[u]);
      aligner.setMovingInFixed(frames[(size_t) u].X0.data());
      aligner.compute();
      const auto t3 = Clock::now();
      if (i >= 0) {
        t_adapt.push_back(ms(t0, t3));
        t_match.push_back(ms(t0, t1));
        t_tri.push_back(ms(t1, t2));
        t_align.push_back(ms(t2, t3));
        std::memcpy(&poses[(size_t) i * 16], aligner.movingInFixed(), 64);
      }
    }
  }
  // ---------------------------------------------------------------------------------------------- C-ABI on flat arrays
  {
    prs_context* c = ctx->get();
    prs_stereo_params sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.maximum_descriptor_distance = m_dist, sp.maximum_distance_ratio_to_second_best = m_ratio, sp.minimum_matching_ratio = m_minratio;
    sp.maximum_disparity_pixels = (int32_t) m_disp, sp.epipolar_line_thickness_pixels = (int32_t) m_thick, sp.image_rows = (int32_t) rows;
    prs_triangulator_params tp;
    tp.fx = fx, tp.fy = fy, tp.cx = cx, tp.cy = cy, tp.b_x = fx * baseline_m, tp.minimum_disparity_pixels = t_mindisp, tp.infinity_depth_meters = t_inf;
    prs_pcf_params pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.maximum_descriptor_distance = f_maxd, pp.maximum_distance_ratio_to_second_best = f_ratio, pp.minimum_matching_ratio = f_minratio;
    pp.minimum_descriptor_distance = f_mind, pp.descriptor_distance_step_size_pixels = f_dstep;
    pp.maximum_search_radius_pixels = (uint64_t) f_maxr, pp.minimum_search_radius_pixels = (uint64_t) f_minr, pp.search_radius_step_size_pixels = (uint64_t) f_rstep;
    pp.minimum_number_of_iterations = (uint64_t) f_minit, pp.maximum_estimate_change_norm_for_convergence = f_eps;
    pp.number_of_solver_iterations_per_projection = (uint64_t) f_every, pp.search_type = PRS_SEARCH_CIRCLE;
    pp.projector.fx = fx, pp.projector.fy = fy, pp.projector.cx = cx, pp.projector.cy = cy;
    pp.projector.canvas_cols = (int32_t) cols, pp.projector.canvas_rows = (int32_t) rows, pp.projector.range_min = range_min, pp.projector.range_max = range_max;
    prs_aligner_params ap;
    std::memset(&ap, 0, sizeof(ap));
    ap.factor_type = PRS_FACTOR_STEREO, ap.fx = fx, ap.fy = fy, ap.cx = cx, ap.cy = cy, ap.image_cols = cols, ap.image_rows = rows;
    ap.baseline_left_in_right_px[0] = b_lr_x;
    ap.diagonal_info[0] = a_i0, ap.diagonal_info[1] = a_i1, ap.diagonal_info[2] = a_i2;
    ap.chi_threshold = a_chi, ap.enable_inverse_depth_weighting = a_idw != 0.f, ap.mean_disparity = -1.0f, ap.damping = a_damp;
    ap.max_iterations = (int32_t) a_maxit, ap.min_num_inliers = (int32_t) a_mininl, ap.min_num_correspondences = (int32_t) a_mincorr;
    ap.stop_at_fixed_point = 1;
    ap.step_norm_exit      = step_norm_exit;
    prs_pcf* h = nullptr;
    if (prs_pcf_create(c, &pp, &h) != PRS_OK) {
      std::fprintf(stderr, "prs_pcf_create failed\n");
      return 1;
    }
    std::vector<prs_corr> corr((size_t) N + 1), acorr((size_t) N + 1);
    std::vector<float> uvuv((size_t) N * 4), xyz((size_t) N * 3), scale((size_t) NM);
    std::vector<uint8_t> fdesc((size_t) N * 32), valid((size_t) N);
    for (int i = -warm; i < n_frames; ++i) {
      const Frame& fr = frames[(size_t) (i < 0 ? (i + warm) % n_frames : i)];
      const auto t0 = Clock::now();
      int32_t n = 0;
      int rc    = prs_stereo_match(c, &sp, reinterpret_cast<const prs_kp2*>(fr.uvl.data()), fr.dl.data(), N, reinterpret_cast<const prs_kp2*>(fr.uvr.data()),
                                   fr.dr.data(), N, corr.data(), (int32_t) corr.size(), &n);
      int m     = 0;
      for (int q = 0; q < n && rc >= 0; ++q) {
        const float ul = fr.uvl[2 * corr[q].fixed_idx], vl = fr.uvl[2 * corr[q].fixed_idx + 1];
        const float ur = fr.uvr[2 * corr[q].moving_idx], vr = fr.uvr[2 * corr[q].moving_idx + 1];
        if (ul - ur < 0.f || vl - vr < 0.f) continue;
        uvuv[4 * m] = ul, uvuv[4 * m + 1] = vl, uvuv[4 * m + 2] = ur, uvuv[4 * m + 3] = vr;
        std::memcpy(&fdesc[(size_t) m * 32], &fr.dl[(size_t) corr[q].fixed_idx * 32], 32);
        ++m;
      }
      if (rc >= 0) rc = prs_triangulate(c, &tp, uvuv.data(), m, xyz.data(), valid.data());
      prs_info_scale_from_nopt(fr.nopt.data(), NM, scale.data());
      if (rc >= 0) rc = prs_pcf_set_fixed(h, uvuv.data(), 4, fdesc.data(), m);
      if (rc >= 0) rc = prs_pcf_set_moving(h, fr.xyz.data(), scale.data(), fr.dm.data(), NM);
      float X[16];
      int32_t nc = 0;
      prs_align_result res;
      if (rc >= 0) rc = prs_pcf_align(h, &ap, fr.X0.data(), nullptr, X, acorr.data(), (int32_t) acorr.size(), &nc, &res);
      const auto t1 = Clock::now();
      if (rc < 0) {
        std::fprintf(stderr, "c_abi flavour: error %d: %s\n", rc, prs_last_error(c));
        return 1;
      }
      if (i >= 0) {
        t_abi.push_back(ms(t0, t1));
        std::memcpy(&poses[(size_t) (n_frames + i) * 16], X, 64);
      }
    }
    prs_pcf_destroy(h);
  }
  double abi_max = 0, adapt_max = 0;
  for (double x : t_abi) abi_max = std::max(abi_max, x);
  for (double x : t_adapt) adapt_max = std::max(adapt_max, x);
  auto stat = [](std::vector<double> v, double& mean, double& med, double& p95) {
    std::sort(v.begin(), v.end());
    mean = 0;
    for (double x : v) mean += x;
    mean /= (double) v.size();
    med = v[v.size() / 2];
    p95 = v[(size_t) ((double) (v.size() - 1) * 0.95)];
  };
  double am, ad, ap95, cm, cd, cp95, mm, md, mp, tm, td, tp95, lm, ld, lp;
  stat(t_adapt, am, ad, ap95);
  stat(t_abi, cm, cd, cp95);
  stat(t_match, mm, md, mp);
  stat(t_tri, tm, td, tp95);
  stat(t_align, lm, ld, lp);
  const std::string out = std::string(argv[1]) + ".poses";
  std::ofstream o(out, std::ios::binary);
  o.write(reinterpret_cast<const char*>(poses.data()), (std::streamsize) (poses.size() * sizeof(float)));
  std::printf("{\"frames\": %d, \"keypoints_per_image\": %d, \"local_map_points\": %d, "
              "\"adapters\": {\"ms_per_frame_mean\": %.4f, \"ms_per_frame_median\": %.4f, \"ms_per_frame_p95\": %.4f, \"fps\": %.1f, "
              "\"ms_matcher_incl_assembly\": %.4f, \"ms_triangulator\": %.4f, \"ms_aligner\": %.4f}, "
              "\"c_abi\": {\"ms_per_frame_mean\": %.4f, \"ms_per_frame_median\": %.4f, \"ms_per_frame_p95\": %.4f, \"fps\": %.1f}, "
              "\"ms_per_frame_max\": {\"adapters\": %.4f, \"c_abi\": %.4f}}\n",
              n_frames, N, NM, am, ad, ap95, 1000.0 / am, mm, tm, lm, cm, cd, cp95, 1000.0 / cm, adapt_max, abi_max);
  return 0;
}

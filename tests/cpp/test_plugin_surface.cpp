// test_plugin_surface.cpp -- the reference's own test shapes driven through the C++ plugin surface
// (plugin/proslam_hip_plugin.hpp) on a real GPU.  Each case names the reference test it restates.
// Built by __graft_entry__.build() with g++, run by tests/test_plugin_surface_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "proslam_hip_plugin.hpp"

using namespace proslam_hip;

static int g_failures = 0;
#define ASSERT_TRUE(cond)                                                    \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
      ++g_failures;                                                          \
      return;                                                                \
    }                                                                        \
  } while (0)
#define ASSERT_EQ(a, b) ASSERT_TRUE((a) == (b))
#define ASSERT_LT_ABS(a, b) ASSERT_TRUE(std::fabs(a) < (b))

static void randomDescriptor(std::mt19937& rng, uint8_t* d) {
  for (int i = 0; i < PRS_DESC_BYTES; ++i) d[i] = (uint8_t) (rng() & 0xff);
}

// KITTI.00To00_CorrespondenceFinderEpipolar (tests/test_correspondence_finders.cpp:152-181)
static void test_Epipolar_CloudVersusItself(ContextPtr ctx) {
  std::mt19937 rng(0);
  PointIntensityDescriptorVectorCloud<3> features(446);
  for (auto& p : features) {
    p.coords[0] = (float) (rng() % 1241);
    p.coords[1] = (float) (rng() % 376);
    p.coords[2] = 0;
    randomDescriptor(rng, p.descriptor_row);
  }
  CorrespondenceFinderDescriptorBasedEpipolarHIP3D3D finder(ctx);
  finder.param_maximum_descriptor_distance.setValue(50);
  finder.param_maximum_distance_ratio_to_second_best.setValue(0.8f);
  finder.param_image_rows.setValue(376);
  CorrespondenceVector correspondences;
  finder.setFixed(&features);
  finder.setMoving(&features);
  finder.setCorrespondences(&correspondences);
  finder.compute();
  ASSERT_EQ(correspondences.size(), features.size());
  for (const Correspondence& c : correspondences) {
    ASSERT_EQ(c.fixed_idx, c.moving_idx);
    ASSERT_EQ(c.response, 0.0f);
  }
  // unchanged inputs keep the last computation state (epipolar_impl.cpp:50-52)
  correspondences.pop_back();
  finder.compute();
  ASSERT_EQ(correspondences.size(), features.size() - 1);
}

// error contract (CF/correspondence_finder_descriptor_based_bruteforce_impl.cpp:203-216)
static void test_Epipolar_ThrowsWhenUnset(ContextPtr ctx) {
  CorrespondenceFinderDescriptorBasedEpipolarHIP3D3D finder(ctx);
  bool thrown = false;
  try {
    finder.compute();
  } catch (const std::runtime_error& e) {
    thrown = std::string(e.what()).find("fixed not set") != std::string::npos;
  }
  ASSERT_TRUE(thrown);
}

struct SyntheticWorld {
  // SyntheticWorld<3,float,...>::generateMap(100, mean(10,10,10), dev(5,5,5), Camera), K = (200,200,100,100), canvas 1000x1000
  PointIntensityDescriptorVectorCloud<3> points_in_world;
  float K[9] = {200, 0, 100, 0, 200, 100, 0, 0, 1};
  explicit SyntheticWorld(unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> n(10.0f, 5.0f);
    points_in_world.resize(100);
    for (auto& p : points_in_world) {
      p.coords[0] = n(rng);
      p.coords[1] = n(rng);
      p.coords[2] = n(rng);
      randomDescriptor(rng, p.descriptor_row);
    }
  }
  // project world points seen from a camera at `cam_in_world` = (R | t), rows of R given
  template <int Dim>
  void project(const float* R9, const float* t3, PointIntensityDescriptorVectorCloud<Dim>& out, std::vector<int>& truth) const {
    out.clear();
    truth.clear();
    for (size_t i = 0; i < points_in_world.size(); ++i) {
      const float* w = points_in_world[i].coords;
      const float d[3] = {w[0] - t3[0], w[1] - t3[1], w[2] - t3[2]};
      float c[3];
      for (int r = 0; r < 3; ++r) c[r] = R9[r] * d[0] + R9[3 + r] * d[1] + R9[6 + r] * d[2];  // R^T d
      if (c[2] < 0.1f || c[2] > 1000.0f) continue;
      const float u = K[0] * c[0] / c[2] + K[2], v = K[4] * c[1] / c[2] + K[5];
      if (u < 0 || u >= 1000 || v < 0 || v >= 1000) continue;
      PointIntensityDescriptor_<Dim> p;
      p.coords[0] = u;
      p.coords[1] = v;
      if (Dim > 2) p.coords[2] = c[2];
      std::memcpy(p.descriptor_row, points_in_world[i].descriptor_row, PRS_DESC_BYTES);
      out.push_back(p);
      truth.push_back((int) i);
    }
  }
};

// SyntheticWorldWithDescriptorsSE3.ProjectiveCircle_NoMotionNoNoise (tests/test_correspondence_finders.cpp:1024-1080)
static void test_ProjectiveCircle_NoMotionNoNoise(ContextPtr ctx) {
  SyntheticWorld world(1);
  const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
  PointIntensityDescriptorVectorCloud<2> points_in_canvas;
  std::vector<int> truth;
  world.project<2>(I9, zero, points_in_canvas, truth);
  CorrespondenceFinderProjectiveCircleHIP2D3D finder(ctx);
  finder.param_minimum_descriptor_distance.setValue(25);
  finder.param_maximum_descriptor_distance.setValue(75);
  finder.param_maximum_distance_ratio_to_second_best.setValue(0.5f);
  finder.param_minimum_search_radius_pixels.setValue(1);
  finder.param_projector->param_canvas_cols.setValue(1000);
  finder.param_projector->param_canvas_rows.setValue(1000);
  finder.param_projector->param_range_min.setValue(0.1f);
  finder.param_projector->param_range_max.setValue(1000);
  finder.param_projector->setCameraMatrix(world.K);
  const float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  finder.setLocalMapInSensor(T);
  CorrespondenceVector correspondences;
  finder.setFixed(&points_in_canvas);
  finder.setMoving(&world.points_in_world);
  finder.setCorrespondences(&correspondences);
  finder.compute();
  ASSERT_TRUE(points_in_canvas.size() > 40);
  ASSERT_EQ(correspondences.size(), points_in_canvas.size());
  for (const Correspondence& c : correspondences) ASSERT_EQ(c.moving_idx, truth[(size_t) c.fixed_idx]);
  // the object is stateful across calls (CF/..projective_base.h:132-154)
  ASSERT_EQ(finder.state().current_iteration, (uint64_t) 1);
}

// SyntheticWorldWithDescriptorsSE3.AlignerSliceProcessorProjective (tests/test_aligners.cpp:15-140)
static void test_AlignerSliceProcessorProjective(ContextPtr ctx) {
  SyntheticWorld world(2);
  // sensor pose 1: translation (0,0,-1), rotation a2r(0.001, 0.001, -0.001) ~ I + [w]x
  const float a = 0.001f, b = 0.001f, c = -0.001f;
  const float R9[9] = {1, -c, b, c, 1, -a, -b, a, 1};  // first-order a2r, rows
  const float t3[3] = {0, 0, -1};
  PointIntensityDescriptorVectorCloud<2> points_in_camera_fixed;
  std::vector<int> truth;
  world.project<2>(R9, t3, points_in_camera_fixed, truth);
  using Finder = CorrespondenceFinderProjectiveKDTreeHIP<PointIntensityDescriptorVectorCloud<2>, PointIntensityDescriptorVectorCloud<3>>;
  AlignerProjectiveHIP<Finder> aligner(ctx);
  Finder& finder = *aligner.param_finder;
  finder.param_maximum_descriptor_distance.setValue(75);
  finder.param_minimum_descriptor_distance.setValue(25);
  finder.param_maximum_distance_ratio_to_second_best.setValue(0.5f);
  finder.param_maximum_search_radius_pixels.setValue(50);
  finder.param_projector->param_canvas_cols.setValue(1000);
  finder.param_projector->param_canvas_rows.setValue(1000);
  finder.param_projector->param_range_min.setValue(0.1f);
  finder.param_projector->param_range_max.setValue(1000);
  finder.param_projector->setCameraMatrix(world.K);
  aligner.param_max_iterations.setValue(10);
  aligner.setFixed(&points_in_camera_fixed);
  aligner.setMoving(&world.points_in_world);
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  aligner.setMovingInFixed(I16);
  ASSERT_EQ(aligner.status(), AlignerProjectiveHIP<Finder>::Fail);
  aligner.compute();
  ASSERT_EQ(aligner.status(), AlignerProjectiveHIP<Finder>::Success);
  // error = t2tnq(movingInFixed * pose) (test_aligners.cpp:132): translation part and rotation residual
  const float* X = aligner.movingInFixed();
  float E[9], et[3];
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) E[3 * r + k] = X[4 * r] * R9[k] + X[4 * r + 1] * R9[3 + k] + X[4 * r + 2] * R9[6 + k];
    et[r] = X[4 * r] * t3[0] + X[4 * r + 1] * t3[1] + X[4 * r + 2] * t3[2] + X[4 * r + 3];
  }
  ASSERT_LT_ABS(et[0], 0.15f);
  ASSERT_LT_ABS(et[1], 0.15f);
  ASSERT_LT_ABS(et[2], 0.15f);
  // quaternion imaginary part ~ half the skew part of the residual rotation
  ASSERT_LT_ABS(0.25f * (E[7] - E[5]), 0.005f);
  ASSERT_LT_ABS(0.25f * (E[2] - E[6]), 0.005f);
  ASSERT_LT_ABS(0.25f * (E[3] - E[1]), 0.005f);
  ASSERT_TRUE(aligner.correspondences().size() > 40);
}

// the readings of the un-vendored srrg2_solver arithmetic as PARAMs of the aligner adapter (include/proslam_hip.h, prs_aligner_params):
// the scenario above converges under each of them to the same bounds, the estimates differ in their last bits, and a value the
// library does not know is refused loudly (the adapter throws like the reference's objects do on a bad configuration)
static void test_AlignerSolverArithmeticParams(ContextPtr ctx) {
  SyntheticWorld world(2);
  const float a = 0.001f, b = 0.001f, c = -0.001f;
  const float R9[9] = {1, -c, b, c, 1, -a, -b, a, 1};
  const float t3[3] = {0, 0, -1};
  PointIntensityDescriptorVectorCloud<2> fixed;
  std::vector<int> truth;
  world.project<2>(R9, t3, fixed, truth);
  using Finder = CorrespondenceFinderProjectiveKDTreeHIP<PointIntensityDescriptorVectorCloud<2>, PointIntensityDescriptorVectorCloud<3>>;
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float first[16] = {0};
  bool any_differs = false;
  for (int form = 0; form < 3; ++form) {  // 0: shipped, 1: tau / chi kernel weight, 2: H + lambda I
    AlignerProjectiveHIP<Finder> aligner(ctx);
    Finder& finder = *aligner.param_finder;
    finder.param_maximum_descriptor_distance.setValue(75);
    finder.param_minimum_descriptor_distance.setValue(25);
    finder.param_maximum_distance_ratio_to_second_best.setValue(0.5f);
    finder.param_maximum_search_radius_pixels.setValue(50);
    finder.param_projector->param_canvas_cols.setValue(1000);
    finder.param_projector->param_canvas_rows.setValue(1000);
    finder.param_projector->param_range_min.setValue(0.1f);
    finder.param_projector->param_range_max.setValue(1000);
    finder.param_projector->setCameraMatrix(world.K);
    aligner.param_max_iterations.setValue(40);
    aligner.param_damping.setValue(0.1f);
    aligner.param_chi_threshold.setValue(25.0f);  // (5 px: the saturated kernel is active in the first iterations)
    aligner.param_robustifier_kernel_weight_form.setValue(form == 1 ? PRS_KERNEL_WEIGHT_TAU_OVER_CHI : PRS_KERNEL_WEIGHT_INV_CHI);
    aligner.param_damping_form.setValue(form == 2 ? PRS_DAMPING_IDENTITY : PRS_DAMPING_DIAG);
    aligner.setFixed(&fixed);
    aligner.setMoving(&world.points_in_world);
    aligner.setMovingInFixed(I16);
    aligner.compute();
    ASSERT_EQ(aligner.status(), AlignerProjectiveHIP<Finder>::Success);
    const float* X = aligner.movingInFixed();
    for (int r = 0; r < 3; ++r) {
      const float et = X[4 * r] * t3[0] + X[4 * r + 1] * t3[1] + X[4 * r + 2] * t3[2] + X[4 * r + 3];
      ASSERT_LT_ABS(et, 0.15f);
    }
    if (form == 0) {
      std::memcpy(first, X, sizeof(first));
    } else {
      any_differs = any_differs || std::memcmp(first, X, sizeof(first)) != 0;
    }
  }
  ASSERT_TRUE(any_differs);  // the switches reach the kernels
  AlignerProjectiveHIP<Finder> bad(ctx);
  bad.param_finder->param_projector->param_canvas_cols.setValue(1000);
  bad.param_finder->param_projector->param_canvas_rows.setValue(1000);
  bad.param_finder->param_projector->setCameraMatrix(world.K);
  bad.param_damping_form.setValue(7);
  bad.setFixed(&fixed);
  bad.setMoving(&world.points_in_world);
  bad.setMovingInFixed(I16);
  bool thrown = false;
  try {
    bad.compute();
  } catch (const std::exception&) {
    thrown = true;
  }
  ASSERT_TRUE(thrown);
}

// SyntheticWorldWithDescriptorsSE3.AlignerSliceProcessorProjectiveDepthWithSensor (tests/test_aligners.cpp:281-426):
// sensor_in_robot = ((0.2, 0.3, 0.4), a2r(0, 0.05 pi, 0)), robot 1 at (0, 0, -1); the estimate is the ROBOT's motion.
// The MultiAligner3DQR flags of icl.conf (inlier-only runs, keep only inliers) are switched on as well.
static void test_AlignerSliceProcessorProjectiveDepthWithSensor(ContextPtr ctx) {
  SyntheticWorld world(3);
  const float ang = 0.05f * 3.14159265358979f, cs = std::cos(ang), sn = std::sin(ang);
  const float R9[9] = {cs, 0, sn, 0, 1, 0, -sn, 0, cs};                      // rotation about y, rows
  const float S16[16] = {cs, 0, sn, 0.2f, 0, 1, 0, 0.3f, -sn, 0, cs, 0.4f, 0, 0, 0, 1};
  const float t3[3] = {0.2f, 0.3f, 0.4f - 1.0f};                              // sensor 1 in world = pose * sensor_in_robot
  PointIntensityDescriptorVectorCloud<3> points_in_camera_fixed;
  std::vector<int> truth;
  world.project<3>(R9, t3, points_in_camera_fixed, truth);
  using Finder = CorrespondenceFinderProjectiveKDTreeHIP<PointIntensityDescriptorVectorCloud<3>, PointIntensityDescriptorVectorCloud<3>>;
  AlignerProjectiveHIP<Finder> aligner(ctx);
  Finder& finder = *aligner.param_finder;
  finder.param_maximum_descriptor_distance.setValue(75);
  finder.param_minimum_descriptor_distance.setValue(25);
  finder.param_maximum_distance_ratio_to_second_best.setValue(0.5f);
  finder.param_maximum_search_radius_pixels.setValue(50);
  finder.param_projector->param_canvas_cols.setValue(1000);
  finder.param_projector->param_canvas_rows.setValue(1000);
  finder.param_projector->param_range_min.setValue(0.1f);
  finder.param_projector->param_range_max.setValue(1000);
  finder.param_projector->setCameraMatrix(world.K);
  aligner.param_max_iterations.setValue(10);
  aligner.param_enable_inlier_only_runs.setValue(true);
  aligner.param_keep_only_inlier_correspondences.setValue(true);
  aligner.setSensorInRobot(S16);
  aligner.setFixed(&points_in_camera_fixed);
  aligner.setMoving(&world.points_in_world);
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  aligner.setMovingInFixed(I16);
  aligner.compute();
  ASSERT_EQ(aligner.status(), AlignerProjectiveHIP<Finder>::Success);
  ASSERT_EQ(aligner.result().iterations, 20);  // 10 + the inlier-only run
  // movingInFixed * pose ~ identity with pose = (I | (0, 0, -1)): X ~ (I | (0, 0, 1))
  const float* X = aligner.movingInFixed();
  ASSERT_LT_ABS(X[3], 0.15f);
  ASSERT_LT_ABS(X[7], 0.15f);
  ASSERT_LT_ABS(X[11] - 1.0f, 0.15f);
  ASSERT_LT_ABS(0.25f * (X[9] - X[6]), 0.005f);
  ASSERT_LT_ABS(0.25f * (X[2] - X[8]), 0.005f);
  ASSERT_LT_ABS(0.25f * (X[4] - X[1]), 0.005f);
  ASSERT_TRUE(aligner.correspondences().size() > 40);
  ASSERT_EQ((int) aligner.correspondences().size(), aligner.result().num_inliers);
}

// triangulate(project(p)) = p and size preservation (tests/fixtures.hpp:939-944, triangulator_rigid_stereo.cpp:39-55)
static void test_TriangulatorRigidStereo(ContextPtr ctx) {
  TriangulatorRigidStereoHIP triangulator(ctx);
  triangulator.param_projector.reset(new ProjectorPinholeHIP());
  const float K[9] = {718.856f, 0, 607.193f, 0, 718.856f, 185.216f, 0, 0, 1};  // tests/fixtures.hpp:810
  triangulator.param_projector->setCameraMatrix(K);
  triangulator.setBaselineRightInLeftMeters(0.537166f, 0, 0);  // tests/fixtures.hpp:816
  PointIntensityDescriptorVectorCloud<4> matches(3);
  const float P[2][3] = {{1.0f, -0.5f, 10.0f}, {-3.0f, 0.7f, 25.0f}};
  for (int i = 0; i < 2; ++i) {
    matches[i].coords[0] = K[0] * P[i][0] / P[i][2] + K[2];
    matches[i].coords[1] = K[4] * P[i][1] / P[i][2] + K[5];
    matches[i].coords[2] = matches[i].coords[0] - K[0] * 0.537166f / P[i][2];
    matches[i].coords[3] = matches[i].coords[1];
  }
  matches[2].coords[0] = 100;
  matches[2].coords[1] = 50;
  matches[2].coords[2] = 99.5f;  // disparity below the 1 px minimum -> Invalid, slot kept
  matches[2].coords[3] = 50;
  PointIntensityDescriptorVectorCloud<3> points;
  triangulator.setMoving(&matches);
  triangulator.setDest(&points);
  triangulator.compute();
  ASSERT_EQ(points.size(), matches.size());
  ASSERT_LT_ABS(triangulator.baselineRigthInLeft()[0] - 386.1448f, 1e-2f);  // tests/fixtures.hpp:811
  for (int i = 0; i < 2; ++i) {
    ASSERT_TRUE(points[i].valid);
    for (int k = 0; k < 3; ++k) ASSERT_LT_ABS(points[i].coords[k] - P[i][k], 2e-3f * P[i][2]);
  }
  ASSERT_TRUE(!points[2].valid);
  ASSERT_EQ(triangulator.indicesInvalidated().size(), (size_t) 1);
}

// SceneClipperProjective3D: no motion keeps the cloud, half roll sees nothing, unset buffers throw
// (tests/test_scene_clippers.cpp:186-219,257-283; scene_clipper_projective_3d.cpp:12-28)
static void test_SceneClipperProjective3D(ContextPtr ctx) {
  SyntheticWorld world(3);
  SceneClipperProjective3DHIP clipper(ctx);
  clipper.param_projector->setCameraMatrix(world.K);
  clipper.param_projector->param_canvas_cols.setValue(1000);
  clipper.param_projector->param_canvas_rows.setValue(1000);
  clipper.param_projector->param_range_max.setValue(1000);
  clipper.param_projector->param_range_min.setValue(0.1f);
  bool thrown = false;
  try {
    clipper.compute();
  } catch (const std::runtime_error&) {
    thrown = true;
  }
  ASSERT_TRUE(thrown);
  PointIntensityDescriptorVectorCloud<3> points_visible;
  clipper.setFullScene(&world.points_in_world);
  clipper.setClippedSceneInRobot(&points_visible);
  clipper.compute();
  const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
  PointIntensityDescriptorVectorCloud<2> expected;
  std::vector<int> truth;
  world.project<2>(I9, zero, expected, truth);
  ASSERT_EQ((int) clipper.status(), (int) SceneClipperProjective3DHIP::Successful);
  ASSERT_EQ(points_visible.size(), expected.size());
  ASSERT_EQ(clipper.globalIndices().size(), expected.size());
  for (size_t k = 0; k < points_visible.size(); ++k) {
    ASSERT_EQ(clipper.globalIndices()[k], truth[k]);
    ASSERT_TRUE(points_visible[k].coords[2] > 0);
    ASSERT_TRUE(std::memcmp(points_visible[k].descriptor_row, world.points_in_world[(size_t) truth[k]].descriptor_row, PRS_DESC_BYTES) == 0);
  }
  const float roll[16] = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};  // AngleAxisf(M_PI, UnitX)
  clipper.setRobotInLocalMap(roll);
  clipper.compute();
  ASSERT_EQ(points_visible.size(), (size_t) 0);
  // empty full scene: status Ready, previous result untouched
  PointIntensityDescriptorVectorCloud<3> nothing;
  points_visible.resize(5);
  clipper.setFullScene(&nothing);
  clipper.compute();
  ASSERT_EQ((int) clipper.status(), (int) SceneClipperProjective3DHIP::Ready);
  ASSERT_EQ(points_visible.size(), (size_t) 5);
}

// SyntheticWorldWithDescriptorsSE3.Bruteforce_CloudVersusItself (tests/test_correspondence_finders.cpp:139-181)
static void test_Bruteforce_CloudVersusItself(ContextPtr ctx) {
  SyntheticWorld world(7);
  CorrespondenceFinderDescriptorBasedBruteforceHIP3D3D finder(ctx);
  CorrespondenceVector correspondences;
  bool thrown = false;
  try {
    finder.compute();
  } catch (const std::runtime_error&) {
    thrown = true;
  }
  ASSERT_TRUE(thrown);
  finder.setFixed(&world.points_in_world);
  finder.setMoving(&world.points_in_world);
  finder.setCorrespondences(&correspondences);
  finder.compute();
  ASSERT_EQ(correspondences.size(), world.points_in_world.size());
  for (const Correspondence& c : correspondences) {
    ASSERT_EQ(c.fixed_idx, c.moving_idx);
    ASSERT_TRUE(c.response == 0.0f);
  }
  // neither input changed: the last result is kept (bruteforce_impl.cpp:12-14)
  correspondences.resize(3);
  finder.compute();
  ASSERT_EQ(correspondences.size(), (size_t) 3);
}

// KITTI 00_FAST_ORB256 / IntensityFeatureExtractorBinned (tests/test_feature_extractors.cpp:7-262) shape on a synthetic image: features
// come back inside the descriptor border, strongest first per region, and the extractor's output feeds the epipolar finder
static void test_IntensityFeatureExtractorBinned(ContextPtr ctx) {
  const int rows = 376, cols = 1241;
  std::vector<uint8_t> image((size_t) rows * cols);
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      // blocks of random brightness (corners of every strength) + a small per-pixel pattern: on perfectly flat blocks the
      // corner responses form plateaus and the strict non-maximum suppression leaves nothing
      std::mt19937 cell((uint32_t) ((r / 12) * 977 + (c / 12)));
      const int v = (int) (cell() & 0xff) + ((r * 7919 + c * 104729 + (r * c) % 31) % 13) - 6;
      image[(size_t) r * cols + c] = (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  IntensityFeatureExtractorBinnedHIP extractor(ctx);
  bool thrown = false;
  try {
    extractor.compute(image.data(), rows, cols, cols);  // target feature buffer not set
  } catch (const std::runtime_error&) {
    thrown = true;
  }
  ASSERT_TRUE(thrown);
  IntensityFeatureExtractorBinnedHIP::PointCloudType features, again;
  extractor.param_detector_threshold.setValue(15);
  extractor.param_target_number_of_keypoints.setValue(500);
  extractor.param_number_of_detectors_vertical.setValue(3);
  extractor.param_number_of_detectors_horizontal.setValue(3);
  extractor.setFeatures(&features);
  extractor.compute(image.data(), rows, cols, cols);
  ASSERT_TRUE(features.size() > 200 && features.size() <= 500);
  for (const auto& p : features) {
    ASSERT_TRUE(p.coords[0] >= 31 && p.coords[0] < cols - 31 && p.coords[1] >= 31 && p.coords[1] < rows - 31);  // cv::ORB border
    ASSERT_EQ(p.intensity_value, (float) image[(size_t) p.coords[1] * cols + (size_t) p.coords[0]]);
  }
  extractor.setFeatures(&again);
  extractor.compute(image.data(), rows, cols, cols);  // deterministic
  ASSERT_EQ(again.size(), features.size());
  // the features of an image against themselves through the epipolar finder: mirror matches (test_correspondence_finders.cpp:152-181)
  PointIntensityDescriptorVectorCloud<3> cloud(features.size());
  for (size_t i = 0; i < features.size(); ++i) {
    ASSERT_TRUE(std::memcmp(again[i].descriptor_row, features[i].descriptor_row, PRS_DESC_BYTES) == 0 && again[i].coords[0] == features[i].coords[0]);
    cloud[i].coords[0] = features[i].coords[0];
    cloud[i].coords[1] = features[i].coords[1];
    cloud[i].coords[2] = 0;
    std::memcpy(cloud[i].descriptor_row, features[i].descriptor_row, PRS_DESC_BYTES);
  }
  CorrespondenceFinderDescriptorBasedEpipolarHIP3D3D finder(ctx);
  finder.param_maximum_descriptor_distance.setValue(50);
  finder.param_maximum_distance_ratio_to_second_best.setValue(0.8f);
  finder.param_image_rows.setValue(rows);
  CorrespondenceVector correspondences;
  finder.setFixed(&cloud);
  finder.setMoving(&cloud);
  finder.setCorrespondences(&correspondences);
  finder.compute();
  ASSERT_EQ(correspondences.size(), cloud.size());
  for (const Correspondence& c : correspondences) {
    ASSERT_EQ(c.fixed_idx, c.moving_idx);
    ASSERT_EQ(c.response, 0.0f);
  }
}

// KITTI 00To00 / 00To01_MergerTriangulation_WeightedMean (tests/test_mergers.cpp:357-405, :464-519) on a synthetic stereo scene:
// merging a cloud with its own measurements leaves it untouched, merging the view from a moved camera with partial
// correspondences grows it by no more than the measurements, and unset buffers throw
static void test_MergerRigidStereoTriangulation(ContextPtr ctx) {
  const float K[9] = {718.856f, 0, 607.193f, 0, 718.856f, 185.216f, 0, 0, 1};
  const float bx   = 386.1448f;
  std::mt19937 rng(11);
  std::uniform_real_distribution<float> ux(-8.f, 8.f), uy(-2.f, 1.5f), uz(6.f, 40.f);
  PointIntensityDescriptorVectorCloud<3> scene;
  PointIntensityDescriptorVectorCloud<4> measurements;
  while (scene.size() < 150) {
    const float X = ux(rng), Y = uy(rng), Z = uz(rng);
    const float uL = std::round(K[0] * X / Z + K[2]), v = std::round(K[4] * Y / Z + K[5]), uR = std::round(uL - bx / Z);
    if (uL < 1 || uL >= 1240 || v < 1 || v >= 375 || uL - uR < 1.f) continue;
    PointIntensityDescriptor_<4> m;
    m.coords[0] = uL;
    m.coords[1] = v;
    m.coords[2] = uR;
    m.coords[3] = v;
    randomDescriptor(rng, m.descriptor_row);
    measurements.push_back(m);
    // the scene point is what the triangulator makes of this measurement (fixtures.hpp:938-944)
    const float d = uL - uR, z = bx / d;
    PointIntensityDescriptor_<3> p;
    p.coords[0] = (uL - K[2]) / K[0] * z;
    p.coords[1] = (v - K[5]) / K[4] * z;
    p.coords[2] = z;
    std::memcpy(p.descriptor_row, m.descriptor_row, PRS_DESC_BYTES);
    scene.push_back(p);
  }
  MergerRigidStereoTriangulationHIP merger(ctx);
  merger.param_projector->setCameraMatrix(K);
  merger.param_projector->param_canvas_rows.setValue(376);
  merger.param_projector->param_canvas_cols.setValue(1241);
  merger.setBaselineRightInLeftPixels(bx);
  merger.param_maximum_distance_appearance.setValue(50);
  merger.param_maximum_distance_geometry_meters_squared.setValue(25);
  bool thrown = false;
  try {
    merger.compute();
  } catch (const std::runtime_error&) {
    thrown = true;
  }
  ASSERT_TRUE(thrown);
  CorrespondenceVector mirror;
  for (size_t i = 0; i < scene.size(); ++i) mirror.push_back(Correspondence{(int) i, (int) i, 0.f});
  const PointIntensityDescriptorVectorCloud<3> backup(scene);
  merger.setScene(&scene);
  merger.setMeasurement(&measurements);
  merger.setCorrespondences(&mirror);
  merger.compute();
  ASSERT_EQ(scene.size(), backup.size());  // :392
  ASSERT_TRUE(merger.numberOfMergedPoints() > 0 && merger.numberOfAddedPoints() == 0);
  for (size_t i = 0; i < backup.size(); ++i) {  // :395-402: element order intact, coordinates within 1e-5 (relative to the depth here)
    for (int k = 0; k < 3; ++k) ASSERT_LT_ABS(scene[i].coords[k] - backup[i].coords[k], 1e-5f * backup[i].coords[2] + 1e-5f);
  }
  // a second frame: the same measurements, every other correspondence missing -> unmatched measurements may be added
  CorrespondenceVector half;
  for (size_t i = 0; i < mirror.size(); i += 2) half.push_back(mirror[i]);
  merger.setCorrespondences(&half);
  merger.compute();
  ASSERT_TRUE(scene.size() >= backup.size() && scene.size() <= backup.size() + measurements.size());  // :503-506
  for (size_t i = 0; i < backup.size(); ++i) {
    ASSERT_TRUE(std::memcmp(scene[i].descriptor_row, backup[i].descriptor_row, PRS_DESC_BYTES) == 0);
  }
}

// a map that has to grow between two merges keeps its landmark statistics (the device copy is the master after the first
// upload): two mergers on the same frames, one of them grown in place between the frames -- same scene afterwards.  (State,
// covariance and measurement history across prs_map_reserve are compared bit for bit, with world != scene, in
// tests/test_ref_mapping_gpu.py.)
static void test_MergerGrowsInPlace(ContextPtr ctx) {
  const float K[9] = {718.856f, 0, 607.193f, 0, 718.856f, 185.216f, 0, 0, 1};
  const float bx   = 386.1448f;
  std::mt19937 rng(12);
  std::uniform_real_distribution<float> ux(-8.f, 8.f), uy(-2.f, 1.5f), uz(6.f, 40.f);
  PointIntensityDescriptorVectorCloud<3> scene_a, scene_b;
  PointIntensityDescriptorVectorCloud<4> measurements;
  while (scene_a.size() < 120) {
    const float X = ux(rng), Y = uy(rng), Z = uz(rng);
    const float uL = std::round(K[0] * X / Z + K[2]), v = std::round(K[4] * Y / Z + K[5]), uR = std::round(uL - bx / Z);
    if (uL < 1 || uL >= 1240 || v < 1 || v >= 375 || uL - uR < 1.f) continue;
    PointIntensityDescriptor_<4> m;
    m.coords[0] = uL;
    m.coords[1] = v;
    m.coords[2] = uR;
    m.coords[3] = v;
    randomDescriptor(rng, m.descriptor_row);
    measurements.push_back(m);
    const float d = uL - uR, z = bx / d;
    PointIntensityDescriptor_<3> p;
    p.coords[0] = (uL - K[2]) / K[0] * z + 0.05f;  // (not exactly the triangulated point: the weighted mean has something to do)
    p.coords[1] = (v - K[5]) / K[4] * z;
    p.coords[2] = z + 0.1f;
    std::memcpy(p.descriptor_row, m.descriptor_row, PRS_DESC_BYTES);
    scene_a.push_back(p);
  }
  scene_b = scene_a;
  float in_scene[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float in_world[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  CorrespondenceVector mirror, half;
  for (size_t i = 0; i < scene_a.size(); ++i) mirror.push_back(Correspondence{(int) i, (int) i, 0.f});
  for (size_t i = 0; i < mirror.size(); i += 2) half.push_back(mirror[i]);
  MergerRigidStereoTriangulationHIP a(ctx), b(ctx);
  MergerRigidStereoTriangulationHIP* both[2] = {&a, &b};
  PointIntensityDescriptorVectorCloud<3>* scenes[2] = {&scene_a, &scene_b};
  for (int k = 0; k < 2; ++k) {
    MergerRigidStereoTriangulationHIP& m = *both[k];
    m.param_projector->setCameraMatrix(K);
    m.param_projector->param_canvas_rows.setValue(376);
    m.param_projector->param_canvas_cols.setValue(1241);
    m.setBaselineRightInLeftPixels(bx);
    m.param_maximum_distance_appearance.setValue(50);
    m.param_maximum_distance_geometry_meters_squared.setValue(25);
    m.setMeasurementInScene(in_scene);
    m.setMeasurementInWorld(in_world);
    m.setScene(scenes[k]);
    m.setMeasurement(&measurements);
    m.setCorrespondences(&mirror);
    m.compute();
    ASSERT_TRUE(m.numberOfMergedPoints() > 0);
    if (k == 1) m.reserve(20000);  // grow the device map between the frames
    m.setCorrespondences(&half);
    m.compute();
    m.compute();
  }
  ASSERT_EQ(scene_a.size(), scene_b.size());
  bool moved = false;
  for (size_t i = 0; i < scene_a.size() && i < scene_b.size(); ++i) {
    for (int c = 0; c < 3; ++c) {
      ASSERT_TRUE(scene_a[i].coords[c] == scene_b[i].coords[c]);
    }
    ASSERT_TRUE(scene_a[i].number_of_optimizations == scene_b[i].number_of_optimizations);
    moved = moved || scene_a[i].number_of_optimizations > 1;
  }
  ASSERT_TRUE(moved);  // the landmarks were refined across the frames (their world states were used after the growth)
}

int main() {
  ContextPtr ctx;
  try {
    ctx.reset(new Context(0));
  } catch (const std::exception& e) {
    std::printf("no device: %s\n", e.what());
    return 2;
  }
#define RUN(t)                                     \
  do {                                             \
    const int before = g_failures;                 \
    t(ctx);                                        \
    std::printf("[%s] %s\n", g_failures == before ? "  OK  " : "FAILED", #t); \
  } while (0)
  RUN(test_Epipolar_CloudVersusItself);
  RUN(test_Epipolar_ThrowsWhenUnset);
  RUN(test_ProjectiveCircle_NoMotionNoNoise);
  RUN(test_AlignerSliceProcessorProjective);
  RUN(test_AlignerSolverArithmeticParams);
  RUN(test_AlignerSliceProcessorProjectiveDepthWithSensor);
  RUN(test_TriangulatorRigidStereo);
  RUN(test_SceneClipperProjective3D);
  RUN(test_Bruteforce_CloudVersusItself);
  RUN(test_IntensityFeatureExtractorBinned);
  RUN(test_MergerRigidStereoTriangulation);
  RUN(test_MergerGrowsInPlace);
  std::printf("%d failure(s)\n", g_failures);
  return g_failures ? 1 : 0;
}

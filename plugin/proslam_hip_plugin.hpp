// proslam_hip_plugin.hpp -- C++ host side above the C-ABI (include/proslam_hip.h), mirroring the
// reference's operator surface for the tracking hot path: same member names, argument meaning and
// error behaviour as the srrg2 classes it stands in for, so call sites (and tests) read like the
// reference's own.  Header-only, C++11, depends only on libproslam_hip.so.
//
//   reference class (srrg2_proslam)                               -> class here
//   CorrespondenceFinderDescriptorBasedEpipolar<..>                -> CorrespondenceFinderDescriptorBasedEpipolarHIP
//     (CF/correspondence_finder_descriptor_based_epipolar.h:8-47)
//   CorrespondenceFinderDescriptorBasedBruteforce<..>              -> CorrespondenceFinderDescriptorBasedBruteforceHIP
//     (CF/correspondence_finder_descriptor_based_bruteforce.h:8-95)
//   CorrespondenceFinderProjective{KDTree,Square,Circle,Rhombus}   -> CorrespondenceFinderProjectiveHIP<SEARCH>
//     (CF/correspondence_finder_projective_base.h:14-155)
//   TriangulatorRigidStereo (mapping/triangulator_rigid_stereo.h)  -> TriangulatorRigidStereoHIP
//   SceneClipperProjective3D (mapping/scene_clipper_projective_3d.h) -> SceneClipperProjective3DHIP
//   MultiAligner3DQR + AlignerSliceProcessorProjective*            -> AlignerProjectiveHIP
//     (registration/aligner_slice_processor_projective.h:14-192, tests/test_aligners.cpp:1237-1253)
//
// When the srrg2 headers are available the same bodies become real plugin subclasses: see
// INTEGRATION.md for the BOSS_REGISTER_CLASS adapters.  Points are AoS like the reference's
// PointIntensityDescriptor_<Dim> (coordinates, intensity, 32-byte descriptor row); the adapters
// gather them into the SoA layout the C-ABI takes.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "proslam_hip.h"

namespace proslam_hip {

// ---- the data model the reference's clouds reduce to on this path -------------------------------
template <int Dim_>
struct PointIntensityDescriptor_ {
  static constexpr int Dim = Dim_;
  float coords[Dim_];
  float intensity_value = 0.f;
  uint8_t descriptor_row[PRS_DESC_BYTES];  // cv::Mat 1x32 CV_8U in the reference
  uint32_t number_of_optimizations = 0;    // statistics().numberOfOptimizations()
  bool valid                        = true; // status == Valid
  float* coordinates() { return coords; }
  const float* coordinates() const { return coords; }
  uint8_t* descriptor() { return descriptor_row; }
  const uint8_t* descriptor() const { return descriptor_row; }
};
using PointIntensityDescriptor2f = PointIntensityDescriptor_<2>;
using PointIntensityDescriptor3f = PointIntensityDescriptor_<3>;
using PointIntensityDescriptor4f = PointIntensityDescriptor_<4>;
template <int Dim_>
using PointIntensityDescriptorVectorCloud = std::vector<PointIntensityDescriptor_<Dim_>>;

struct Correspondence {
  int fixed_idx;
  int moving_idx;
  float response;
};
using CorrespondenceVector = std::vector<Correspondence>;
static_assert(sizeof(Correspondence) == sizeof(prs_corr), "Correspondence must match prs_corr");

// PARAM(PropertyT, name, ...) stand-in: value() / setValue() like srrg2_core properties
template <typename T>
class Property_ {
public:
  explicit Property_(const T& v, bool* changed_flag = nullptr) : _v(v), _flag(changed_flag) {}
  const T& value() const { return _v; }
  void setValue(const T& v) {
    _v = v;
    if (_flag) *_flag = true;
  }

private:
  T _v;
  bool* _flag;
};
using PropertyFloat       = Property_<float>;
using PropertyUnsignedInt = Property_<uint64_t>;
using PropertyBool        = Property_<bool>;
using PropertyInt         = Property_<int>;

// one prs_context shared by the plugin objects of a process (one device, one stream)
class Context {
public:
  explicit Context(int device = 0) {
    // the header this adapter was compiled against and the library it loaded must describe the same structs
    if (PRS_ABI_CHECK() != PRS_OK) {
      throw std::runtime_error("proslam_hip::Context|ERROR: libproslam_hip.so (version " + std::to_string(prs_version()) +
                               ") does not match proslam_hip.h (version " + std::to_string(PRS_ABI_VERSION) + "): rebuild the plugin");
    }
    const int rc = prs_context_create(device, &_ctx);
    if (rc != PRS_OK) throw std::runtime_error(std::string("proslam_hip::Context|ERROR: ") + prs_status_string(rc));
  }
  ~Context() { prs_context_destroy(_ctx); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  prs_context* get() const { return _ctx; }

private:
  prs_context* _ctx = nullptr;
};
using ContextPtr = std::shared_ptr<Context>;

inline void warn(const char* who, int flags) {
  // the reference prints yellow warnings to std::cerr and returns (bruteforce_impl.cpp:217-226,237-242)
  if (flags & PRS_WARN_EMPTY_INPUT) std::cerr << who << "|WARNING: no points in fixed or moving" << std::endl;
  if (flags & PRS_WARN_NO_MATCHES) std::cerr << who << "|WARNING: no correspondences found" << std::endl;
  if (flags & PRS_WARN_LOW_RATIO) std::cerr << who << "|low matching ratio" << std::endl;
  if (flags & PRS_WARN_RETRIED) std::cerr << who << "|WARNING: bad initial guess - triggering internal repeat with increased search radius" << std::endl;
  if (flags & PRS_WARN_TRACK_LOST) std::cerr << who << "|WARNING: complete track loss - fallback to identity motion guess" << std::endl;
}

// ---- stereo epipolar matcher ---------------------------------------------------------------------
template <typename FixedType_, typename MovingType_>
class CorrespondenceFinderDescriptorBasedEpipolarHIP {
public:
  using FixedType  = FixedType_;
  using MovingType = MovingType_;
  explicit CorrespondenceFinderDescriptorBasedEpipolarHIP(ContextPtr ctx) : _ctx(std::move(ctx)) {}
  // CF/correspondence_finder_descriptor_based_bruteforce.h:22-36
  PropertyFloat param_maximum_descriptor_distance{50.0f};
  PropertyFloat param_maximum_distance_ratio_to_second_best{0.9f};
  PropertyFloat param_minimum_matching_ratio{0.25f};
  // CF/correspondence_finder_descriptor_based_epipolar.h:22-32
  PropertyUnsignedInt param_maximum_disparity_pixels{100};
  PropertyUnsignedInt param_epipolar_line_thickness_pixels{0};
  // extent of the row table (image rows); the reference needs none because it compare-sorts
  PropertyUnsignedInt param_image_rows{4096};
  PropertyUnsignedInt param_image_cols{0};  // reserved

  void setFixed(const FixedType* fixed_) {
    _fixed              = fixed_;
    _fixed_changed_flag = true;
  }
  void setMoving(const MovingType* moving_) {
    _moving              = moving_;
    _moving_changed_flag = true;
  }
  void setCorrespondences(CorrespondenceVector* correspondences_) { _correspondences = correspondences_; }

  void compute() {
    // _preCompute (CF/..bruteforce_impl.cpp:203-216)
    if (!_fixed) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: fixed not set");
    if (!_moving) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: moving not set");
    if (!_correspondences) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: correspondences not set");
    // unchanged inputs keep the last computation state (CF/..epipolar_impl.cpp:50-52)
    if (!_fixed_changed_flag && !_moving_changed_flag) return;
    std::vector<prs_kp2> kl(_fixed->size()), kr(_moving->size());
    std::vector<uint8_t> dl(_fixed->size() * PRS_DESC_BYTES), dr(_moving->size() * PRS_DESC_BYTES);
    for (size_t i = 0; i < _fixed->size(); ++i) {
      kl[i] = prs_kp2{(*_fixed)[i].coordinates()[0], (*_fixed)[i].coordinates()[1]};
      std::memcpy(&dl[i * PRS_DESC_BYTES], (*_fixed)[i].descriptor(), PRS_DESC_BYTES);
    }
    for (size_t i = 0; i < _moving->size(); ++i) {
      kr[i] = prs_kp2{(*_moving)[i].coordinates()[0], (*_moving)[i].coordinates()[1]};
      std::memcpy(&dr[i * PRS_DESC_BYTES], (*_moving)[i].descriptor(), PRS_DESC_BYTES);
    }
    prs_stereo_params p;
    p.maximum_descriptor_distance           = param_maximum_descriptor_distance.value();
    p.maximum_distance_ratio_to_second_best = param_maximum_distance_ratio_to_second_best.value();
    p.minimum_matching_ratio                = param_minimum_matching_ratio.value();
    p.maximum_disparity_pixels              = (int32_t) param_maximum_disparity_pixels.value();
    p.epipolar_line_thickness_pixels        = (int32_t) param_epipolar_line_thickness_pixels.value();
    p.image_rows                            = (int32_t) param_image_rows.value();
    p.image_cols                            = (int32_t) param_image_cols.value();
    _correspondences->clear();
    _correspondences->resize(_fixed->size() + 1);
    int32_t n    = 0;
    const int rc = prs_stereo_match(_ctx->get(), &p, kl.data(), dl.data(), (int32_t) kl.size(), kr.data(), dr.data(), (int32_t) kr.size(),
                                    reinterpret_cast<prs_corr*>(_correspondences->data()), (int32_t) _correspondences->size(), &n);
    if (rc < 0) {
      _correspondences->clear();
      throw std::runtime_error(std::string("CorrespondenceFinderDescriptorBasedEpipolarHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    }
    _correspondences->resize((size_t) n);
    warn("CorrespondenceFinderDescriptorBasedEpipolarHIP::compute", rc);
    // _postCompute (CF/..bruteforce_impl.cpp:231-236)
    _fixed_changed_flag = _moving_changed_flag = false;
  }

protected:
  ContextPtr _ctx;
  const FixedType* _fixed                 = nullptr;
  const MovingType* _moving               = nullptr;
  CorrespondenceVector* _correspondences = nullptr;
  bool _fixed_changed_flag = false, _moving_changed_flag = false;
};
using CorrespondenceFinderDescriptorBasedEpipolarHIP3D3D =
  CorrespondenceFinderDescriptorBasedEpipolarHIP<PointIntensityDescriptorVectorCloud<3>, PointIntensityDescriptorVectorCloud<3>>;
using CorrespondenceFinderDescriptorBasedEpipolarHIP2D2D =
  CorrespondenceFinderDescriptorBasedEpipolarHIP<PointIntensityDescriptorVectorCloud<2>, PointIntensityDescriptorVectorCloud<2>>;

// ---- bijective brute-force matcher -----------------------------------------------------------------
// CorrespondenceFinderDescriptorBasedBruteforce (CF/correspondence_finder_descriptor_based_bruteforce.h:8-95)
template <typename FixedType_, typename MovingType_>
class CorrespondenceFinderDescriptorBasedBruteforceHIP {
public:
  using FixedType  = FixedType_;
  using MovingType = MovingType_;
  explicit CorrespondenceFinderDescriptorBasedBruteforceHIP(ContextPtr ctx) : _ctx(std::move(ctx)) {}
  // CF/correspondence_finder_descriptor_based_bruteforce.h:22-36
  PropertyFloat param_maximum_descriptor_distance{50.0f};
  PropertyFloat param_maximum_distance_ratio_to_second_best{0.9f};
  PropertyFloat param_minimum_matching_ratio{0.25f};
  void setFixed(const FixedType* fixed_) {
    _fixed              = fixed_;
    _fixed_changed_flag = true;
  }
  void setMoving(const MovingType* moving_) {
    _moving              = moving_;
    _moving_changed_flag = true;
  }
  void setCorrespondences(CorrespondenceVector* correspondences_) { _correspondences = correspondences_; }
  void compute() {
    // _preCompute (CF/..bruteforce_impl.cpp:203-226)
    if (!_fixed) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: fixed not set");
    if (!_moving) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: moving not set");
    if (!_correspondences) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: correspondences not set");
    if (!_fixed_changed_flag && !_moving_changed_flag) return;  // :12-14
    std::vector<uint8_t> df(_fixed->size() * PRS_DESC_BYTES), dm(_moving->size() * PRS_DESC_BYTES);
    for (size_t i = 0; i < _fixed->size(); ++i) std::memcpy(&df[i * PRS_DESC_BYTES], (*_fixed)[i].descriptor(), PRS_DESC_BYTES);
    for (size_t i = 0; i < _moving->size(); ++i) std::memcpy(&dm[i * PRS_DESC_BYTES], (*_moving)[i].descriptor(), PRS_DESC_BYTES);
    prs_bruteforce_params p;
    p.maximum_descriptor_distance           = param_maximum_descriptor_distance.value();
    p.maximum_distance_ratio_to_second_best = param_maximum_distance_ratio_to_second_best.value();
    p.minimum_matching_ratio                = param_minimum_matching_ratio.value();
    _correspondences->clear();
    _correspondences->resize(std::min(_fixed->size(), _moving->size()) + 1);
    int32_t n    = 0;
    const int rc = prs_bruteforce_match(_ctx->get(), &p, df.data(), (int32_t) _fixed->size(), dm.data(), (int32_t) _moving->size(),
                                        reinterpret_cast<prs_corr*>(_correspondences->data()), (int32_t) _correspondences->size(), &n);
    if (rc < 0) {
      _correspondences->clear();
      throw std::runtime_error(std::string("CorrespondenceFinderDescriptorBasedBruteforceHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    }
    _correspondences->resize((size_t) n);
    warn("CorrespondenceFinderDescriptorBasedBruteforceHIP::compute", rc);
    _fixed_changed_flag = _moving_changed_flag = false;  // _postCompute (:231-236)
  }

protected:
  ContextPtr _ctx;
  const FixedType* _fixed                 = nullptr;
  const MovingType* _moving               = nullptr;
  CorrespondenceVector* _correspondences = nullptr;
  bool _fixed_changed_flag = false, _moving_changed_flag = false;
};
using CorrespondenceFinderDescriptorBasedBruteforceHIP3D3D =
  CorrespondenceFinderDescriptorBasedBruteforceHIP<PointIntensityDescriptorVectorCloud<3>, PointIntensityDescriptorVectorCloud<3>>;

// ---- pinhole projector parameters (PointProjectorPinhole_ as seen through param_projector) -------
struct ProjectorPinholeHIP {
  float camera_matrix[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major K
  PropertyUnsignedInt param_canvas_cols{0};
  PropertyUnsignedInt param_canvas_rows{0};
  PropertyFloat param_range_min{0.3f};
  PropertyFloat param_range_max{20.0f};
  void setCameraMatrix(const float* K9) { std::memcpy(camera_matrix, K9, sizeof(camera_matrix)); }
  prs_projector raw() const {
    prs_projector p;
    p.fx = camera_matrix[0];
    p.fy = camera_matrix[4];
    p.cx = camera_matrix[2];
    p.cy = camera_matrix[5];
    p.canvas_cols = (int32_t) param_canvas_cols.value();
    p.canvas_rows = (int32_t) param_canvas_rows.value();
    p.range_min   = param_range_min.value();
    p.range_max   = param_range_max.value();
    return p;
  }
};
using ProjectorPinholeHIPPtr = std::shared_ptr<ProjectorPinholeHIP>;

// ---- projective finder ------------------------------------------------------------------------------
template <int SEARCH_, typename FixedType_, typename MovingType_>
class CorrespondenceFinderProjectiveHIP {
public:
  using FixedType  = FixedType_;
  using MovingType = MovingType_;
  explicit CorrespondenceFinderProjectiveHIP(ContextPtr ctx) : _ctx(std::move(ctx)), param_projector(new ProjectorPinholeHIP()) {}
  ~CorrespondenceFinderProjectiveHIP() {
    if (_h) prs_pcf_destroy(_h);
  }
  // CF/correspondence_finder_descriptor_based_bruteforce.h:22-36
  PropertyFloat param_maximum_descriptor_distance{50.0f};
  PropertyFloat param_maximum_distance_ratio_to_second_best{0.9f};
  PropertyFloat param_minimum_matching_ratio{0.25f};
  // CF/correspondence_finder_projective_base.h:30-74
  PropertyFloat param_minimum_descriptor_distance{25.0f, &_config_changed};
  PropertyFloat param_descriptor_distance_step_size_pixels{5.0f};
  PropertyUnsignedInt param_maximum_search_radius_pixels{100, &_config_changed};
  PropertyUnsignedInt param_minimum_search_radius_pixels{10};
  PropertyUnsignedInt param_search_radius_step_size_pixels{5};
  PropertyUnsignedInt param_minimum_number_of_iterations{10};
  PropertyFloat param_maximum_estimate_change_norm_for_convergence{1e-5f};
  PropertyUnsignedInt param_number_of_solver_iterations_per_projection{25};
  PropertyUnsignedInt param_minimum_number_of_points_per_cluster{10};  // KD-tree finder only (CF/..projective_kdtree.h:24-28)
  ProjectorPinholeHIPPtr param_projector;

  void setFixed(const FixedType* fixed_) {
    _fixed         = fixed_;
    _fixed_changed = true;
  }
  void setMoving(const MovingType* moving_) {
    _moving         = moving_;
    _moving_changed = true;
  }
  void setCorrespondences(CorrespondenceVector* correspondences_) { _correspondences = correspondences_; }
  void setLocalMapInSensor(const float* T16_row_major) { std::memcpy(_local_map_in_sensor, T16_row_major, sizeof(_local_map_in_sensor)); }
  void setSearchradiusPixels(const size_t& r) {  // CF/..projective_base.h:82-85
    ensureHandle();
    prs_pcf_set_search_radius(_h, r);
    _config_changed = false;
  }
  void setDescriptorDistance(const float& d) {  // CF/..projective_base.h:94-97
    ensureHandle();
    prs_pcf_set_descriptor_distance(_h, d);
    _config_changed = false;
  }
  size_t searchRadiusPixels() {
    prs_pcf_state s = state();
    return (size_t) s.search_radius_pixels;
  }
  prs_pcf_state state() {
    ensureHandle();
    prs_pcf_state s;
    prs_pcf_get_state(_h, &s);
    return s;
  }
  prs_pcf* handle() {
    ensureHandle();
    uploadIfChanged();
    return _h;
  }
  prs_pcf_params rawParams() const {
    prs_pcf_params p;
    p.maximum_descriptor_distance                  = param_maximum_descriptor_distance.value();
    p.maximum_distance_ratio_to_second_best        = param_maximum_distance_ratio_to_second_best.value();
    p.minimum_matching_ratio                       = param_minimum_matching_ratio.value();
    p.minimum_descriptor_distance                  = param_minimum_descriptor_distance.value();
    p.descriptor_distance_step_size_pixels         = param_descriptor_distance_step_size_pixels.value();
    p.maximum_search_radius_pixels                 = param_maximum_search_radius_pixels.value();
    p.minimum_search_radius_pixels                 = param_minimum_search_radius_pixels.value();
    p.search_radius_step_size_pixels               = param_search_radius_step_size_pixels.value();
    p.minimum_number_of_iterations                 = param_minimum_number_of_iterations.value();
    p.maximum_estimate_change_norm_for_convergence = param_maximum_estimate_change_norm_for_convergence.value();
    p.number_of_solver_iterations_per_projection   = param_number_of_solver_iterations_per_projection.value();
    p.search_type                                  = SEARCH_;
    p.projector                                    = param_projector->raw();
    p.minimum_number_of_points_per_cluster         = (int32_t) param_minimum_number_of_points_per_cluster.value();
    return p;
  }

  void compute() {
    if (!_fixed) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: fixed not set");
    if (!_moving) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: moving not set");
    if (!_correspondences) throw std::runtime_error("CorrespondenceFinderDescriptorBased::compute|ERROR: correspondences not set");
    if (!param_projector) throw std::runtime_error("CorrespondenceFinderProjective::compute|ERROR: projector not set");
    ensureHandle();
    uploadIfChanged();
    prs_pcf_set_local_map_in_sensor(_h, _local_map_in_sensor);
    std::vector<prs_corr> out(_fixed->size() + 1);
    int32_t n    = 0;
    const int rc = prs_pcf_compute(_h, out.data(), (int32_t) out.size(), &n);
    if (rc < 0) throw std::runtime_error(std::string("CorrespondenceFinderProjectiveHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    _correspondences->assign(reinterpret_cast<Correspondence*>(out.data()), reinterpret_cast<Correspondence*>(out.data()) + n);
    warn("CorrespondenceFinderProjectiveHIP::compute", rc & ~PRS_WARN_LOW_RATIO);
  }

protected:
  void ensureHandle() {
    if (!_h) {
      prs_pcf_params p = rawParams();
      const int rc     = prs_pcf_create(_ctx->get(), &p, &_h);
      if (rc != PRS_OK) throw std::runtime_error("CorrespondenceFinderProjectiveHIP|ERROR: cannot create finder handle");
      _config_changed = false;
    } else if (_config_changed) {
      prs_pcf_params p = rawParams();
      prs_pcf_set_params(_h, &p);
      _config_changed = false;
    }
  }
  void uploadIfChanged() {
    if (_fixed && _fixed_changed) {
      std::vector<float> c(_fixed->size() * FixedType::value_type::Dim);
      std::vector<uint8_t> d(_fixed->size() * PRS_DESC_BYTES);
      for (size_t i = 0; i < _fixed->size(); ++i) {
        std::memcpy(&c[i * FixedType::value_type::Dim], (*_fixed)[i].coordinates(), sizeof(float) * FixedType::value_type::Dim);
        std::memcpy(&d[i * PRS_DESC_BYTES], (*_fixed)[i].descriptor(), PRS_DESC_BYTES);
      }
      if (prs_pcf_set_fixed(_h, c.data(), FixedType::value_type::Dim, d.data(), (int32_t) _fixed->size()) < 0)
        throw std::runtime_error("CorrespondenceFinderProjectiveHIP|ERROR: set_fixed failed");
      _fixed_changed = false;
    }
    if (_moving && _moving_changed) {
      std::vector<float> c(_moving->size() * 3), s(_moving->size());
      std::vector<uint8_t> d(_moving->size() * PRS_DESC_BYTES);
      std::vector<uint32_t> nopt(_moving->size());
      for (size_t i = 0; i < _moving->size(); ++i) {
        std::memcpy(&c[i * 3], (*_moving)[i].coordinates(), sizeof(float) * 3);
        std::memcpy(&d[i * PRS_DESC_BYTES], (*_moving)[i].descriptor(), PRS_DESC_BYTES);
        nopt[i] = (*_moving)[i].number_of_optimizations;
      }
      // setupFactor's information scaling by landmark age (aligner_slice_processor_projective.cpp:46-52)
      prs_info_scale_from_nopt(nopt.data(), (int32_t) nopt.size(), s.data());
      if (prs_pcf_set_moving(_h, c.data(), s.data(), d.data(), (int32_t) _moving->size()) < 0)
        throw std::runtime_error("CorrespondenceFinderProjectiveHIP|ERROR: set_moving failed");
      _moving_changed = false;
    }
  }
  ContextPtr _ctx;
  prs_pcf* _h                            = nullptr;
  const FixedType* _fixed                = nullptr;
  const MovingType* _moving              = nullptr;
  CorrespondenceVector* _correspondences = nullptr;
  bool _fixed_changed = false, _moving_changed = false;
  bool _config_changed          = true;  // CF/..projective_base.h:134
  float _local_map_in_sensor[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
template <typename F, typename M>
using CorrespondenceFinderProjectiveKDTreeHIP = CorrespondenceFinderProjectiveHIP<PRS_SEARCH_KDTREE, F, M>;
template <typename F, typename M>
using CorrespondenceFinderProjectiveSquareHIP = CorrespondenceFinderProjectiveHIP<PRS_SEARCH_SQUARE, F, M>;
template <typename F, typename M>
using CorrespondenceFinderProjectiveCircleHIP = CorrespondenceFinderProjectiveHIP<PRS_SEARCH_CIRCLE, F, M>;
template <typename F, typename M>
using CorrespondenceFinderProjectiveRhombusHIP = CorrespondenceFinderProjectiveHIP<PRS_SEARCH_RHOMBUS, F, M>;
using CorrespondenceFinderProjectiveCircleHIP4D3D =
  CorrespondenceFinderProjectiveCircleHIP<PointIntensityDescriptorVectorCloud<4>, PointIntensityDescriptorVectorCloud<3>>;
using CorrespondenceFinderProjectiveCircleHIP2D3D =
  CorrespondenceFinderProjectiveCircleHIP<PointIntensityDescriptorVectorCloud<2>, PointIntensityDescriptorVectorCloud<3>>;

// ---- triangulator -----------------------------------------------------------------------------------
class TriangulatorRigidStereoHIP {
public:
  using MeasurementType = PointIntensityDescriptorVectorCloud<4>;
  using DestType        = PointIntensityDescriptorVectorCloud<3>;
  explicit TriangulatorRigidStereoHIP(ContextPtr ctx) : _ctx(std::move(ctx)) {}
  PropertyFloat param_minimum_disparity_pixels{1.0f};               // mapping/triangulator_rigid_stereo.h:34-38
  PropertyFloat param_infinity_depth_meters{1.8446743e19f};         // :39-43 sqrt(FLT_MAX)
  ProjectorPinholeHIPPtr param_projector;
  void setDest(DestType* dest_) { _dest = dest_; }
  void setMoving(const MeasurementType* matches_) { _stereo_intensity_matches = matches_; }
  // platform->getTransform(camera_right in camera_left).translation(), mapping/triangulator_rigid_stereo.cpp:103-106
  void setBaselineRightInLeftMeters(float tx, float ty, float tz) {
    _t[0] = tx;
    _t[1] = ty;
    _t[2] = tz;
    _baseline_set = false;
  }
  void initializeBaseline() {
    if (_baseline_set || !param_projector) return;
    const float* K = param_projector->camera_matrix;
    for (int r = 0; r < 3; ++r) _baseline_right_in_left[r] = K[3 * r] * _t[0] + K[3 * r + 1] * _t[1] + K[3 * r + 2] * _t[2];
    _baseline_set = true;
  }
  const float* baselineRigthInLeft() const { return _baseline_right_in_left; }
  const std::vector<size_t>& indicesInvalidated() const { return _indices_invalidated; }
  void compute() {
    if (!_stereo_intensity_matches) {  // mapping/triangulator_rigid_stereo.cpp:9-16: log + return
      std::cerr << "TriangulatorRigidStereo::compute|ERROR: input not set" << std::endl;
      return;
    }
    if (!_dest) {
      std::cerr << "TriangulatorRigidStereo::compute|ERROR: result buffer not set" << std::endl;
      return;
    }
    initializeBaseline();
    const size_t n = _stereo_intensity_matches->size();
    std::vector<float> uvuv(n * 4), xyz(n * 3);
    std::vector<uint8_t> valid(n);
    for (size_t i = 0; i < n; ++i) std::memcpy(&uvuv[i * 4], (*_stereo_intensity_matches)[i].coordinates(), sizeof(float) * 4);
    prs_triangulator_params p;
    const float* K = param_projector->camera_matrix;
    p.fx = K[0];
    p.fy = K[4];
    p.cx = K[2];
    p.cy = K[5];
    p.b_x = _baseline_right_in_left[0];
    p.minimum_disparity_pixels = param_minimum_disparity_pixels.value();
    p.infinity_depth_meters    = param_infinity_depth_meters.value();
    const int rc = prs_triangulate(_ctx->get(), &p, uvuv.data(), (int32_t) n, xyz.data(), valid.data());
    if (rc < 0) throw std::runtime_error(std::string("TriangulatorRigidStereoHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    _dest->clear();
    _dest->resize(n);
    _indices_invalidated.clear();
    for (size_t i = 0; i < n; ++i) {
      PointIntensityDescriptor3f& d = (*_dest)[i];
      d.valid                      = valid[i] != 0;
      std::memcpy(d.coords, &xyz[i * 3], sizeof(float) * 3);
      if (d.valid) {  // descriptor + intensity of the left measurement are kept (:49-50)
        std::memcpy(d.descriptor_row, (*_stereo_intensity_matches)[i].descriptor(), PRS_DESC_BYTES);
        d.intensity_value = (*_stereo_intensity_matches)[i].intensity_value;
      } else {
        _indices_invalidated.push_back(i);
      }
    }
  }

protected:
  ContextPtr _ctx;
  const MeasurementType* _stereo_intensity_matches = nullptr;
  DestType* _dest                                  = nullptr;
  float _t[3]                       = {0, 0, 0};
  float _baseline_right_in_left[3] = {0, 0, 0};
  bool _baseline_set               = false;
  std::vector<size_t> _indices_invalidated;
};

// ---- aligner: MultiAligner3DQR with one projective slice -------------------------------------------
template <typename FinderType_>
class AlignerProjectiveHIP {
public:
  enum Status { Fail = 0, Success = 1 };
  using FixedType  = typename FinderType_::FixedType;
  using MovingType = typename FinderType_::MovingType;
  explicit AlignerProjectiveHIP(ContextPtr ctx) : _ctx(ctx), param_finder(new FinderType_(ctx)) {}
  std::shared_ptr<FinderType_> param_finder;      // slice->param_finder
  PropertyUnsignedInt param_max_iterations{10};   // MultiAligner3DQR
  PropertyUnsignedInt param_min_num_inliers{6};
  PropertyUnsignedInt param_min_num_correspondences{0};
  PropertyFloat param_damping{0.0f};              // IterationAlgorithmGN
  PropertyFloat param_chi_threshold{100.0f * 100.0f};  // RobustifierSaturated (aligner_slice_processor_projective.cpp:18)
  float param_diagonal_info_matrix[3]     = {1, 1, 1};
  bool param_enable_inverse_depth_weighting = false;
  float baseline_left_in_right_pixels[3]    = {0, 0, 0};  // K * t_left_in_right (aligner_slice_processor_projective.cpp:98-104)
  // MultiAligner3DQR flags the RGB-D configurations switch on (configurations/icl.conf:50-64, tum.conf:90-104)
  PropertyBool param_enable_inlier_only_runs{false};
  PropertyBool param_keep_only_inlier_correspondences{false};
  // readings of the un-vendored srrg2_solver arithmetic (include/proslam_hip.h, prs_aligner_params): 0 = the family
  // this library ships; a maintainer whose srrg2_solver does otherwise selects the other reading here (or in the .conf)
  PropertyInt param_robustifier_kernel_weight_form{PRS_KERNEL_WEIGHT_INV_CHI};     // RobustifierSaturated: Omega / chi | Omega * tau / chi
  PropertyFloat param_step_norm_exit{0.f};  // OPT-IN, not a parameter of the reference: leave the loop once the finder has latched and |dx| is below this (0 = off)
  PropertyInt param_damping_form{PRS_DAMPING_DIAG};                                // IterationAlgorithmGN: H + lambda diag(H) | H + lambda I
  PropertyInt param_translation_weight_form{PRS_TRANSLATION_WEIGHT_OFFSET};        // min(0.01 + d / mean, 1) | clamp(d / mean, 0.01, 1)
  // AlignerSliceMotionModel3D (kitti.conf:747-772): prior on movingInFixed around setMotionPriorMean() (identity by default)
  PropertyBool param_enable_motion_model_slice{false};
  float param_motion_model_information[6] = {1, 1, 1, 1, 1, 1};

  // the ...WithSensor slice processors read sensor_in_robot from the Platform (setPlatform,
  // aligner_slice_processor_projective.h:80-83): row-major 4x4; the estimate is then the ROBOT's movingInFixed
  void setSensorInRobot(const float* T16_row_major) {
    std::memcpy(_sensor_in_robot, T16_row_major, sizeof(_sensor_in_robot));
    _with_sensor = true;
  }
  void setMotionPriorMean(const float* T16_row_major) { prs_pcf_set_motion_prior_mean(param_finder->handle(), T16_row_major); }

  void setFixed(const FixedType* fixed_) { param_finder->setFixed(fixed_); _n_fixed = fixed_ ? fixed_->size() : 0; }
  void setMoving(const MovingType* moving_) { param_finder->setMoving(moving_); }
  void setMovingInFixed(const float* T16_row_major) { std::memcpy(_moving_in_fixed, T16_row_major, sizeof(_moving_in_fixed)); }
  const float* movingInFixed() const { return _moving_in_fixed; }
  Status status() const { return _status; }
  const CorrespondenceVector& correspondences() const { return _correspondences; }
  const prs_align_result& result() const { return _result; }

  void compute() {
    prs_aligner_params a;
    std::memset(&a, 0, sizeof(a));
    const prs_projector pr = param_finder->param_projector->raw();
    a.factor_type = FixedType::value_type::Dim;
    a.fx = pr.fx; a.fy = pr.fy; a.cx = pr.cx; a.cy = pr.cy;
    a.image_cols = (float) pr.canvas_cols;
    a.image_rows = (float) pr.canvas_rows;
    for (int i = 0; i < 3; ++i) {
      a.baseline_left_in_right_px[i] = baseline_left_in_right_pixels[i];
      a.diagonal_info[i]             = param_diagonal_info_matrix[i];
    }
    a.chi_threshold                  = param_chi_threshold.value();
    a.enable_inverse_depth_weighting = param_enable_inverse_depth_weighting ? 1 : 0;
    a.mean_disparity                 = -1.0f;  // bindFixed: computed over all fixed points on the device
    a.damping                        = param_damping.value();
    a.max_iterations                 = (int32_t) param_max_iterations.value();
    a.min_num_inliers                = (int32_t) param_min_num_inliers.value();
    a.min_num_correspondences        = (int32_t) param_min_num_correspondences.value();
    a.stop_at_fixed_point            = 1;
    a.enable_inlier_only_runs          = param_enable_inlier_only_runs.value() ? 1 : 0;
    a.keep_only_inlier_correspondences = param_keep_only_inlier_correspondences.value() ? 1 : 0;
    a.with_sensor                      = _with_sensor ? 1 : 0;
    std::memcpy(a.sensor_in_robot, _sensor_in_robot, sizeof(_sensor_in_robot));
    a.kernel_weight_form      = (int32_t) param_robustifier_kernel_weight_form.value();
    a.damping_form            = (int32_t) param_damping_form.value();
    a.step_norm_exit          = param_step_norm_exit.value();
    a.translation_weight_form = (int32_t) param_translation_weight_form.value();
    a.enable_motion_prior = param_enable_motion_model_slice.value() ? 1 : 0;
    for (int i = 0; i < 6; ++i) a.motion_prior_info[i] = param_motion_model_information[i];
    std::vector<prs_corr> out(_n_fixed + 1);
    int32_t n = 0;
    float X[16];
    const int rc = prs_pcf_align(param_finder->handle(), &a, _moving_in_fixed, nullptr, X, out.data(), (int32_t) out.size(), &n, &_result);
    if (rc < 0) throw std::runtime_error(std::string("AlignerProjectiveHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    std::memcpy(_moving_in_fixed, X, sizeof(X));
    _correspondences.assign(reinterpret_cast<Correspondence*>(out.data()), reinterpret_cast<Correspondence*>(out.data()) + n);
    _status = _result.status ? Success : Fail;
  }

protected:
  ContextPtr _ctx;
  size_t _n_fixed = 0;
  float _moving_in_fixed[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float _sensor_in_robot[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  bool _with_sensor          = false;
  Status _status = Fail;
  CorrespondenceVector _correspondences;
  prs_align_result _result;
};

// ---- scene clipper ----------------------------------------------------------------------------------
// SceneClipperProjective3D (mapping/scene_clipper_projective_3d.h:10-43, .cpp:9-67)
class SceneClipperProjective3DHIP {
public:
  using SceneType = PointIntensityDescriptorVectorCloud<3>;
  enum Status { Error = 0, Ready = 1, Successful = 2 };
  explicit SceneClipperProjective3DHIP(ContextPtr ctx) : _ctx(std::move(ctx)), param_projector(new ProjectorPinholeHIP()) {
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(_robot_in_local_map, I, sizeof(I));
    std::memcpy(_sensor_in_robot, I, sizeof(I));
  }
  void setFullScene(const SceneType* full_scene_) { _full_scene = full_scene_; }
  void setClippedSceneInRobot(SceneType* clipped_) { _clipped_scene_in_robot = clipped_; }
  void setRobotInLocalMap(const float* T16) { std::memcpy(_robot_in_local_map, T16, sizeof(_robot_in_local_map)); }
  void setSensorInRobot(const float* T16) { std::memcpy(_sensor_in_robot, T16, sizeof(_sensor_in_robot)); }
  const std::vector<int> globalIndices() const { return _global_indices; }
  Status status() const { return _status; }
  void compute() {
    _status = Error;
    if (!param_projector) throw std::runtime_error("SceneClipperProjective3D::compute|ERROR: missing projector");
    if (!_clipped_scene_in_robot) throw std::runtime_error("SceneClipperProjective3D::compute|ERROR: missing clipped scene");
    if (!_full_scene) throw std::runtime_error("SceneClipperProjective3D::compute|ERROR: missing global scene");
    if (_full_scene->empty()) {  // scene_clipper_projective_3d.cpp:21-28: nothing is cleared
      std::cerr << "SceneClipperProjective3D::compute|WARNING: global scene is empty, no clipping will be performed" << std::endl;
      _status = Ready;
      return;
    }
    const size_t n = _full_scene->size();
    std::vector<float> xyzw(4 * n), out(4 * n);
    std::vector<uint8_t> desc(PRS_DESC_BYTES * n), odesc(PRS_DESC_BYTES * n);
    std::vector<int32_t> idx(n);
    for (size_t i = 0; i < n; ++i) {  // AoS -> SoA gather; w carries the index so the other fields can be copied back
      std::memcpy(&xyzw[4 * i], (*_full_scene)[i].coords, sizeof(float) * 3);
      xyzw[4 * i + 3] = 0.f;
      std::memcpy(&desc[PRS_DESC_BYTES * i], (*_full_scene)[i].descriptor_row, PRS_DESC_BYTES);
    }
    const prs_projector p = param_projector->raw();
    int32_t m             = 0;
    const int rc = prs_scene_clip(_ctx->get(), &p, _robot_in_local_map, _sensor_in_robot, xyzw.data(), desc.data(), (int32_t) n,
                                  out.data(), odesc.data(), idx.data(), (int32_t) n, &m);
    if (rc < 0) throw std::runtime_error(std::string("SceneClipperProjective3DHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    _clipped_scene_in_robot->clear();
    _global_indices.clear();
    _clipped_scene_in_robot->reserve((size_t) m);
    for (int32_t k = 0; k < m; ++k) {
      PointIntensityDescriptor3f q = (*_full_scene)[(size_t) idx[k]];  // intensity, statistics, descriptor travel with the point
      std::memcpy(q.coords, &out[4 * (size_t) k], sizeof(float) * 3);
      _clipped_scene_in_robot->push_back(q);
      _global_indices.push_back(idx[k]);
    }
    if (rc & PRS_WARN_NO_PROJECTION) std::cerr << "SceneClipperProjective3D::compute|WARNING: clipped empty scene" << std::endl;
    _status = Successful;
  }
  ProjectorPinholeHIPPtr param_projector;

protected:
  ContextPtr _ctx;
  const SceneType* _full_scene       = nullptr;
  SceneType* _clipped_scene_in_robot = nullptr;
  float _robot_in_local_map[16];
  float _sensor_in_robot[16];
  std::vector<int> _global_indices;
  Status _status = Error;
};

// IntensityFeatureExtractorBinned_ (sensor_processing/feature_extractors/intensity_feature_extractor_binned.{h,cpp},
// base class intensity_feature_extractor_base.h): same PARAM names, setFeatures() / compute(image); the image is a plain
// 8-bit buffer here where the reference takes a cv::Mat.  Features are (u, v) points with intensity and a 256-bit descriptor.
class IntensityFeatureExtractorBinnedHIP {
public:
  using PointCloudType = PointIntensityDescriptorVectorCloud<2>;
  explicit IntensityFeatureExtractorBinnedHIP(ContextPtr ctx) : _ctx(std::move(ctx)) {}
  PropertyFloat param_detector_threshold{10.f};                  // intensity_feature_extractor_base.h:36-40
  PropertyBool param_enable_non_maximum_suppression{true};       // :48-52
  Property_<int> param_target_number_of_keypoints{500};          // :54-58
  PropertyUnsignedInt param_number_of_detectors_vertical{1};     // intensity_feature_extractor_binned.h:17-22
  PropertyUnsignedInt param_number_of_detectors_horizontal{1};   // :23-28
  // which members of a response class survive the per-region cut: the reference's std::sort order (GCC) by default
  Property_<int> param_selection_order{PRS_SELECT_LIBSTDCXX};
  void setFeatures(PointCloudType* features_) { _features = features_; }
  void compute(const uint8_t* image, int rows, int cols, int pitch) {
    if (!image || rows <= 0 || cols <= 0) throw std::runtime_error("IntensityFeatureExtractor::compute|ERROR: image not set");
    if (!_features) throw std::runtime_error("IntensityFeatureExtractor::compute|ERROR: target feature buffer not set");
    prs_extractor_params p;
    p.detector_threshold             = (int32_t) param_detector_threshold.value();
    p.enable_non_maximum_suppression = param_enable_non_maximum_suppression.value() ? 1 : 0;
    p.target_number_of_keypoints     = param_target_number_of_keypoints.value();
    p.number_of_detectors_vertical   = (int32_t) param_number_of_detectors_vertical.value();
    p.number_of_detectors_horizontal = (int32_t) param_number_of_detectors_horizontal.value();
    p.selection_order                = param_selection_order.value();
    p.max_raw_detections             = 32768;
    const int32_t capacity = 8192;
    std::vector<float> kp(2 * (size_t) capacity), inten((size_t) capacity);
    std::vector<uint8_t> desc(PRS_DESC_BYTES * (size_t) capacity);
    int32_t n    = 0;
    const int rc = prs_extract_features(_ctx->get(), &p, image, rows, cols, pitch, kp.data(), inten.data(), desc.data(), capacity, &n);
    if (rc < 0) throw std::runtime_error(std::string("IntensityFeatureExtractorBinnedHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    _features->clear();
    _features->reserve((size_t) n);
    for (int32_t i = 0; i < n; ++i) {
      PointIntensityDescriptor_<2> q;
      q.coords[0] = kp[2 * (size_t) i];
      q.coords[1] = kp[2 * (size_t) i + 1];
      q.intensity_value = inten[(size_t) i];
      std::memcpy(q.descriptor_row, &desc[PRS_DESC_BYTES * (size_t) i], PRS_DESC_BYTES);
      _features->push_back(q);
    }
  }

protected:
  ContextPtr _ctx;
  PointCloudType* _features = nullptr;
};

// MergerRigidStereoTriangulation with LandmarkEstimatorWeightedMean4D3D (mapping/mergers/merger_rigid_stereo_triangulation.h,
// merger_projective.h, landmarks/landmark_estimator_weighted_mean.h): same setters and PARAM names; the scene is mirrored into a
// device-resident map (prs_map) on setScene and read back after compute(), so a caller sees its scene cloud updated in place like
// with the reference object.  Other variants / estimators only differ in the prs_merger_params they fill.
class MergerRigidStereoTriangulationHIP {
public:
  using SceneType       = PointIntensityDescriptorVectorCloud<3>;
  using MeasurementType = PointIntensityDescriptorVectorCloud<4>;
  explicit MergerRigidStereoTriangulationHIP(ContextPtr ctx) : param_projector(new ProjectorPinholeHIP()), _ctx(std::move(ctx)) {
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(_measurement_in_scene, I, sizeof(I));
    std::memcpy(_measurement_in_world, I, sizeof(I));
  }
  ~MergerRigidStereoTriangulationHIP() {
    if (_map) prs_map_destroy(_map);
  }
  PropertyFloat param_maximum_distance_appearance{50.f};       // merger_projective.h:42-46
  PropertyUnsignedInt param_number_of_row_bins{10};            // :47-51
  PropertyUnsignedInt param_number_of_col_bins{30};            // :52-56
  PropertyFloat param_target_merge_ratio{0.5f};                // :57-61
  PropertyBool param_enable_binning{true};
  PropertyUnsignedInt param_target_number_of_merges{100};
  PropertyFloat param_maximum_distance_geometry_meters_squared{1.f};  // landmark_estimator_base.hpp:20-25 (of the landmark estimator)
  PropertyFloat param_minimum_disparity_pixels{1.f};           // triangulator_rigid_stereo.h:34-38 (of the triangulator)
  ProjectorPinholeHIPPtr param_projector;
  void setBaselineRightInLeftPixels(float bx) { _baseline_px = bx; }
  void setScene(SceneType* scene_) {
    _scene         = scene_;
    _scene_changed = true;
  }
  void setMeasurement(const MeasurementType* measurement_) { _measurement = measurement_; }
  void setCorrespondences(const CorrespondenceVector* correspondences_) { _correspondences = correspondences_; }
  void setMeasurementInScene(const float* T16) { std::memcpy(_measurement_in_scene, T16, sizeof(_measurement_in_scene)); }
  void setMeasurementInWorld(const float* T16) { std::memcpy(_measurement_in_world, T16, sizeof(_measurement_in_world)); }
  // pre-sizes the device-resident map (grown in place: landmark states, covariances and counters are kept)
  void reserve(size_t capacity) {
    if (_map && (int32_t) capacity > _capacity) {
      if (prs_map_reserve(_map, (int32_t) capacity) != PRS_OK) {
        throw std::runtime_error(std::string("MergerRigidStereoTriangulationHIP::reserve|ERROR: ") + prs_last_error(_ctx->get()));
      }
      _capacity = (int32_t) capacity;
    }
  }
  size_t numberOfMergedPoints() const { return _n_merged; }
  size_t numberOfAddedPoints() const { return _n_added; }
  void compute() {
    // merger_projective_impl.cpp:12-27
    if (!_scene) throw std::runtime_error("MergerProjective::compute|ERROR: scene not set");
    if (!_measurement) throw std::runtime_error("MergerProjective::compute|ERROR: measurement not set");
    if (!_correspondences) throw std::runtime_error("MergerProjective::compute|ERROR: correspondences not set");
    if (!param_projector) throw std::runtime_error("MergerProjective::compute|ERROR: projector not set");
    // room for this frame's additions.  After the first upload the DEVICE copy of the scene is the master (landmark states in
    // world coordinates, covariances, counters live there), so a map that has to grow is grown in place (prs_map_reserve keeps
    // every array); only setScene() makes the host scene authoritative again.
    int32_t on_device = 0;
    if (_map && !_scene_changed) prs_map_size(_map, &on_device, nullptr);
    const size_t scene_size = _scene_changed ? _scene->size() : (size_t) on_device;
    const int32_t capacity  = (int32_t) (scene_size + _measurement->size() + 1024);
    if (!_map) {
      _capacity    = 2 * capacity;
      const int rc = prs_map_create(_ctx->get(), _capacity, 0, 4096, 8192, &_map);
      if (rc != PRS_OK) throw std::runtime_error(std::string("MergerRigidStereoTriangulationHIP|ERROR: ") + prs_last_error(_ctx->get()));
      _scene_changed = true;
    } else if (capacity > _capacity) {
      _capacity = 2 * capacity;
      if (prs_map_reserve(_map, _capacity) != PRS_OK) {
        throw std::runtime_error(std::string("MergerRigidStereoTriangulationHIP|ERROR: ") + prs_last_error(_ctx->get()));
      }
    }
    if (_scene_changed) {  // upload the scene once; afterwards the device copy is the master
      const size_t n = _scene->size();
      std::vector<float> xyz(3 * n);
      std::vector<uint8_t> desc(PRS_DESC_BYTES * n);
      std::vector<uint32_t> nopt(n);
      for (size_t i = 0; i < n; ++i) {
        std::memcpy(&xyz[3 * i], (*_scene)[i].coords, sizeof(float) * 3);
        std::memcpy(&desc[PRS_DESC_BYTES * i], (*_scene)[i].descriptor_row, PRS_DESC_BYTES);
        nopt[i] = (*_scene)[i].number_of_optimizations;
      }
      prs_map_clear(_map);
      if (prs_map_set_scene(_map, xyz.data(), nullptr, nullptr, desc.data(), nopt.data(), nullptr, (int32_t) n) != PRS_OK) {
        throw std::runtime_error(std::string("MergerRigidStereoTriangulationHIP|ERROR: ") + prs_last_error(_ctx->get()));
      }
      _scene_changed = false;
    }
    prs_merger_params p;
    std::memset(&p, 0, sizeof(p));
    const float* K = param_projector->camera_matrix;
    p.variant                     = PRS_MERGER_STEREO_TRIANGULATION;
    p.enable_binning              = param_enable_binning.value() ? 1 : 0;
    p.number_of_row_bins          = (uint32_t) param_number_of_row_bins.value();
    p.number_of_col_bins          = (uint32_t) param_number_of_col_bins.value();
    p.canvas_rows                 = (int32_t) param_projector->param_canvas_rows.value();
    p.canvas_cols                 = (int32_t) param_projector->param_canvas_cols.value();
    p.maximum_distance_appearance = param_maximum_distance_appearance.value();
    p.target_number_of_merges     = (uint32_t) param_target_number_of_merges.value();
    p.target_merge_ratio          = param_target_merge_ratio.value();
    p.triangulator.fx = K[0];
    p.triangulator.fy = K[4];
    p.triangulator.cx = K[2];
    p.triangulator.cy = K[5];
    p.triangulator.b_x                      = _baseline_px;
    p.triangulator.minimum_disparity_pixels = param_minimum_disparity_pixels.value();
    p.triangulator.infinity_depth_meters    = 1.8446743e19f;
    p.fx = K[0];
    p.fy = K[4];
    p.cx = K[2];
    p.cy = K[5];
    p.estimator.type            = PRS_EST_WEIGHTED_MEAN;
    p.estimator.measurement_dim = 4;
    p.estimator.maximum_distance_geometry_meters_squared = param_maximum_distance_geometry_meters_squared.value();
    const size_t nm = _measurement->size();
    std::vector<float> z(4 * nm);
    std::vector<uint8_t> zd(PRS_DESC_BYTES * nm);
    for (size_t i = 0; i < nm; ++i) {
      std::memcpy(&z[4 * i], (*_measurement)[i].coords, sizeof(float) * 4);
      std::memcpy(&zd[PRS_DESC_BYTES * i], (*_measurement)[i].descriptor_row, PRS_DESC_BYTES);
    }
    static_assert(sizeof(Correspondence) == sizeof(prs_corr), "layout");
    prs_merge_result res;
    const int rc = prs_map_merge(_map, &p, _measurement_in_world, _measurement_in_scene, z.data(), zd.data(), (int32_t) nm,
                                 reinterpret_cast<const prs_corr*>(_correspondences->data()), (int32_t) _correspondences->size(), nullptr, 0, &res);
    if (rc < 0) throw std::runtime_error(std::string("MergerRigidStereoTriangulationHIP::compute|ERROR: ") + prs_last_error(_ctx->get()));
    if (rc & PRS_WARN_NO_MATCHES) std::cerr << "MergerProjective::compute|WARNING: all merge attempts failed" << std::endl;
    _n_merged = (size_t) res.n_merged;
    _n_added  = (size_t) res.n_added;
    // mirror the scene back: element order intact, new points appended (merger_projective_impl.cpp:230-308)
    int32_t n = 0;
    prs_map_size(_map, &n, nullptr);
    std::vector<float> xyz(3 * (size_t) _capacity);
    std::vector<uint8_t> desc(PRS_DESC_BYTES * (size_t) _capacity);
    std::vector<uint32_t> nopt((size_t) _capacity);
    if (prs_map_get_scene(_map, _capacity, xyz.data(), nullptr, desc.data(), nopt.data(), nullptr, &n) != PRS_OK) {
      throw std::runtime_error(std::string("MergerRigidStereoTriangulationHIP|ERROR: ") + prs_last_error(_ctx->get()));
    }
    _scene->resize((size_t) n);
    for (int32_t i = 0; i < n; ++i) {
      std::memcpy((*_scene)[(size_t) i].coords, &xyz[3 * (size_t) i], sizeof(float) * 3);
      std::memcpy((*_scene)[(size_t) i].descriptor_row, &desc[PRS_DESC_BYTES * (size_t) i], PRS_DESC_BYTES);
      (*_scene)[(size_t) i].number_of_optimizations = nopt[(size_t) i];
    }
  }

protected:
  ContextPtr _ctx;
  prs_map* _map      = nullptr;
  int32_t _capacity  = 0;
  SceneType* _scene  = nullptr;
  bool _scene_changed = true;
  const MeasurementType* _measurement          = nullptr;
  const CorrespondenceVector* _correspondences = nullptr;
  float _measurement_in_scene[16];
  float _measurement_in_world[16];
  float _baseline_px = 0.f;
  size_t _n_merged = 0, _n_added = 0;
};

}  // namespace proslam_hip

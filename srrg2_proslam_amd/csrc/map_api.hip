// map_api.hip -- host-pointer entry points of the mergers: a stateful handle that owns ONE device-resident local map
// (the structure of arrays prs_merge_batch describes) the way the reference's merger owns its scene pointer
// (mapping/mergers/merger_projective.h: setScene / setMeasurement / setCorrespondences / setMeasurementInScene / compute),
// on top of the batched device kernel (batch = 1).  An adapter keeps one prs_map per local map: the scene is uploaded once,
// every compute() uploads the frame (measurements, correspondences, two transforms) and downloads only what it asks for.
#include <string.h>

#include <vector>

#include "prs_host.h"

struct prs_map {
  prs_context* ctx = nullptr;
  int capacity = 0, max_measurements = 0, max_frames = 0, max_measured = 0;
  int frames = 0;  // frames merged since the last clear = pose-table slot of the next one
  int n_points = 0;
  float* d_coords = nullptr;
  uint8_t* d_desc = nullptr;
  float* d_state = nullptr;
  float* d_cov = nullptr;
  uint32_t* d_n_opt = nullptr;
  uint8_t* d_inlier = nullptr;
  uint32_t* d_n_meas = nullptr;
  prs_camera_measurement* d_meas = nullptr;
  prs_frame_pose* d_poses = nullptr;
  float* d_measurement = nullptr;
  uint8_t* d_mdesc = nullptr;
  prs_corr* d_corr = nullptr;
  int32_t* d_index_map = nullptr;
  unsigned char* d_small = nullptr;  // n_points, n_measured, n_corr, frame, result, two transforms
};

namespace {

constexpr size_t kSmall = 512;
struct Small {
  int32_t* n_points;
  int32_t* n_measured;
  int32_t* n_corr;
  int32_t* frame;
  prs_merge_result* result;
  float* in_world;
  float* in_scene;
};
Small small_of(unsigned char* d) {
  Small s;
  s.n_points   = reinterpret_cast<int32_t*>(d + 0);
  s.n_measured = reinterpret_cast<int32_t*>(d + 4);
  s.n_corr     = reinterpret_cast<int32_t*>(d + 8);
  s.frame      = reinterpret_cast<int32_t*>(d + 12);
  s.result     = reinterpret_cast<prs_merge_result*>(d + 16);
  s.in_world   = reinterpret_cast<float*>(d + 64);
  s.in_scene   = reinterpret_cast<float*>(d + 128);
  return s;
}

int fail(prs_map* h, int status, const char* what) {
  return prs::ctx_fail(h ? h->ctx : nullptr, status, what);
}

// (an error return synchronises first: asynchronous copies from the caller's or local staging buffers may be in flight)
#define MAP_TRY(x)                                           \
  do {                                                       \
    hipError_t e_ = (x);                                     \
    if (e_ != hipSuccess) {                                  \
      (void) hipStreamSynchronize(h->ctx->stream);           \
      return prs::ctx_fail_hip(h->ctx, e_, "prs_map: " #x);  \
    }                                                        \
  } while (0)

template <typename T>
hipError_t alloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T));
}

void release(prs_map* h) {
  void* ptrs[] = {h->d_coords, h->d_desc, h->d_state, h->d_cov, h->d_n_opt, h->d_inlier, h->d_n_meas, h->d_meas,
                  h->d_poses, h->d_measurement, h->d_mdesc, h->d_corr, h->d_index_map, h->d_small};
  for (void* p : ptrs) {
    if (p) {
      (void) hipFree(p);
    }
  }
}

}  // namespace

extern "C" {

int prs_map_create(prs_context* ctx, int32_t capacity, int32_t max_measurements, int32_t max_frames, int32_t max_measured, prs_map** out) {
  if (!ctx || !out) {
    return PRS_ERR_NULL;
  }
  if (capacity <= 0 || max_measurements < 0 || max_frames <= 0 || max_measured <= 0) {
    return prs::ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_map_create: capacity, max_frames and max_measured must be positive");
  }
  (void) hipSetDevice(ctx->device);
  prs_map* h = new prs_map();
  h->ctx = ctx;
  h->capacity = capacity;
  h->max_measurements = max_measurements;
  h->max_frames = max_frames;
  h->max_measured = max_measured;
  const size_t cap = (size_t) capacity;
  hipError_t e = alloc(&h->d_coords, cap * 4);
  e = e == hipSuccess ? alloc(&h->d_desc, cap * PRS_DESC_BYTES) : e;
  e = e == hipSuccess ? alloc(&h->d_state, cap * 4) : e;
  e = e == hipSuccess ? alloc(&h->d_cov, cap * 9) : e;
  e = e == hipSuccess ? alloc(&h->d_n_opt, cap) : e;
  e = e == hipSuccess ? alloc(&h->d_inlier, cap) : e;
  e = e == hipSuccess ? alloc(&h->d_n_meas, cap) : e;
  e = e == hipSuccess ? alloc(&h->d_meas, cap * (size_t) (max_measurements > 0 ? max_measurements : 1)) : e;
  e = e == hipSuccess ? alloc(&h->d_poses, (size_t) max_frames) : e;
  e = e == hipSuccess ? alloc(&h->d_measurement, (size_t) max_measured * 4) : e;
  e = e == hipSuccess ? alloc(&h->d_mdesc, (size_t) max_measured * PRS_DESC_BYTES) : e;
  e = e == hipSuccess ? alloc(&h->d_corr, (size_t) max_measured) : e;
  e = e == hipSuccess ? alloc(&h->d_index_map, cap) : e;
  e = e == hipSuccess ? alloc(&h->d_small, kSmall) : e;
  if (e == hipSuccess) {
    e = hipMemsetAsync(h->d_small, 0, kSmall, ctx->stream);
  }
  if (e == hipSuccess) {
    e = hipMemsetAsync(h->d_poses, 0, (size_t) max_frames * sizeof(prs_frame_pose), ctx->stream);
  }
  if (e != hipSuccess) {
    release(h);
    delete h;
    return prs::ctx_fail_hip(ctx, e, "prs_map_create: device allocation");
  }
  *out = h;
  return PRS_OK;
}

int prs_map_destroy(prs_map* h) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(h->ctx->device);
  (void) hipStreamSynchronize(h->ctx->stream);
  release(h);
  delete h;
  return PRS_OK;
}

int prs_map_reserve(prs_map* h, int32_t capacity) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  if (capacity <= h->capacity) {
    return PRS_OK;
  }
  (void) hipSetDevice(h->ctx->device);
  hipStream_t s    = h->ctx->stream;
  const size_t cap = (size_t) capacity, n = (size_t) h->n_points;
  const size_t mm  = (size_t) (h->max_measurements > 0 ? h->max_measurements : 1);
  float *coords = nullptr, *state = nullptr, *cov = nullptr;
  uint8_t *desc = nullptr, *inlier = nullptr;
  uint32_t *n_opt = nullptr, *n_meas = nullptr;
  prs_camera_measurement* meas = nullptr;
  int32_t* index_map           = nullptr;
  hipError_t e = alloc(&coords, cap * 4);
  e = e == hipSuccess ? alloc(&desc, cap * PRS_DESC_BYTES) : e;
  e = e == hipSuccess ? alloc(&state, cap * 4) : e;
  e = e == hipSuccess ? alloc(&cov, cap * 9) : e;
  e = e == hipSuccess ? alloc(&n_opt, cap) : e;
  e = e == hipSuccess ? alloc(&inlier, cap) : e;
  e = e == hipSuccess ? alloc(&n_meas, cap) : e;
  e = e == hipSuccess ? alloc(&meas, cap * mm) : e;
  e = e == hipSuccess ? alloc(&index_map, cap) : e;
  auto copy = [&](void* dst, const void* src, size_t bytes) {
    if (e == hipSuccess && bytes) {
      e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    }
  };
  copy(coords, h->d_coords, n * 16);
  copy(desc, h->d_desc, n * PRS_DESC_BYTES);
  copy(state, h->d_state, n * 16);
  copy(cov, h->d_cov, n * 36);
  copy(n_opt, h->d_n_opt, n * 4);
  copy(inlier, h->d_inlier, n);
  copy(n_meas, h->d_n_meas, n * 4);
  copy(meas, h->d_meas, n * mm * sizeof(prs_camera_measurement));
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  void* fresh[] = {coords, desc, state, cov, n_opt, inlier, n_meas, meas, index_map};
  if (e != hipSuccess) {
    for (void* p : fresh) {
      if (p) {
        (void) hipFree(p);
      }
    }
    return prs::ctx_fail_hip(h->ctx, e, "prs_map_reserve: device allocation / copy");
  }
  void* old[] = {h->d_coords, h->d_desc, h->d_state, h->d_cov, h->d_n_opt, h->d_inlier, h->d_n_meas, h->d_meas, h->d_index_map};
  for (void* p : old) {
    (void) hipFree(p);
  }
  h->d_coords = coords;
  h->d_desc = desc;
  h->d_state = state;
  h->d_cov = cov;
  h->d_n_opt = n_opt;
  h->d_inlier = inlier;
  h->d_n_meas = n_meas;
  h->d_meas = meas;
  h->d_index_map = index_map;
  h->capacity = capacity;
  return PRS_OK;
}

int prs_map_clear(prs_map* h) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(h->ctx->device);
  MAP_TRY(hipMemsetAsync(h->d_small, 0, kSmall, h->ctx->stream));
  MAP_TRY(hipMemsetAsync(h->d_poses, 0, (size_t) h->max_frames * sizeof(prs_frame_pose), h->ctx->stream));
  h->n_points = 0;
  h->frames   = 0;
  return PRS_OK;
}

int prs_map_size(prs_map* h, int32_t* n_points, int32_t* frames_merged) {
  if (!h || !n_points) {
    return PRS_ERR_NULL;
  }
  *n_points = h->n_points;
  if (frames_merged) {
    *frames_merged = h->frames;
  }
  return PRS_OK;
}

int prs_map_set_scene(prs_map* h,
                      const float* coords_in_scene,
                      const float* state_in_world,
                      const float* covariance,
                      const uint8_t* desc,
                      const uint32_t* n_opt,
                      const prs_camera_measurement* first_measurement,
                      int32_t n) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  if (n < 0 || (n > 0 && (!coords_in_scene || !desc))) {
    return fail(h, PRS_ERR_NULL, "prs_map_set_scene: scene not set");  // merger_projective_impl.cpp:12-20
  }
  if (n > h->capacity) {
    return fail(h, PRS_ERR_CAPACITY, "prs_map_set_scene: scene larger than the map capacity");
  }
  if (first_measurement && h->max_measurements <= 0) {
    return fail(h, PRS_ERR_UNSUPPORTED, "prs_map_set_scene: the map keeps no measurement history");
  }
  (void) hipSetDevice(h->ctx->device);
  const size_t nn = (size_t) n;
  const size_t mm = (size_t) (h->max_measurements > 0 ? h->max_measurements : 1);
  std::vector<float> c4(nn * 4, 0.0f), s4(nn * 4, 0.0f), cov(nn * 9, 0.0f);
  std::vector<uint32_t> nopt(nn, 0u), nmeas(nn, 0u);
  std::vector<uint8_t> inl(nn, 0);
  std::vector<prs_camera_measurement> meas(first_measurement ? nn * mm : 0);
  if (first_measurement) {
    memset(meas.data(), 0, meas.size() * sizeof(prs_camera_measurement));
  }
  for (size_t i = 0; i < nn; ++i) {
    const float* st = state_in_world ? state_in_world : coords_in_scene;  // statistics().setState(coordinates()) (test_mergers.cpp:268-271)
    for (int k = 0; k < 3; ++k) {
      c4[4 * i + k] = coords_in_scene[3 * i + k];
      s4[4 * i + k] = st[3 * i + k];
    }
    for (int k = 0; k < 9; ++k) {
      cov[9 * i + k] = covariance ? covariance[9 * i + k] : 0.0f;  // statistics().allocate() without setCovariance (test_mergers.cpp:268-271)
    }
    nopt[i] = n_opt ? n_opt[i] : 0u;
    if (first_measurement) {
      meas[i * mm] = first_measurement[i];
      nmeas[i]     = 1u;
    }
  }
  hipStream_t s = h->ctx->stream;
  if (n > 0) {
    MAP_TRY(hipMemcpyAsync(h->d_coords, c4.data(), nn * 16, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_state, s4.data(), nn * 16, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_cov, cov.data(), nn * 36, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_desc, desc, nn * PRS_DESC_BYTES, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_n_opt, nopt.data(), nn * 4, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_inlier, inl.data(), nn, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_n_meas, nmeas.data(), nn * 4, hipMemcpyHostToDevice, s));
    if (first_measurement) {
      MAP_TRY(hipMemcpyAsync(h->d_meas, meas.data(), meas.size() * sizeof(prs_camera_measurement), hipMemcpyHostToDevice, s));
    }
  }
  const int32_t np = n;
  MAP_TRY(hipMemcpyAsync(small_of(h->d_small).n_points, &np, 4, hipMemcpyHostToDevice, s));
  MAP_TRY(hipStreamSynchronize(s));  // the staging vectors go out of scope
  h->n_points = n;
  return PRS_OK;
}

int prs_map_set_frame_pose(prs_map* h, int32_t frame, const float* sensor_in_world16) {
  if (!h || !sensor_in_world16) {
    return PRS_ERR_NULL;
  }
  if (frame < 0 || frame >= h->max_frames) {
    return fail(h, PRS_ERR_CAPACITY, "prs_map_set_frame_pose: frame outside the pose table");
  }
  (void) hipSetDevice(h->ctx->device);
  prs_frame_pose p;
  const float* T = sensor_in_world16;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 4; ++c) {
      p.sensor_in_world[4 * r + c] = T[4 * r + c];
    }
  }
  // inverse of an isometry: R^T, -R^T t
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      p.world_in_sensor[4 * r + c] = T[4 * c + r];
    }
    p.world_in_sensor[4 * r + 3] = -((T[0 + r] * T[3] + T[4 + r] * T[7]) + T[8 + r] * T[11]);
  }
  MAP_TRY(hipMemcpyAsync(h->d_poses + frame, &p, sizeof(p), hipMemcpyHostToDevice, h->ctx->stream));
  MAP_TRY(hipStreamSynchronize(h->ctx->stream));
  if (frame >= h->frames) {
    h->frames = frame + 1;
  }
  return PRS_OK;
}

int prs_map_merge(prs_map* h,
                  const prs_merger_params* params,
                  const float* measurement_in_world16,
                  const float* measurement_in_scene16,
                  const float* measurement,
                  const uint8_t* measurement_desc,
                  int32_t n_measured,
                  const prs_corr* corr,
                  int32_t n_corr,
                  const int32_t* scene_index_map,
                  int32_t corr_from_aligner,
                  prs_merge_result* result) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  // merger_projective_impl.cpp:12-27: scene, measurement, correspondences and the transforms have to be set
  if (!params || !measurement_in_world16 || !measurement_in_scene16) {
    return fail(h, PRS_ERR_NULL, "prs_map_merge: parameters / transforms not set");
  }
  if (n_measured < 0 || (n_measured > 0 && (!measurement || !measurement_desc))) {
    return fail(h, PRS_ERR_NULL, "prs_map_merge: measurement not set");
  }
  if (n_corr < 0 || (n_corr > 0 && !corr)) {
    return fail(h, PRS_ERR_NULL, "prs_map_merge: correspondences not set");
  }
  if (n_measured > h->max_measured || n_corr > h->max_measured) {
    return fail(h, PRS_ERR_CAPACITY, "prs_map_merge: more measurements / correspondences than the handle was created for");
  }
  if (h->frames >= h->max_frames) {
    return fail(h, PRS_ERR_CAPACITY, "prs_map_merge: pose table full (prs_map_clear starts a new local map)");
  }
  const int dim = params->estimator.measurement_dim;
  if (dim < 2 || dim > 4) {
    return fail(h, PRS_ERR_UNSUPPORTED, "prs_map_merge: measurement_dim must be 2, 3 or 4");
  }
  // host-side validation of the correspondence vector before anything is uploaded
  int32_t need = 0;
  for (int32_t i = 0; i < n_corr; ++i) {
    if (corr[i].fixed_idx < 0 || corr[i].moving_idx < 0) {
      return fail(h, PRS_ERR_RANGE, "prs_map_merge: negative index in the correspondence vector");
    }
    const int32_t ci = corr_from_aligner ? corr[i].moving_idx : corr[i].fixed_idx;
    need             = ci + 1 > need ? ci + 1 : need;
  }
  if (scene_index_map && need > h->capacity) {
    return fail(h, PRS_ERR_RANGE, "prs_map_merge: correspondence names a clipped index beyond the map capacity");
  }
  (void) hipSetDevice(h->ctx->device);
  hipStream_t s = h->ctx->stream;
  const size_t nm = (size_t) n_measured;
  std::vector<float> z4(nm * 4, 0.0f);
  for (size_t i = 0; i < nm; ++i) {
    for (int k = 0; k < dim; ++k) {
      z4[4 * i + k] = measurement[(size_t) dim * i + k];
    }
  }
  Small sm = small_of(h->d_small);
  if (n_measured > 0) {
    MAP_TRY(hipMemcpyAsync(h->d_measurement, z4.data(), nm * 16, hipMemcpyHostToDevice, s));
    MAP_TRY(hipMemcpyAsync(h->d_mdesc, measurement_desc, nm * PRS_DESC_BYTES, hipMemcpyHostToDevice, s));
  }
  if (n_corr > 0) {
    MAP_TRY(hipMemcpyAsync(h->d_corr, corr, (size_t) n_corr * sizeof(prs_corr), hipMemcpyHostToDevice, s));
  }
  if (scene_index_map && need > 0) {
    // clipped index -> scene index for every clipped point the correspondences may name
    MAP_TRY(hipMemcpyAsync(h->d_index_map, scene_index_map, (size_t) need * 4, hipMemcpyHostToDevice, s));
  }
  int32_t head[4] = {h->n_points, n_measured, n_corr, h->frames};
  MAP_TRY(hipMemcpyAsync(h->d_small, head, sizeof(head), hipMemcpyHostToDevice, s));
  MAP_TRY(hipMemcpyAsync(sm.in_world, measurement_in_world16, 64, hipMemcpyHostToDevice, s));
  MAP_TRY(hipMemcpyAsync(sm.in_scene, measurement_in_scene16, 64, hipMemcpyHostToDevice, s));
  prs_merge_batch b;
  memset(&b, 0, sizeof(b));
  b.batch = 1;
  b.capacity = h->capacity;
  b.max_measurements = h->max_measurements;
  b.max_frames = h->max_frames;
  b.coords = h->d_coords;
  b.desc = h->d_desc;
  b.state = h->d_state;
  b.covariance = h->d_cov;
  b.n_opt = h->d_n_opt;
  b.inlier = h->d_inlier;
  b.n_meas = h->d_n_meas;
  b.meas = h->max_measurements > 0 ? h->d_meas : nullptr;
  b.poses = h->d_poses;
  b.n_points = sm.n_points;
  b.measurement_stride = h->max_measured;
  b.measurement = h->d_measurement;
  b.measurement_desc = h->d_mdesc;
  b.n_measured = sm.n_measured;
  b.corr_stride = h->max_measured;
  b.corr = h->d_corr;
  b.n_corr = sm.n_corr;
  b.scene_index_map = scene_index_map ? h->d_index_map : nullptr;
  b.measurement_in_world = sm.in_world;
  b.measurement_in_scene = sm.in_scene;
  b.frame = sm.frame;
  b.result = sm.result;
  b.corr_from_aligner = corr_from_aligner ? 1 : 0;
  const int rc = prs::merge_batch_launch(h->ctx, params, &b);
  if (rc != PRS_OK) {
    (void) hipStreamSynchronize(s);  // (the staging vector of the measurements goes out of scope)
    return rc;
  }
  struct {
    int32_t n_points;
    int32_t pad[3];
    prs_merge_result res;
  } back;
  MAP_TRY(hipMemcpyAsync(&back, h->d_small, sizeof(back), hipMemcpyDeviceToHost, s));
  MAP_TRY(hipStreamSynchronize(s));
  if (result) {
    *result = back.res;
  }
  if (back.res.status < 0) {
    return fail(h, back.res.status, "prs_map_merge: the merge kernel reported an error (scene full, history full or duplicate scene index)");
  }
  h->n_points = back.n_points;
  ++h->frames;
  return back.res.status;
}

int prs_map_get_scene(prs_map* h, int32_t capacity, float* coords_in_scene, float* state_in_world, uint8_t* desc, uint32_t* n_opt,
                      uint8_t* inlier, int32_t* n_points) {
  if (!h || !n_points) {
    return PRS_ERR_NULL;
  }
  const int n = h->n_points;
  if (capacity < n) {
    return fail(h, PRS_ERR_CAPACITY, "prs_map_get_scene: output capacity below the map size");
  }
  (void) hipSetDevice(h->ctx->device);
  hipStream_t s = h->ctx->stream;
  const size_t nn = (size_t) n;
  std::vector<float> c4(nn * 4), s4(nn * 4);
  if (n > 0) {
    if (coords_in_scene) {
      MAP_TRY(hipMemcpyAsync(c4.data(), h->d_coords, nn * 16, hipMemcpyDeviceToHost, s));
    }
    if (state_in_world) {
      MAP_TRY(hipMemcpyAsync(s4.data(), h->d_state, nn * 16, hipMemcpyDeviceToHost, s));
    }
    if (desc) {
      MAP_TRY(hipMemcpyAsync(desc, h->d_desc, nn * PRS_DESC_BYTES, hipMemcpyDeviceToHost, s));
    }
    if (n_opt) {
      MAP_TRY(hipMemcpyAsync(n_opt, h->d_n_opt, nn * 4, hipMemcpyDeviceToHost, s));
    }
    if (inlier) {
      MAP_TRY(hipMemcpyAsync(inlier, h->d_inlier, nn, hipMemcpyDeviceToHost, s));
    }
  }
  MAP_TRY(hipStreamSynchronize(s));
  for (size_t i = 0; i < nn; ++i) {
    for (int k = 0; k < 3; ++k) {
      if (coords_in_scene) {
        coords_in_scene[3 * i + k] = c4[4 * i + k];
      }
      if (state_in_world) {
        state_in_world[3 * i + k] = s4[4 * i + k];
      }
    }
  }
  *n_points = n;
  return PRS_OK;
}

}  // extern "C"

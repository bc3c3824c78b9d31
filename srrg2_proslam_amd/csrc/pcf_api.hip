// pcf_api.hip -- host-pointer, single-frame entry points of the projective finder / aligner:
// a stateful handle mirroring the reference's CorrespondenceFinderProjective* object
// (setFixed / setMoving / setLocalMapInSensor / compute, CF/correspondence_finder_projective_base.h)
// on top of the batched device kernel (batch = 1).
#include <math.h>
#include <string.h>

#include <vector>

#include "prs_host.h"

struct prs_pcf {
  prs_context* ctx = nullptr;
  prs_pcf_params params;
  prs_pcf_state state;  // host mirror of the device-resident state
  bool state_dirty   = true;
  bool fixed_set     = false;
  bool moving_set    = false;
  bool inputs_changed = false;
  int n_fixed = 0, n_moving = 0, fixed_dim = 2;
  int cap_f = 0, cap_m = 0;
  float* d_fixed     = nullptr;
  uint8_t* d_fdesc   = nullptr;
  float* d_moving    = nullptr;
  uint8_t* d_mdesc   = nullptr;
  prs_corr* d_corr   = nullptr;
  unsigned char* d_small = nullptr;  // counts, state, X, result, changed, prior
  float local_map_in_sensor[16];
  std::vector<prs_corr> corr;  // the persisting correspondence vector (host mirror)
  int n_corr = 0;
  bool has_prior_mean = false;  // mean of the motion prior (prs_pcf_set_motion_prior_mean)
  float prior_mean[16];
};

namespace {

struct SmallLayout {
  int32_t* n_fixed;
  int32_t* n_moving;
  int32_t* n_corr;
  uint8_t* changed;
  prs_pcf_state* state;
  float* X;
  prs_align_result* result;
  float* prior;
  float* prior_mean;
  float* H;
  float* b;
  int* ok;
};

constexpr size_t kSmallBytes = 4096;

SmallLayout small_layout(unsigned char* d) {
  SmallLayout s;
  s.n_fixed  = reinterpret_cast<int32_t*>(d + 0);
  s.n_moving = reinterpret_cast<int32_t*>(d + 4);
  s.n_corr   = reinterpret_cast<int32_t*>(d + 8);
  s.changed  = reinterpret_cast<uint8_t*>(d + 12);
  s.state    = reinterpret_cast<prs_pcf_state*>(d + 64);
  s.X        = reinterpret_cast<float*>(d + 512);
  s.result   = reinterpret_cast<prs_align_result*>(d + 1024);
  s.prior    = reinterpret_cast<float*>(d + 2048);
  s.prior_mean = reinterpret_cast<float*>(d + 2304);
  s.H        = reinterpret_cast<float*>(d + 2560);
  s.b        = reinterpret_cast<float*>(d + 2816);
  s.ok       = reinterpret_cast<int*>(d + 2880);
  return s;
}

int fail(prs_pcf* h, int status, const char* what) {
  return prs::ctx_fail(h ? h->ctx : nullptr, status, what);
}

#define PCF_TRY(x)                                             \
  do {                                                         \
    hipError_t e_ = (x);                                       \
    if (e_ != hipSuccess) {                                    \
      return prs::ctx_fail_hip(h->ctx, e_, "prs_pcf: " #x);    \
    }                                                          \
  } while (0)

int ensure_capacity(prs_pcf* h, int nf, int nm) {
  if (!h->d_small) {
    PCF_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_small), kSmallBytes));
    PCF_TRY(hipMemset(h->d_small, 0, kSmallBytes));
  }
  if (nf > h->cap_f || !h->d_fixed) {
    const int cap = nf + nf / 4 + 16;
    if (h->d_fixed) {
      (void) hipFree(h->d_fixed);
      (void) hipFree(h->d_fdesc);
      (void) hipFree(h->d_corr);
    }
    PCF_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_fixed), sizeof(float) * 4 * (size_t) cap));
    PCF_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_fdesc), (size_t) PRS_DESC_BYTES * (size_t) cap));
    PCF_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_corr), sizeof(prs_corr) * (size_t) cap));
    h->cap_f = cap;
  }
  if (nm > h->cap_m || !h->d_moving) {
    const int cap = nm + nm / 4 + 16;
    if (h->d_moving) {
      (void) hipFree(h->d_moving);
      (void) hipFree(h->d_mdesc);
    }
    PCF_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_moving), sizeof(float) * 4 * (size_t) cap));
    PCF_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_mdesc), (size_t) PRS_DESC_BYTES * (size_t) cap));
    h->cap_m = cap;
  }
  return PRS_OK;
}

// one launch of the batch-1 kernel in `mode`; uploads dirty state, downloads state/result/correspondences
int run(prs_pcf* h, const prs_aligner_params* aligner, int mode, const float* X_in, const float* prior42,
        float* X_out, prs_align_result* result_out) {
  // _preCompute (CF/..bruteforce_impl.cpp:203-216): unset buffers are hard errors
  if (!h->fixed_set || !h->moving_set) {
    return fail(h, PRS_ERR_NULL, "prs_pcf: fixed or moving not set");
  }
  (void) hipSetDevice(h->ctx->device);
  hipStream_t s  = h->ctx->stream;
  SmallLayout sl = small_layout(h->d_small);
  int32_t counts[3] = {h->n_fixed, h->n_moving, h->n_corr};
  PCF_TRY(hipMemcpyAsync(sl.n_fixed, counts, sizeof(counts), hipMemcpyHostToDevice, s));
  const uint8_t changed = h->inputs_changed ? 1 : 0;
  PCF_TRY(hipMemcpyAsync(sl.changed, &changed, 1, hipMemcpyHostToDevice, s));
  if (h->state_dirty) {
    PCF_TRY(hipMemcpyAsync(sl.state, &h->state, sizeof(prs_pcf_state), hipMemcpyHostToDevice, s));
    h->state_dirty = false;
  }
  PCF_TRY(hipMemcpyAsync(sl.X, X_in, sizeof(float) * 16, hipMemcpyHostToDevice, s));
  if (prior42) {
    PCF_TRY(hipMemcpyAsync(sl.prior, prior42, sizeof(float) * 42, hipMemcpyHostToDevice, s));
  }
  prs_aligner_params ap;
  if (aligner) {
    ap = *aligner;
  } else {
    memset(&ap, 0, sizeof(ap));
    ap.factor_type = h->fixed_dim >= 2 && h->fixed_dim <= 4 ? h->fixed_dim : PRS_FACTOR_MONO;
  }
  prs_align_batch b;
  memset(&b, 0, sizeof(b));
  b.batch          = 1;
  b.fixed_stride   = h->cap_f > 0 ? h->cap_f : 1;
  b.moving_stride  = h->cap_m > 0 ? h->cap_m : 1;
  b.fixed          = h->d_fixed;
  b.fixed_desc     = h->d_fdesc;
  b.n_fixed        = sl.n_fixed;
  b.moving         = h->d_moving;
  b.moving_desc    = h->d_mdesc;
  b.n_moving       = sl.n_moving;
  b.inputs_changed = sl.changed;
  b.state          = sl.state;
  b.X              = sl.X;
  b.corr           = h->d_corr;
  b.n_corr         = sl.n_corr;
  b.result         = sl.result;
  b.prior          = prior42 ? sl.prior : nullptr;
  if (h->has_prior_mean) {
    PCF_TRY(hipMemcpyAsync(sl.prior_mean, h->prior_mean, sizeof(float) * 16, hipMemcpyHostToDevice, s));
    b.prior_mean = sl.prior_mean;
  }
  int rc           = prs::align_batch_launch(h->ctx, &h->params, &ap, &b, mode, 0);
  if (rc == PRS_OK) {
    rc = prs::align_batch_finish(h->ctx);
  }
  if (rc != PRS_OK) {
    return rc;
  }
  prs_align_result res;
  float X[16];
  int32_t n_corr = 0;
  PCF_TRY(hipMemcpyAsync(&res, sl.result, sizeof(res), hipMemcpyDeviceToHost, s));
  PCF_TRY(hipMemcpyAsync(&h->state, sl.state, sizeof(prs_pcf_state), hipMemcpyDeviceToHost, s));
  PCF_TRY(hipMemcpyAsync(X, sl.X, sizeof(X), hipMemcpyDeviceToHost, s));
  PCF_TRY(hipMemcpyAsync(&n_corr, sl.n_corr, sizeof(n_corr), hipMemcpyDeviceToHost, s));
  PCF_TRY(hipStreamSynchronize(s));
  if (res.warnings < 0) {
    return fail(h, res.warnings, "prs_pcf: fixed coordinates outside the projector canvas / int16 lattice, or bad correspondence index");
  }
  if (mode != PRS_MODE_LINEARIZE) {
    h->n_corr = n_corr;
    if ((int) h->corr.size() < h->cap_f) {
      h->corr.resize((size_t) h->cap_f);
    }
    if (n_corr > 0) {
      PCF_TRY(hipMemcpy(h->corr.data(), h->d_corr, sizeof(prs_corr) * (size_t) n_corr, hipMemcpyDeviceToHost));
    }
    h->inputs_changed = false;
    memcpy(h->local_map_in_sensor, h->state.local_map_in_sensor, sizeof(float) * 16);
  }
  if (X_out) {
    memcpy(X_out, X, sizeof(X));
  }
  if (result_out) {
    *result_out = res;
  }
  return res.warnings;
}

}  // namespace

extern "C" {

int prs_pcf_create(prs_context* ctx, const prs_pcf_params* params, prs_pcf** out) {
  if (!ctx || !params || !out) {
    return PRS_ERR_NULL;
  }
  prs_pcf* h = new prs_pcf();
  h->ctx     = ctx;
  h->params  = *params;
  memset(&h->state, 0, sizeof(h->state));
  h->state.config_changed = 1;  // CF/..projective_base.h:134
  for (int i = 0; i < 16; ++i) {
    const float v                          = (i % 5 == 0) ? 1.0f : 0.0f;
    h->state.local_map_in_sensor[i]          = v;
    h->state.local_map_in_sensor_previous[i] = v;
    h->local_map_in_sensor[i]                = v;
  }
  *out = h;
  return PRS_OK;
}

int prs_pcf_destroy(prs_pcf* h) {
  if (!h) {
    return PRS_OK;
  }
  (void) hipSetDevice(h->ctx->device);
  (void) hipStreamSynchronize(h->ctx->stream);
  void* bufs[] = {h->d_fixed, h->d_fdesc, h->d_moving, h->d_mdesc, h->d_corr, h->d_small};
  for (void* p : bufs) {
    if (p) {
      (void) hipFree(p);
    }
  }
  delete h;
  return PRS_OK;
}

int prs_pcf_set_params(prs_pcf* h, const prs_pcf_params* params) {
  if (!h || !params) {
    return PRS_ERR_NULL;
  }
  h->params               = *params;
  h->state.config_changed = 1;  // PARAM(.., &_config_changed), CF/..projective_base.h:30-44
  h->state_dirty          = true;
  return PRS_OK;
}

int prs_pcf_set_fixed(prs_pcf* h, const float* coords, int32_t fixed_dim, const uint8_t* desc, int32_t n) {
  if (!h || n < 0 || fixed_dim < 2 || fixed_dim > 4 || (n > 0 && (!coords || !desc))) {
    return fail(h, PRS_ERR_NULL, "prs_pcf_set_fixed: fixed not set");
  }
  (void) hipSetDevice(h->ctx->device);
  int rc = ensure_capacity(h, n, h->n_moving);
  if (rc != PRS_OK) {
    return rc;
  }
  if (n > 0) {
    std::vector<float> packed((size_t) n * 4, 0.0f);
    for (int i = 0; i < n; ++i) {
      for (int d = 0; d < fixed_dim; ++d) {
        packed[(size_t) i * 4 + d] = coords[(size_t) i * fixed_dim + d];
      }
    }
    PCF_TRY(hipMemcpy(h->d_fixed, packed.data(), sizeof(float) * 4 * (size_t) n, hipMemcpyHostToDevice));
    PCF_TRY(hipMemcpy(h->d_fdesc, desc, (size_t) PRS_DESC_BYTES * (size_t) n, hipMemcpyHostToDevice));
  }
  h->n_fixed        = n;
  h->fixed_dim      = fixed_dim;
  h->fixed_set      = true;
  h->inputs_changed = true;
  h->n_corr         = 0;
  return PRS_OK;
}

int prs_pcf_set_moving(prs_pcf* h, const float* xyz, const float* info_scale, const uint8_t* desc, int32_t n) {
  if (!h || n < 0 || (n > 0 && (!xyz || !desc))) {
    return fail(h, PRS_ERR_NULL, "prs_pcf_set_moving: moving not set");
  }
  (void) hipSetDevice(h->ctx->device);
  int rc = ensure_capacity(h, h->n_fixed, n);
  if (rc != PRS_OK) {
    return rc;
  }
  if (n > 0) {
    std::vector<float> packed((size_t) n * 4);
    for (int i = 0; i < n; ++i) {
      packed[(size_t) i * 4 + 0] = xyz[(size_t) i * 3 + 0];
      packed[(size_t) i * 4 + 1] = xyz[(size_t) i * 3 + 1];
      packed[(size_t) i * 4 + 2] = xyz[(size_t) i * 3 + 2];
      packed[(size_t) i * 4 + 3] = info_scale ? info_scale[i] : 1.0f;
    }
    PCF_TRY(hipMemcpy(h->d_moving, packed.data(), sizeof(float) * 4 * (size_t) n, hipMemcpyHostToDevice));
    PCF_TRY(hipMemcpy(h->d_mdesc, desc, (size_t) PRS_DESC_BYTES * (size_t) n, hipMemcpyHostToDevice));
  }
  h->n_moving       = n;
  h->moving_set     = true;
  h->inputs_changed = true;
  return PRS_OK;
}

int prs_pcf_set_local_map_in_sensor(prs_pcf* h, const float* T16) {
  if (!h || !T16) {
    return PRS_ERR_NULL;
  }
  memcpy(h->local_map_in_sensor, T16, sizeof(float) * 16);
  return PRS_OK;
}

int prs_pcf_set_search_radius(prs_pcf* h, uint64_t radius_pixels) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  h->state.search_radius_pixels = radius_pixels;  // CF/..projective_base.h:82-85
  h->state.config_changed       = 0;
  h->state_dirty                = true;
  return PRS_OK;
}

int prs_pcf_set_descriptor_distance(prs_pcf* h, float distance) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  h->state.descriptor_distance = distance;  // CF/..projective_base.h:94-97
  h->state.config_changed      = 0;
  h->state_dirty               = true;
  return PRS_OK;
}

int prs_pcf_get_state(prs_pcf* h, prs_pcf_state* out) {
  if (!h || !out) {
    return PRS_ERR_NULL;
  }
  *out = h->state;
  memcpy(out->local_map_in_sensor, h->local_map_in_sensor, sizeof(float) * 16);
  return PRS_OK;
}

int prs_pcf_set_motion_prior_mean(prs_pcf* h, const float* Z16) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  h->has_prior_mean = Z16 != nullptr;
  if (Z16) {
    memcpy(h->prior_mean, Z16, sizeof(float) * 16);
  }
  return PRS_OK;
}

int prs_pcf_compute(prs_pcf* h, prs_corr* out, int32_t capacity, int32_t* n_out) {
  if (!h || !out || !n_out) {
    return fail(h, PRS_ERR_NULL, "prs_pcf_compute: correspondences not set");
  }
  const int rc = run(h, nullptr, PRS_MODE_FINDER, h->local_map_in_sensor, nullptr, nullptr, nullptr);
  if (rc < 0) {
    return rc;
  }
  if (capacity < h->n_corr) {
    return fail(h, PRS_ERR_CAPACITY, "prs_pcf_compute: capacity < number of correspondences");
  }
  if (h->n_corr > 0) {
    memcpy(out, h->corr.data(), sizeof(prs_corr) * (size_t) h->n_corr);
  }
  *n_out = h->n_corr;
  return rc;
}

int prs_pcf_align(prs_pcf* h, const prs_aligner_params* aligner, const float* X_init16, const float* prior42, float* X_out16,
                  prs_corr* corr_out, int32_t capacity, int32_t* n_corr_out, prs_align_result* result) {
  if (!h || !aligner || !X_init16 || !X_out16 || !corr_out || !n_corr_out || !result) {
    return fail(h, PRS_ERR_NULL, "prs_pcf_align: argument not set");
  }
  const int rc = run(h, aligner, PRS_MODE_ALIGN, X_init16, prior42, X_out16, result);
  if (rc < 0) {
    return rc;
  }
  if (capacity < h->n_corr) {
    return fail(h, PRS_ERR_CAPACITY, "prs_pcf_align: capacity < number of correspondences");
  }
  if (h->n_corr > 0) {
    memcpy(corr_out, h->corr.data(), sizeof(prs_corr) * (size_t) h->n_corr);
  }
  *n_corr_out = h->n_corr;
  return rc;
}

int prs_pcf_linearize(prs_pcf* h, const prs_aligner_params* aligner, const float* X16, const prs_corr* corr, int32_t n_corr,
                      prs_align_result* result) {
  if (!h || !aligner || !X16 || !result || n_corr < 0 || (n_corr > 0 && !corr)) {
    return fail(h, PRS_ERR_NULL, "prs_pcf_linearize: argument not set");
  }
  if (!h->fixed_set || !h->moving_set) {
    return fail(h, PRS_ERR_NULL, "prs_pcf: fixed or moving not set");
  }
  if (n_corr > h->cap_f) {
    return fail(h, PRS_ERR_CAPACITY, "prs_pcf_linearize: more correspondences than fixed points");
  }
  (void) hipSetDevice(h->ctx->device);
  const int saved = h->n_corr;
  if (n_corr > 0) {
    PCF_TRY(hipMemcpy(h->d_corr, corr, sizeof(prs_corr) * (size_t) n_corr, hipMemcpyHostToDevice));
  }
  h->n_corr    = n_corr;
  const int rc = run(h, aligner, PRS_MODE_LINEARIZE, X16, nullptr, nullptr, result);
  // restore the finder's own vector on the device
  h->n_corr = saved;
  if (saved > 0) {
    PCF_TRY(hipMemcpy(h->d_corr, h->corr.data(), sizeof(prs_corr) * (size_t) saved, hipMemcpyHostToDevice));
  }
  return rc;
}

int prs_gn_step(prs_context* ctx, const float* H36, const float* b6, float damping, float* X16) {
  return prs_gn_step_ex(ctx, H36, b6, damping, PRS_DAMPING_DIAG, X16);
}

int prs_gn_step_ex(prs_context* ctx, const float* H36, const float* b6, float damping, int32_t damping_form, float* X16) {
  if (!ctx || !H36 || !b6 || !X16) {
    return PRS_ERR_NULL;
  }
  if (damping_form != PRS_DAMPING_DIAG && damping_form != PRS_DAMPING_IDENTITY) {
    return prs::ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_gn_step_ex: unknown damping form");
  }
  (void) hipSetDevice(ctx->device);
  float* d = static_cast<float*>(prs::ctx_device_scratch_slot(ctx, 2, 512));
  if (!d) {
    return prs::ctx_fail(ctx, PRS_ERR_HIP, "prs_gn_step: scratch allocation failed");
  }
  hipStream_t s = ctx->stream;
  hipError_t e  = hipMemcpyAsync(d, H36, sizeof(float) * 36, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    e = hipMemcpyAsync(d + 36, b6, sizeof(float) * 6, hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess) {
    e = hipMemcpyAsync(d + 48, X16, sizeof(float) * 16, hipMemcpyHostToDevice, s);
  }
  if (e != hipSuccess) {
    return prs::ctx_fail_hip(ctx, e, "prs_gn_step upload");
  }
  const int rc = prs::gn_step_launch(ctx, d, d + 36, damping, damping_form, d + 48, reinterpret_cast<int*>(d + 64));
  if (rc != PRS_OK) {
    return rc;
  }
  int ok = 0;
  e      = hipMemcpyAsync(X16, d + 48, sizeof(float) * 16, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    e = hipMemcpyAsync(&ok, d + 64, sizeof(int), hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    return prs::ctx_fail_hip(ctx, e, "prs_gn_step download");
  }
  return ok ? PRS_OK : 1;  // 1: system not positive definite, X unchanged
}

void prs_info_scale_from_nopt(const uint32_t* n_opt, int32_t n, float* scale) {
  // aligner_slice_processor_projective.cpp:46-52: diagonal_info *= (1 + std::log(n)), double narrowed to float
  for (int32_t i = 0; i < n; ++i) {
    scale[i] = n_opt[i] > 2 ? (float) (1.0 + log((double) n_opt[i])) : 1.0f;
  }
}

}  // extern "C"

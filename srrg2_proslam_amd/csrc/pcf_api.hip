// pcf_api.hip -- host-pointer, single-frame entry points of the projective finder / aligner:
// a stateful handle mirroring the reference's CorrespondenceFinderProjective* object
// (setFixed / setMoving / setLocalMapInSensor / compute, CF/correspondence_finder_projective_base.h)
// on top of the batched device kernel (batch = 1).
#include <math.h>
#include <string.h>

#include "prs_host.h"

// The handle keeps ONE block of device memory and a pinned host mirror with the same layout:
//   [ correspondences (cap_f x 12 B) | small: state, X, n_corr, result | header: counts, changed, prior, prior mean |
//     fixed: float4[cap_f], rows[cap_f] | moving: float4[cap_m], rows[cap_m] ]
// setFixed / setMoving pack straight into the mirror (no allocation, no copy command); a call uploads the dirty span of the
// mirror with ONE asynchronous copy from pinned memory, launches, and downloads [correspondences | small] with ONE copy.
struct prs_pcf {
  prs_context* ctx = nullptr;
  prs_pcf_params params;
  prs_pcf_state state;  // host copy of the device-resident state
  bool fixed_set     = false;
  bool moving_set    = false;
  bool inputs_changed = false;
  bool fixed_dirty = false, moving_dirty = false;  // packed into the mirror since the last upload
  int n_fixed = 0, n_moving = 0, fixed_dim = 2;
  int cap_f = 0, cap_m = 0;
  unsigned char* d_block = nullptr;
  unsigned char* h_block = nullptr;  // pinned
  size_t block_bytes = 0;
  size_t off_small = 0, off_head = 0, off_fixed = 0, off_fdesc = 0, off_moving = 0, off_mdesc = 0;
  float local_map_in_sensor[16];
  int n_corr = 0;  // size of the persisting correspondence vector (its host copy: the front of h_block)
  bool has_prior_mean = false;  // mean of the motion prior (prs_pcf_set_motion_prior_mean)
  float prior_mean[16];
};

namespace {

// the small in / out words (offsets from off_small) and the input-only header (offsets from off_head)
constexpr size_t kSmState = 0, kSmX = 192, kSmNcorr = 256, kSmResult = 320, kSmallBytes = 640;
constexpr size_t kHdNfixed = 0, kHdNmoving = 4, kHdChanged = 8, kHdPrior = 64, kHdPriorMean = 256, kHeadBytes = 384;
static_assert(sizeof(prs_pcf_state) <= kSmX - kSmState && sizeof(prs_align_result) <= kSmallBytes - kSmResult, "small block layout");

size_t align256(size_t v) {
  return (v + 255) / 256 * 256;
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

prs_corr* host_corr(prs_pcf* h) {
  return reinterpret_cast<prs_corr*>(h->h_block);
}

// (re)allocates the block for nf fixed and nm moving points; the mirror's contents survive
int ensure_capacity(prs_pcf* h, int nf, int nm) {
  if (h->d_block && nf <= h->cap_f && nm <= h->cap_m) {
    return PRS_OK;
  }
  // (generous headroom: a reallocation costs milliseconds, and the counts of a sequence wander by tens of per cent)
  // (never zero: an empty cloud still gets 256 slots, so the correspondence region and the small block cannot alias and fixed_stride is real)
  const int cap_f = nf > h->cap_f || !h->d_block ? ((nf + nf / 2) / 256 + 1) * 256 : h->cap_f;
  const int cap_m = nm > h->cap_m || !h->d_block ? ((nm + nm / 2) / 256 + 1) * 256 : h->cap_m;
  prs_pcf n        = *h;
  n.cap_f          = cap_f;
  n.cap_m          = cap_m;
  n.off_small      = align256(sizeof(prs_corr) * (size_t) cap_f);
  n.off_head       = n.off_small + kSmallBytes;
  n.off_fixed      = align256(n.off_head + kHeadBytes);
  n.off_fdesc      = n.off_fixed + sizeof(float) * 4 * (size_t) cap_f;
  n.off_moving     = align256(n.off_fdesc + (size_t) PRS_DESC_BYTES * (size_t) cap_f);
  n.off_mdesc      = n.off_moving + sizeof(float) * 4 * (size_t) cap_m;
  n.block_bytes    = align256(n.off_mdesc + (size_t) PRS_DESC_BYTES * (size_t) cap_m);
  n.d_block        = nullptr;
  n.h_block        = nullptr;
  (void) hipStreamSynchronize(h->ctx->stream);
  PCF_TRY(hipMalloc(reinterpret_cast<void**>(&n.d_block), n.block_bytes));
  if (hipHostMalloc(reinterpret_cast<void**>(&n.h_block), n.block_bytes, hipHostMallocDefault) != hipSuccess) {
    (void) hipFree(n.d_block);
    return fail(h, PRS_ERR_HIP, "prs_pcf: pinned staging allocation failed");
  }
  memset(n.h_block, 0, n.block_bytes);
  if (h->h_block) {
    memcpy(n.h_block, h->h_block, sizeof(prs_corr) * (size_t) h->n_corr);
    memcpy(n.h_block + n.off_fixed, h->h_block + h->off_fixed, sizeof(float) * 4 * (size_t) h->n_fixed);
    memcpy(n.h_block + n.off_fdesc, h->h_block + h->off_fdesc, (size_t) PRS_DESC_BYTES * (size_t) h->n_fixed);
    memcpy(n.h_block + n.off_moving, h->h_block + h->off_moving, sizeof(float) * 4 * (size_t) h->n_moving);
    memcpy(n.h_block + n.off_mdesc, h->h_block + h->off_mdesc, (size_t) PRS_DESC_BYTES * (size_t) h->n_moving);
    (void) hipHostFree(h->h_block);
    (void) hipFree(h->d_block);
  }
  n.fixed_dirty  = h->fixed_set;
  n.moving_dirty = h->moving_set;
  *h             = n;
  // everything the device block holds is stale: upload the whole mirror (incl. the persisting correspondence vector) next time
  PCF_TRY(hipMemcpyAsync(h->d_block, h->h_block, h->off_small, hipMemcpyHostToDevice, h->ctx->stream));
  return PRS_OK;
}

// one call of the batch-1 pipeline in `mode`: ONE upload of the dirty span of the mirror, the launches, ONE download
int run(prs_pcf* h, const prs_aligner_params* aligner, int mode, const float* X_in, const float* prior42,
        float* X_out, prs_align_result* result_out) {
  // _preCompute (CF/..bruteforce_impl.cpp:203-216): unset buffers are hard errors
  if (!h->fixed_set || !h->moving_set) {
    return fail(h, PRS_ERR_NULL, "prs_pcf: fixed or moving not set");
  }
  (void) hipSetDevice(h->ctx->device);
  hipStream_t s     = h->ctx->stream;
  unsigned char* hb = h->h_block;
  unsigned char* db = h->d_block;
  // ---- header + small words into the mirror
  memcpy(hb + h->off_small + kSmState, &h->state, sizeof(prs_pcf_state));
  memcpy(hb + h->off_small + kSmX, X_in, sizeof(float) * 16);
  memcpy(hb + h->off_small + kSmNcorr, &h->n_corr, sizeof(int32_t));
  int32_t counts[2] = {h->n_fixed, h->n_moving};
  memcpy(hb + h->off_head + kHdNfixed, counts, sizeof(counts));
  hb[h->off_head + kHdChanged] = h->inputs_changed ? 1 : 0;
  if (prior42) {
    memcpy(hb + h->off_head + kHdPrior, prior42, sizeof(float) * 42);
  }
  if (h->has_prior_mean) {
    memcpy(hb + h->off_head + kHdPriorMean, h->prior_mean, sizeof(float) * 16);
  }
  size_t hi = h->off_head + kHeadBytes;
  if (h->fixed_dirty) {
    hi = h->off_fdesc + (size_t) PRS_DESC_BYTES * (size_t) h->n_fixed;
  }
  if (h->moving_dirty) {
    hi = h->off_mdesc + (size_t) PRS_DESC_BYTES * (size_t) h->n_moving;  // (a clean fixed region in between is uploaded again: same bytes)
  }
  PCF_TRY(hipMemcpyAsync(db + h->off_small, hb + h->off_small, hi - h->off_small, hipMemcpyHostToDevice, s));
  h->fixed_dirty = h->moving_dirty = false;
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
  b.fixed          = reinterpret_cast<const float*>(db + h->off_fixed);
  b.fixed_desc     = db + h->off_fdesc;
  b.n_fixed        = reinterpret_cast<const int32_t*>(db + h->off_head + kHdNfixed);
  b.moving         = reinterpret_cast<const float*>(db + h->off_moving);
  b.moving_desc    = db + h->off_mdesc;
  b.n_moving       = reinterpret_cast<const int32_t*>(db + h->off_head + kHdNmoving);
  b.inputs_changed = db + h->off_head + kHdChanged;
  b.state          = reinterpret_cast<prs_pcf_state*>(db + h->off_small + kSmState);
  b.X              = reinterpret_cast<float*>(db + h->off_small + kSmX);
  b.corr           = reinterpret_cast<prs_corr*>(db);
  b.n_corr         = reinterpret_cast<int32_t*>(db + h->off_small + kSmNcorr);
  b.result         = reinterpret_cast<prs_align_result*>(db + h->off_small + kSmResult);
  b.prior          = prior42 ? reinterpret_cast<const float*>(db + h->off_head + kHdPrior) : nullptr;
  b.prior_mean     = h->has_prior_mean ? reinterpret_cast<const float*>(db + h->off_head + kHdPriorMean) : nullptr;
  b.max_fixed      = (h->n_fixed + 63) / 64 * 64 > 0 ? (h->n_fixed + 63) / 64 * 64 : 64;  // the kernels' LDS is sized for this frame, not for the block's capacity
  if (b.max_fixed > b.fixed_stride) {
    b.max_fixed = b.fixed_stride;
  }
  int rc           = prs::align_batch_launch(h->ctx, &h->params, &ap, &b, mode, 0);
  if (rc == PRS_OK) {
    rc = prs::align_batch_finish(h->ctx);
  }
  if (rc != PRS_OK) {
    return rc;
  }
  // ---- ONE download: [correspondences | state, X, n_corr, result]; a linearisation leaves the correspondences alone.
  // (The span covers the block's capacity, not n_corr -- which is only known once it has arrived: at most (1.5 n_fixed + 256) x 12 B,
  // ~13 kB for a KITTI frame = 0.2 us of a 56 GB/s link, against a second copy + synchronisation of ~10 us for the exact count.)
  const size_t lo = mode == PRS_MODE_LINEARIZE ? h->off_small : 0;
  PCF_TRY(hipMemcpyAsync(hb + lo, db + lo, h->off_small + kSmallBytes - lo, hipMemcpyDeviceToHost, s));
  PCF_TRY(hipStreamSynchronize(s));
  prs_align_result res;
  int32_t n_corr = 0;
  memcpy(&res, hb + h->off_small + kSmResult, sizeof(res));
  memcpy(&h->state, hb + h->off_small + kSmState, sizeof(prs_pcf_state));
  memcpy(&n_corr, hb + h->off_small + kSmNcorr, sizeof(n_corr));
  if (res.warnings < 0) {
    return fail(h, res.warnings, "prs_pcf: fixed coordinates outside the projector canvas / int16 lattice, or bad correspondence index");
  }
  if (mode != PRS_MODE_LINEARIZE) {
    h->n_corr         = n_corr;
    h->inputs_changed = false;
    memcpy(h->local_map_in_sensor, h->state.local_map_in_sensor, sizeof(float) * 16);
  }
  if (X_out) {
    memcpy(X_out, hb + h->off_small + kSmX, sizeof(float) * 16);
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
  if (h->d_block) {
    (void) hipFree(h->d_block);
  }
  if (h->h_block) {
    (void) hipHostFree(h->h_block);
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
  float* packed = reinterpret_cast<float*>(h->h_block + h->off_fixed);
  for (int i = 0; i < n; ++i) {
    for (int d = 0; d < 4; ++d) {
      packed[(size_t) i * 4 + d] = d < fixed_dim ? coords[(size_t) i * fixed_dim + d] : 0.0f;
    }
  }
  if (n > 0) {
    memcpy(h->h_block + h->off_fdesc, desc, (size_t) PRS_DESC_BYTES * (size_t) n);
  }
  h->fixed_dirty    = true;
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
  float* packed = reinterpret_cast<float*>(h->h_block + h->off_moving);
  for (int i = 0; i < n; ++i) {
    packed[(size_t) i * 4 + 0] = xyz[(size_t) i * 3 + 0];
    packed[(size_t) i * 4 + 1] = xyz[(size_t) i * 3 + 1];
    packed[(size_t) i * 4 + 2] = xyz[(size_t) i * 3 + 2];
    packed[(size_t) i * 4 + 3] = info_scale ? info_scale[i] : 1.0f;
  }
  if (n > 0) {
    memcpy(h->h_block + h->off_mdesc, desc, (size_t) PRS_DESC_BYTES * (size_t) n);
  }
  h->moving_dirty   = true;
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
  return PRS_OK;
}

int prs_pcf_set_descriptor_distance(prs_pcf* h, float distance) {
  if (!h) {
    return PRS_ERR_NULL;
  }
  h->state.descriptor_distance = distance;  // CF/..projective_base.h:94-97
  h->state.config_changed      = 0;
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
    memcpy(out, host_corr(h), sizeof(prs_corr) * (size_t) h->n_corr);
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
    memcpy(corr_out, host_corr(h), sizeof(prs_corr) * (size_t) h->n_corr);
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
  // the caller's correspondences take the place of the finder's own vector on the device for this launch only (the host copy of
  // that vector stays in the mirror and is put back afterwards)
  const int saved = h->n_corr;
  if (n_corr > 0) {
    PCF_TRY(hipMemcpyAsync(h->d_block, corr, sizeof(prs_corr) * (size_t) n_corr, hipMemcpyHostToDevice, h->ctx->stream));
  }
  h->n_corr    = n_corr;
  const int rc = run(h, aligner, PRS_MODE_LINEARIZE, X16, nullptr, nullptr, result);
  h->n_corr    = saved;
  if (saved > 0) {
    PCF_TRY(hipMemcpyAsync(h->d_block, host_corr(h), sizeof(prs_corr) * (size_t) saved, hipMemcpyHostToDevice, h->ctx->stream));
  }
  return rc;
}

int prs_gn_step(prs_context* ctx, const float* H36, const float* b6, float damping, float* X16) {
  return prs_gn_step_ex(ctx, H36, b6, damping, PRS_DAMPING_DIAG, X16);
}

int prs_selftest_reciprocal(prs_context* ctx, uint64_t counts[2]) {
  if (!ctx || !counts) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  unsigned long long* d = static_cast<unsigned long long*>(prs::ctx_device_scratch_slot(ctx, 2, 512));
  if (!d) {
    return prs::ctx_fail(ctx, PRS_ERR_HIP, "prs_selftest_reciprocal: scratch allocation failed");
  }
  hipStream_t s = ctx->stream;
  hipError_t e  = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), s);
  if (e != hipSuccess) {
    return prs::ctx_fail_hip(ctx, e, "prs_selftest_reciprocal");
  }
  const int rc = prs::recip_selftest_launch(ctx, d);
  if (rc != PRS_OK) {
    return rc;
  }
  unsigned long long h[2] = {0, 0};
  e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    return prs::ctx_fail_hip(ctx, e, "prs_selftest_reciprocal download");
  }
  counts[0] = h[0];
  counts[1] = h[1];
  return PRS_OK;
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

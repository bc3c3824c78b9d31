// api.hip -- extern "C" boundary of libproslam_hip.so (declared in include/proslam_hip.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "prs_host.h"

namespace prs {

int ctx_fail(prs_context* ctx, int status, const char* what) {
  if (ctx) {
    ctx->last_error = what ? what : "";
  }
  return status;
}

int ctx_fail_hip(prs_context* ctx, hipError_t e, const char* what) {
  if (ctx) {
    ctx->last_error = std::string(what ? what : "") + ": " + hipGetErrorString(e);
  }
  return PRS_ERR_HIP;
}

void* ctx_device_scratch(prs_context* ctx, size_t bytes) {
  if (bytes <= ctx->d_scratch_size) {
    return ctx->d_scratch;
  }
  if (ctx->d_scratch) {
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipFree(ctx->d_scratch);
    ctx->d_scratch      = nullptr;
    ctx->d_scratch_size = 0;
  }
  size_t want = bytes + bytes / 2 + 4096;
  if (hipMalloc(&ctx->d_scratch, want) != hipSuccess) {
    ctx->d_scratch = nullptr;
    return nullptr;
  }
  ctx->d_scratch_size = want;
  return ctx->d_scratch;
}

const float* ctx_info_scale_table(prs_context* ctx) {
  if (!ctx->d_info_lut) {
    constexpr int kN = 4096;
    uint32_t n[kN];
    float scale[kN];
    for (int i = 0; i < kN; ++i) {
      n[i] = (uint32_t) i;
    }
    prs_info_scale_from_nopt(n, kN, scale);
    float* d = nullptr;
    if (hipMalloc(&d, sizeof(scale)) != hipSuccess) {
      return nullptr;
    }
    if (hipMemcpy(d, scale, sizeof(scale), hipMemcpyHostToDevice) != hipSuccess) {
      (void) hipFree(d);
      return nullptr;
    }
    ctx->d_info_lut = d;
  }
  return ctx->d_info_lut;
}

void* ctx_device_scratch_slot(prs_context* ctx, int slot, size_t bytes) {
  if (bytes <= ctx->d_slot_size[slot]) {
    return ctx->d_slot[slot];
  }
  if (ctx->d_slot[slot]) {
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipFree(ctx->d_slot[slot]);
    ctx->d_slot[slot]      = nullptr;
    ctx->d_slot_size[slot] = 0;
  }
  size_t want = bytes + bytes / 4 + 4096;
  if (hipMalloc(&ctx->d_slot[slot], want) != hipSuccess) {
    ctx->d_slot[slot] = nullptr;
    return nullptr;
  }
  ctx->d_slot_size[slot] = want;
  return ctx->d_slot[slot];
}

void* ctx_pinned_scratch(prs_context* ctx, size_t bytes) {
  if (bytes <= ctx->h_pinned_size) {
    return ctx->h_pinned;
  }
  if (ctx->h_pinned) {
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipHostFree(ctx->h_pinned);
    ctx->h_pinned      = nullptr;
    ctx->h_pinned_size = 0;
  }
  size_t want = bytes + bytes / 2 + 4096;
  if (hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault) != hipSuccess) {
    ctx->h_pinned = nullptr;
    return nullptr;
  }
  ctx->h_pinned_size = want;
  return ctx->h_pinned;
}

unsigned long long* ctx_stamps(prs_context* ctx, size_t bytes) {
  if (!ctx->stamps_enabled) {
    return nullptr;
  }
  if (bytes > ctx->d_stamps_size) {
    if (ctx->d_stamps) {
      (void) hipFree(ctx->d_stamps);
    }
    if (hipMalloc(reinterpret_cast<void**>(&ctx->d_stamps), bytes) != hipSuccess) {
      ctx->d_stamps      = nullptr;
      ctx->d_stamps_size = 0;
      return nullptr;
    }
    ctx->d_stamps_size = bytes;
  }
  return ctx->d_stamps;
}

void ctx_report_stamps(prs_context* ctx, int blocks, int n_stamps, const char* legend, bool raw, size_t first_block) {
  (void) hipStreamSynchronize(ctx->stream);
  std::vector<unsigned long long> h((size_t) blocks * 16);
  (void) hipMemcpy(h.data(), ctx->d_stamps + first_block * 16, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  fprintf(stderr, "[prs stamps] %s\n[prs stamps] mean cycles per phase over %d blocks:", legend, blocks);
  double total = 0;
  for (int i = 1; i < n_stamps; ++i) {
    double acc = 0;
    for (int b = 0; b < blocks; ++b) {
      acc += raw ? (double) h[(size_t) b * 16 + i] : (double) (h[(size_t) b * 16 + i] - h[(size_t) b * 16 + i - 1]);
    }
    fprintf(stderr, " %.0f", acc / blocks);
    total += acc / blocks;
  }
  fprintf(stderr, " | total %.0f\n", total);
}

static inline size_t align256(size_t v) {
  return (v + 255) / 256 * 256;
}

} // namespace prs

using namespace prs;

extern "C" {

int prs_version(void) {
  return PRS_ABI_VERSION;  // prs_aligner_params grew at its end: round 5 kernel_weight_form, damping_form, translation_weight_form; round 6 step_norm_exit
}

int prs_abi_check(int32_t header_version, uint64_t sizeof_stereo_params, uint64_t sizeof_pcf_params, uint64_t sizeof_aligner_params,
                  uint64_t sizeof_align_batch) {
  // (no context: the answer is the status alone)
  if (header_version != PRS_ABI_VERSION || sizeof_stereo_params != sizeof(prs_stereo_params) || sizeof_pcf_params != sizeof(prs_pcf_params) ||
      sizeof_aligner_params != sizeof(prs_aligner_params) || sizeof_align_batch != sizeof(prs_align_batch)) {
    return PRS_ERR_UNSUPPORTED;
  }
  return PRS_OK;
}

const char* prs_status_string(int status) {
  switch (status) {
    case PRS_OK: return "ok";
    case PRS_ERR_NULL: return "required input/output buffer not set";
    case PRS_ERR_CAPACITY: return "output capacity too small";
    case PRS_ERR_HIP: return "HIP runtime error";
    case PRS_ERR_RANGE: return "input outside the supported coordinate domain";
    case PRS_ERR_UNSUPPORTED: return "size beyond kernel limits";
    case PRS_ERR_NO_DEVICE: return "no HIP device";
    case PRS_ERR_HISTORY: return "measurement history of a landmark is full";
    case PRS_ERR_SCENE_FULL: return "map capacity exhausted";
    case PRS_ERR_DUPLICATE: return "scene index referenced by two correspondences";
    default: return status > 0 ? "warning bits set" : "unknown error";
  }
}

int prs_context_create(int device_id, prs_context** out) {
  if (!out) {
    return PRS_ERR_NULL;
  }
  *out      = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    return PRS_ERR_NO_DEVICE;  // loud: there is no CPU fallback behind this library
  }
  if (device_id < 0 || device_id >= count) {
    return PRS_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device_id) != hipSuccess) {
    return PRS_ERR_HIP;
  }
  prs_context* ctx = new prs_context();
  ctx->device      = device_id;
  if (hipStreamCreateWithFlags(&ctx->own, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return PRS_ERR_HIP;
  }
  ctx->stream          = ctx->own;
  const char* unstaged = getenv("PRS_FORCE_UNSTAGED");
  ctx->force_unstaged  = unstaged && unstaged[0] == '1';
  const char* v3       = getenv("PRS_MATCHER_V3");
  ctx->matcher_v3      = v3 && v3[0] == '1';
  const char* fused    = getenv("PRS_FUSED_ALIGN");
  ctx->fused_align     = fused && fused[0] == '1';
  const char* stamps   = getenv("PRS_STAMPS");
  ctx->stamps_enabled  = stamps && stamps[0] == '1';
  const char* ssplit   = getenv("PRS_STAMPS_SPLIT");
  ctx->stamps_split    = ssplit && ssplit[0] == '1';
  const char* nopre    = getenv("PRS_NO_PREFILTER");
  ctx->no_prefilter    = nopre && nopre[0] == '1';
  const char* nolone   = getenv("PRS_NO_LONE_GN");
  ctx->no_lone_gn      = nolone && nolone[0] == '1';
  const char* p96      = getenv("PRS_PREFILTER_96_LIMIT");
  ctx->prefilter_96_limit = p96 ? atoi(p96) : 32;
  const char* bfm      = getenv("PRS_BF_MFMA");
  ctx->bf_mfma         = !bfm || bfm[0] == 'a' ? PRS_BF_DENSE_MATRIX_WHEN_FULL : (bfm[0] == '1' ? PRS_BF_DENSE_MATRIX : PRS_BF_DENSE_POPCOUNT);
  const char* mfused   = getenv("PRS_MERGE_FUSED");
  ctx->merge_fused     = mfused && mfused[0] == '1';
  *out                 = ctx;
  return PRS_OK;
}

int prs_context_destroy(prs_context* ctx) {
  if (!ctx) {
    return PRS_OK;
  }
  (void) hipSetDevice(ctx->device);
  (void) hipStreamSynchronize(ctx->stream);
  if (ctx->align_job && ctx->align_job_free) {
    ctx->align_job_free(ctx->align_job);
  }
  if (ctx->d_scratch) {
    (void) hipFree(ctx->d_scratch);
  }
  if (ctx->h_pinned) {
    (void) hipHostFree(ctx->h_pinned);
  }
  if (ctx->d_stamps) {
    (void) hipFree(ctx->d_stamps);
  }
  if (ctx->d_info_lut) {
    (void) hipFree(ctx->d_info_lut);
  }
  for (int i = 0; i < 4; ++i) {
    if (ctx->d_slot[i]) {
      (void) hipFree(ctx->d_slot[i]);
    }
  }
  if (ctx->own) {
    (void) hipStreamDestroy(ctx->own);
  }
  for (hipEvent_t e : ctx->timing_ev) {
    if (e) {
      (void) hipEventDestroy(e);
    }
  }
  delete ctx;
  return PRS_OK;
}

int prs_context_set_bruteforce_dense_phase(prs_context* ctx, int32_t mode) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  if (mode != PRS_BF_DENSE_POPCOUNT && mode != PRS_BF_DENSE_MATRIX_WHEN_FULL && mode != PRS_BF_DENSE_MATRIX) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_context_set_bruteforce_dense_phase: mode must be one of PRS_BF_DENSE_*");
  }
  ctx->bf_mfma = mode;
  return PRS_OK;
}

int prs_context_enable_timing(prs_context* ctx, int32_t on) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  ctx->timing      = on != 0;
  ctx->t_search_ms = ctx->t_gn_ms = 0.0;
  ctx->n_search = ctx->n_gn = 0;
  for (int i = 0; i < 16; ++i) {
    ctx->t_search_round[i] = ctx->t_gn_round[i] = 0.0;
  }
  ctx->n_batches_timed = 0;
  return PRS_OK;
}

int prs_context_get_align_round_timing(prs_context* ctx, double* search_ms16, double* gn_ms16, int64_t* batches) {
  if (!ctx || !search_ms16 || !gn_ms16 || !batches) {
    return PRS_ERR_NULL;
  }
  for (int i = 0; i < 16; ++i) {
    search_ms16[i] = ctx->t_search_round[i];
    gn_ms16[i]     = ctx->t_gn_round[i];
  }
  *batches = ctx->n_batches_timed;
  return PRS_OK;
}

int prs_context_get_align_timing(prs_context* ctx, double* search_ms, double* gn_ms, int64_t* search_launches, int64_t* gn_launches) {
  if (!ctx || !search_ms || !gn_ms || !search_launches || !gn_launches) {
    return PRS_ERR_NULL;
  }
  *search_ms       = ctx->t_search_ms;
  *gn_ms           = ctx->t_gn_ms;
  *search_launches = ctx->n_search;
  *gn_launches     = ctx->n_gn;
  return PRS_OK;
}

int prs_context_set_stream(prs_context* ctx, void* hip_stream) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  if (align_job_active(ctx)) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_context_set_stream: an enqueued aligner batch has not been finished (prs_align_batch_finish)");
  }
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = HIP's default (null) stream
  return PRS_OK;
}

int prs_context_use_own_stream(prs_context* ctx) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  ctx->stream = ctx->own;
  return PRS_OK;
}

int prs_context_synchronize(prs_context* ctx) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  return e == hipSuccess ? PRS_OK : ctx_fail_hip(ctx, e, "hipStreamSynchronize");
}

const char* prs_last_error(const prs_context* ctx) {
  return ctx ? ctx->last_error.c_str() : "null context";
}

int prs_stereo_match_batch(prs_context* ctx, const prs_stereo_params* params, const prs_stereo_batch* batch) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return stereo_match_batch_launch(ctx, params, batch);
}

int prs_stereo_match(prs_context* ctx,
                     const prs_stereo_params* params,
                     const prs_kp2* left,
                     const uint8_t* desc_left,
                     int32_t n_left,
                     const prs_kp2* right,
                     const uint8_t* desc_right,
                     int32_t n_right,
                     prs_corr* out,
                     int32_t capacity,
                     int32_t* n_out) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  // _preCompute contract (CF/..bruteforce_impl.cpp:203-216): unset buffers are hard errors
  if (!params || !out || !n_out || n_left < 0 || n_right < 0 || (n_left > 0 && (!left || !desc_left)) ||
      (n_right > 0 && (!right || !desc_right))) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_stereo_match: fixed, moving or correspondences not set");
  }
  if (capacity < n_left) {
    return ctx_fail(ctx, PRS_ERR_CAPACITY, "prs_stereo_match: capacity < n_left");
  }
  (void) hipSetDevice(ctx->device);
  *n_out           = 0;
  const int stride = n_left > n_right ? (n_left > 0 ? n_left : 1) : (n_right > 0 ? n_right : 1);
  // one block on the device and a pinned mirror with the same layout: [ matches | meta | left kp | right kp | left rows | right rows ];
  // ONE upload of [meta .. right rows] from pinned memory, the launch, ONE download of [matches | meta]
  const size_t sz_kp   = align256(sizeof(prs_kp2) * (size_t) stride);
  const size_t sz_desc = align256((size_t) PRS_DESC_BYTES * (size_t) stride);
  const size_t sz_corr = align256(sizeof(prs_corr) * (size_t) stride);
  const size_t off_meta = sz_corr, off_kpl = off_meta + 256, off_kpr = off_kpl + sz_kp, off_dl = off_kpr + sz_kp, off_dr = off_dl + sz_desc;
  const size_t total    = off_dr + sz_desc;
  unsigned char* d      = static_cast<unsigned char*>(ctx_device_scratch(ctx, total));
  unsigned char* h      = static_cast<unsigned char*>(ctx_pinned_scratch(ctx, total));
  if (!d || !h) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_stereo_match: scratch allocation failed");
  }
  int32_t* d_meta = reinterpret_cast<int32_t*>(d + off_meta);  // n_left, n_right, n_matches, status
  int32_t* h_meta = reinterpret_cast<int32_t*>(h + off_meta);
  hipStream_t s   = ctx->stream;
  hipError_t e    = hipSuccess;
  h_meta[0] = n_left, h_meta[1] = n_right, h_meta[2] = 0, h_meta[3] = 0;
  if (n_left > 0) {
    memcpy(h + off_kpl, left, sizeof(prs_kp2) * (size_t) n_left);
    memcpy(h + off_dl, desc_left, (size_t) PRS_DESC_BYTES * (size_t) n_left);
  }
  if (n_right > 0) {
    memcpy(h + off_kpr, right, sizeof(prs_kp2) * (size_t) n_right);
    memcpy(h + off_dr, desc_right, (size_t) PRS_DESC_BYTES * (size_t) n_right);
  }
#define PRS_TRY(x)                                  \
  do {                                              \
    e = (x);                                        \
    if (e != hipSuccess) {                          \
      return ctx_fail_hip(ctx, e, "prs_stereo_match"); \
    }                                               \
  } while (0)
  PRS_TRY(hipMemcpyAsync(d + off_meta, h + off_meta, off_dr + (size_t) PRS_DESC_BYTES * (size_t) (n_right > 0 ? n_right : 0) - off_meta, hipMemcpyHostToDevice, s));
  prs_stereo_batch b;
  memset(&b, 0, sizeof(b));
  b.batch      = 1;
  b.stride     = stride;
  b.left_kp    = reinterpret_cast<prs_kp2*>(d + off_kpl);
  b.left_desc  = d + off_dl;
  b.n_left     = d_meta + 0;
  b.right_kp   = reinterpret_cast<prs_kp2*>(d + off_kpr);
  b.right_desc = d + off_dr;
  b.n_right    = d_meta + 1;
  b.matches    = reinterpret_cast<prs_corr*>(d);
  b.n_matches  = d_meta + 2;
  b.status     = d_meta + 3;
  const int rc = stereo_match_batch_launch(ctx, params, &b);
  if (rc != PRS_OK) {
    return rc;
  }
  // (a frame has at most n_left matches: the copy covers them and the meta words behind the match array)
  const size_t lo = 0;
  PRS_TRY(hipMemcpyAsync(h + lo, d + lo, off_meta + 16 - lo, hipMemcpyDeviceToHost, s));
  PRS_TRY(hipStreamSynchronize(s));
#undef PRS_TRY
  if (h_meta[3] < 0) {
    return ctx_fail(ctx, h_meta[3], "prs_stereo_match: keypoint outside the supported domain (0<=u<32768, 0<=v<image_rows)");
  }
  if (h_meta[2] > 0) {
    memcpy(out, h, sizeof(prs_corr) * (size_t) h_meta[2]);
  }
  *n_out = h_meta[2];
  return h_meta[3];
}

int prs_align_batch_run(prs_context* ctx, const prs_pcf_params* finder, const prs_aligner_params* aligner, const prs_align_batch* batch, int32_t mode) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  int rc = align_batch_launch(ctx, finder, aligner, batch, mode, 0);
  if (rc != PRS_OK) {
    return rc;
  }
  if (align_job_active(ctx)) {
    return align_batch_finish(ctx);
  }
  // (the one-launch paths -- finder / linearise modes, the fused kernel -- have nothing enqueued to finish)
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  return e == hipSuccess ? PRS_OK : ctx_fail_hip(ctx, e, "prs_align_batch_run");
}

int prs_align_batch_enqueue(prs_context* ctx, const prs_pcf_params* finder, const prs_aligner_params* aligner, const prs_align_batch* batch, int32_t rounds) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  // (an explicit round count: the split pipeline, which is what a caller that enqueues -- and may capture -- asked for)
  return align_batch_launch(ctx, finder, aligner, batch, PRS_MODE_ALIGN, rounds > 0 ? rounds : 5);
}

int prs_align_batch_finish(prs_context* ctx) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return align_batch_finish(ctx);
}

int prs_align_batch_rearm(prs_context* ctx) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  return align_batch_rearm(ctx, nullptr);
}

int prs_align_batch_rearm_on(prs_context* ctx, void* hip_stream) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  return align_batch_rearm(ctx, static_cast<hipStream_t>(hip_stream));
}

int prs_triangulate_dev(prs_context* ctx, const prs_triangulator_params* params, const float* d_uvuv, int64_t n, float* d_xyz4) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return triangulate_launch(ctx, params, d_uvuv, n, d_xyz4);
}

int prs_triangulate(prs_context* ctx, const prs_triangulator_params* params, const float* uvuv, int32_t n, float* xyz, uint8_t* valid) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  // triangulator_rigid_stereo.cpp:9-16: unset buffers are reported, nothing is computed
  if (!params || n < 0 || (n > 0 && (!uvuv || !xyz || !valid))) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_triangulate: input or result buffer not set");
  }
  if (n == 0) {
    return PRS_WARN_EMPTY_INPUT;
  }
  (void) hipSetDevice(ctx->device);
  const size_t bytes = sizeof(float) * 4 * (size_t) n;
  unsigned char* d   = static_cast<unsigned char*>(ctx_device_scratch(ctx, 2 * align256(bytes)));
  float* h_in        = static_cast<float*>(ctx_pinned_scratch(ctx, 2 * align256(bytes)));  // [measurements | points]: pinned both ways
  if (!d || !h_in) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_triangulate: scratch allocation failed");
  }
  float* h      = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(h_in) + align256(bytes));
  float* d_in   = reinterpret_cast<float*>(d);
  float* d_out  = reinterpret_cast<float*>(d + align256(bytes));
  hipStream_t s = ctx->stream;
  memcpy(h_in, uvuv, bytes);
  hipError_t e  = hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_triangulate upload");
  }
  const int rc = triangulate_launch(ctx, params, d_in, n, d_out);
  if (rc != PRS_OK) {
    return rc;
  }
  e = hipMemcpyAsync(h, d_out, bytes, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_triangulate download");
  }
  for (int32_t i = 0; i < n; ++i) {
    xyz[3 * i + 0] = h[4 * i + 0];
    xyz[3 * i + 1] = h[4 * i + 1];
    xyz[3 * i + 2] = h[4 * i + 2];
    valid[i]       = h[4 * i + 3] != 0.0f ? 1 : 0;
  }
  return PRS_OK;
}

int prs_scene_clip_batch(prs_context* ctx, const prs_projector* projector, const float* sensor_in_robot16, const prs_clip_batch* batch) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return scene_clip_launch(ctx, projector, sensor_in_robot16, batch);
}

int prs_scene_clip(prs_context* ctx,
                   const prs_projector* projector,
                   const float* robot_in_local_map16,
                   const float* sensor_in_robot16,
                   const float* scene_xyzw,
                   const uint8_t* scene_desc,
                   int32_t n,
                   float* clipped_xyzw,
                   uint8_t* clipped_desc,
                   int32_t* global_indices,
                   int32_t capacity,
                   int32_t* n_clipped) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  // scene_clipper_projective_3d.cpp:12-20: missing projector / clipped scene / global scene throw
  if (!projector) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: missing projector");
  }
  if (!clipped_xyzw || !global_indices || !n_clipped) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: missing clipped scene");
  }
  if (n < 0 || (n > 0 && !scene_xyzw) || !robot_in_local_map16 || !sensor_in_robot16) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: missing global scene");
  }
  if ((scene_desc == nullptr) != (clipped_desc == nullptr)) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: descriptor rows need both the scene and the clipped buffer");
  }
  if (n == 0) {
    return PRS_WARN_EMPTY_INPUT;  // :21-28, nothing is cleared
  }
  if (capacity < n) {
    return ctx_fail(ctx, PRS_ERR_CAPACITY, "prs_scene_clip: output capacity below the scene size");
  }
  (void) hipSetDevice(ctx->device);
  const size_t nn      = (size_t) n;
  const size_t b_xyzw  = align256(nn * 16);
  const size_t b_desc  = scene_desc ? align256(nn * PRS_DESC_BYTES) : 0;
  const size_t b_idx   = align256(nn * 4);
  const size_t b_small = 256;  // n_scene, pose, n_clipped, status
  const size_t total   = 2 * b_xyzw + 2 * b_desc + b_idx + b_small;
  unsigned char* d     = static_cast<unsigned char*>(ctx_device_scratch(ctx, total));
  unsigned char* h     = static_cast<unsigned char*>(ctx_pinned_scratch(ctx, total));
  if (!d || !h) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_scene_clip: scratch allocation failed");
  }
  // staging layout (same on both sides): [inputs: xyzw | desc | small] [outputs: xyzw | desc | idx]
  const size_t o_in_xyzw = 0, o_in_desc = b_xyzw, o_small = b_xyzw + b_desc;
  const size_t o_out_xyzw = o_small + b_small, o_out_desc = o_out_xyzw + b_xyzw, o_out_idx = o_out_desc + b_desc;
  memcpy(h + o_in_xyzw, scene_xyzw, nn * 16);
  if (scene_desc) {
    memcpy(h + o_in_desc, scene_desc, nn * PRS_DESC_BYTES);
  }
  int32_t* hs = reinterpret_cast<int32_t*>(h + o_small);
  hs[0]       = n;   // n_scene
  hs[1]       = 0;   // n_clipped
  hs[2]       = 0;   // status
  memcpy(hs + 4, robot_in_local_map16, 64);
  hipStream_t s = ctx->stream;
  hipError_t e  = hipMemcpyAsync(d, h, o_out_xyzw, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_scene_clip upload");
  }
  prs_clip_batch b;
  b.batch              = 1;
  b.stride             = n;
  b.scene_xyzw         = reinterpret_cast<const float*>(d + o_in_xyzw);
  b.scene_desc         = scene_desc ? d + o_in_desc : nullptr;
  b.n_scene            = reinterpret_cast<const int32_t*>(d + o_small);
  b.robot_in_local_map = reinterpret_cast<const float*>(d + o_small + 16);
  b.clipped_xyzw       = reinterpret_cast<float*>(d + o_out_xyzw);
  b.clipped_desc       = scene_desc ? d + o_out_desc : nullptr;
  b.global_indices     = reinterpret_cast<int32_t*>(d + o_out_idx);
  b.n_clipped          = reinterpret_cast<int32_t*>(d + o_small + 4);
  b.status             = reinterpret_cast<int32_t*>(d + o_small + 8);
  b.scene_n_opt        = nullptr;
  const int rc = scene_clip_launch(ctx, projector, sensor_in_robot16, &b);
  if (rc != PRS_OK) {
    return rc;
  }
  e = hipMemcpyAsync(h + o_small, d + o_small, total - o_small, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_scene_clip download");
  }
  const int32_t m = hs[1];
  memcpy(clipped_xyzw, h + o_out_xyzw, (size_t) m * 16);
  if (clipped_desc) {
    memcpy(clipped_desc, h + o_out_desc, (size_t) m * PRS_DESC_BYTES);
  }
  memcpy(global_indices, h + o_out_idx, (size_t) m * 4);
  *n_clipped = m;
  return hs[2];
}

int prs_bruteforce_match_batch(prs_context* ctx, const prs_bruteforce_params* params, const prs_bruteforce_batch* batch) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return bruteforce_batch_launch(ctx, params, batch);
}

int prs_bruteforce_match(prs_context* ctx,
                         const prs_bruteforce_params* params,
                         const uint8_t* fixed_desc,
                         int32_t n_fixed,
                         const uint8_t* moving_desc,
                         int32_t n_moving,
                         prs_corr* correspondences,
                         int32_t capacity,
                         int32_t* n_correspondences) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  // bruteforce_impl.cpp:203-216: unset buffers throw
  if (!params || n_fixed < 0 || (n_fixed > 0 && !fixed_desc)) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_bruteforce_match: fixed not set");
  }
  if (n_moving < 0 || (n_moving > 0 && !moving_desc)) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_bruteforce_match: moving not set");
  }
  if (!correspondences || !n_correspondences) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_bruteforce_match: correspondences not set");
  }
  *n_correspondences = 0;
  if (n_fixed == 0 || n_moving == 0) {
    return PRS_WARN_EMPTY_INPUT | PRS_WARN_NO_MATCHES;  // :217-226, :237-242
  }
  const int32_t n_min = n_fixed < n_moving ? n_fixed : n_moving;
  if (capacity < n_min) {
    return ctx_fail(ctx, PRS_ERR_CAPACITY, "prs_bruteforce_match: output capacity below min(n_fixed, n_moving)");
  }
  (void) hipSetDevice(ctx->device);
  const size_t b_f   = align256((size_t) n_fixed * PRS_DESC_BYTES);
  const size_t b_m   = align256((size_t) n_moving * PRS_DESC_BYTES);
  const size_t b_s   = 256;  // n_fixed, n_moving, n_matches, status
  const size_t b_out = align256((size_t) n_min * sizeof(prs_corr));
  const size_t total = b_f + b_m + b_s + b_out;
  // slot 3: the launch itself uses slots 0..2
  unsigned char* d   = static_cast<unsigned char*>(ctx_device_scratch_slot(ctx, 3, total));
  unsigned char* h   = static_cast<unsigned char*>(ctx_pinned_scratch(ctx, total));
  if (!d || !h) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_bruteforce_match: scratch allocation failed");
  }
  memcpy(h, fixed_desc, (size_t) n_fixed * PRS_DESC_BYTES);
  memcpy(h + b_f, moving_desc, (size_t) n_moving * PRS_DESC_BYTES);
  int32_t* hs = reinterpret_cast<int32_t*>(h + b_f + b_m);
  hs[0] = n_fixed;
  hs[1] = n_moving;
  hs[2] = 0;
  hs[3] = 0;
  hipStream_t s = ctx->stream;
  hipError_t e  = hipMemcpyAsync(d, h, b_f + b_m + b_s, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_bruteforce_match upload");
  }
  prs_bruteforce_batch b;
  b.batch              = 1;
  b.fixed_stride       = n_fixed;
  b.moving_stride      = n_moving;
  b.fixed_desc         = d;
  b.n_fixed            = reinterpret_cast<const int32_t*>(d + b_f + b_m);
  b.moving_desc        = d + b_f;
  b.n_moving           = reinterpret_cast<const int32_t*>(d + b_f + b_m + 4);
  b.matches            = reinterpret_cast<prs_corr*>(d + b_f + b_m + b_s);
  b.n_matches          = reinterpret_cast<int32_t*>(d + b_f + b_m + 8);
  b.status             = reinterpret_cast<int32_t*>(d + b_f + b_m + 12);
  // the candidate list defaults to 16 entries per descriptor; a loose threshold on correlated descriptors can need more
  // (at most every pair): grow and repeat, like the reference's std::vector would
  const long long all_pairs = (long long) n_fixed * (long long) n_moving;
  long long cand_cap        = 16ll * (long long) (n_fixed > n_moving ? n_fixed : n_moving);
  for (;;) {
    if (cand_cap > all_pairs) {
      cand_cap = all_pairs;
    }
    b.candidate_capacity = (int32_t) (cand_cap > 0x7fffffffll ? 0x7fffffffll : cand_cap);
    const int rc         = bruteforce_batch_launch(ctx, params, &b);
    if (rc != PRS_OK) {
      return rc;
    }
    e = hipMemcpyAsync(h + b_f + b_m, d + b_f + b_m, b_s + b_out, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) {
      e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) {
      return ctx_fail_hip(ctx, e, "prs_bruteforce_match download");
    }
    if (hs[3] != PRS_ERR_CAPACITY || cand_cap >= all_pairs) {
      break;
    }
    cand_cap *= 8;
    hs[2] = 0;
    hs[3] = 0;
    e     = hipMemcpyAsync(d + b_f + b_m, h + b_f + b_m, b_s, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
      return ctx_fail_hip(ctx, e, "prs_bruteforce_match upload");
    }
  }
  if (hs[3] < 0) {
    return ctx_fail(ctx, hs[3], "prs_bruteforce_match: more candidates below the threshold than the kernel's candidate capacity");
  }
  memcpy(correspondences, h + b_f + b_m + b_s, (size_t) hs[2] * sizeof(prs_corr));
  *n_correspondences = hs[2];
  return hs[3];
}

int prs_merge_batch_run(prs_context* ctx, const prs_merger_params* params, const prs_merge_batch* batch) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return merge_batch_launch(ctx, params, batch);
}

int prs_pose_compose_batch(prs_context* ctx, int32_t batch, const float* prediction, const float* X, float* pose_out) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return pose_compose_launch(ctx, batch, prediction, X, pose_out);
}

int prs_motion_predict_batch(prs_context* ctx, int32_t batch, const float* pose_prev2, const float* pose_prev1, float* pose_pred) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return motion_predict_launch(ctx, batch, pose_prev2, pose_prev1, pose_pred);
}

int prs_extract_features_batch(prs_context* ctx, const prs_extractor_params* params, const prs_extract_batch* batch) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  (void) hipSetDevice(ctx->device);
  return extract_features_launch(ctx, params, batch);
}

int prs_selection_order(prs_context* ctx, const uint8_t* response, int32_t n, int32_t* order) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  if (n < 0 || n > 32768) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_selection_order: more than 32768 keypoints in a region");
  }
  if (n == 0) {
    return PRS_OK;
  }
  if (!response || !order) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_selection_order: responses / order not set");
  }
  for (int32_t i = 0; i < n; ++i) {
    if (response[i] == 0) {
      return ctx_fail(ctx, PRS_ERR_RANGE, "prs_selection_order: a response of 0 (a detected corner scores at least 1)");
    }
  }
  (void) hipSetDevice(ctx->device);
  const size_t b_resp = align256((size_t) n), b_order = align256((size_t) n * 4);
  const size_t total  = b_resp + b_order + 256;
  unsigned char* d    = static_cast<unsigned char*>(ctx_device_scratch(ctx, total));
  unsigned char* h    = static_cast<unsigned char*>(ctx_pinned_scratch(ctx, total));
  if (!d || !h) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_selection_order: scratch allocation failed");
  }
  memcpy(h, response, (size_t) n);
  hipStream_t s = ctx->stream;
  hipError_t e  = hipMemcpyAsync(d, h, (size_t) n, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_selection_order upload");
  }
  const int rc = selection_order_launch(ctx, d, n, reinterpret_cast<int32_t*>(d + b_resp), reinterpret_cast<int32_t*>(d + b_resp + b_order));
  if (rc != PRS_OK) {
    return rc;
  }
  e = hipMemcpyAsync(h + b_resp, d + b_resp, b_order + 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_selection_order download");
  }
  const int32_t status = *reinterpret_cast<const int32_t*>(h + b_resp + b_order);
  if (status != PRS_OK) {
    return ctx_fail(ctx, status, "prs_selection_order: the sort did not finish");
  }
  memcpy(order, h + b_resp, (size_t) n * 4);
  return PRS_OK;
}

int prs_extract_features(prs_context* ctx,
                         const prs_extractor_params* params,
                         const uint8_t* image,
                         int32_t rows,
                         int32_t cols,
                         int32_t pitch,
                         float* keypoints,
                         float* intensity,
                         uint8_t* descriptors,
                         int32_t capacity,
                         int32_t* n_features) {
  if (!ctx) {
    return PRS_ERR_NULL;
  }
  if (!params || !image) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_extract_features: image not set");
  }
  if (!keypoints || !descriptors || !n_features) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_extract_features: target feature buffer not set");
  }
  if (rows <= 0 || cols <= 0 || pitch < cols || capacity <= 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_extract_features: invalid image size, pitch or capacity");
  }
  (void) hipSetDevice(ctx->device);
  const size_t cap    = (size_t) capacity;
  const size_t b_img  = align256((size_t) rows * (size_t) pitch);
  const size_t b_kp   = align256(cap * sizeof(prs_kp2));
  const size_t b_int  = align256(cap * sizeof(float));
  const size_t b_desc = align256(cap * PRS_DESC_BYTES);
  const size_t b_small = 256;
  const size_t total   = b_img + b_small + b_kp + b_int + b_desc;
  unsigned char* d     = static_cast<unsigned char*>(ctx_device_scratch(ctx, total));
  unsigned char* h     = static_cast<unsigned char*>(ctx_pinned_scratch(ctx, total));
  if (!d || !h) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_extract_features: scratch allocation failed");
  }
  // staging layout (same on both sides): image | n_features, status | keypoints | intensity | descriptors
  const size_t o_small = b_img, o_kp = o_small + b_small, o_int = o_kp + b_kp, o_desc = o_int + b_int;
  // (the last row of a pitched view, e.g. a cv::Mat ROI, owns only `cols` bytes)
  const size_t image_bytes = (size_t) (rows - 1) * (size_t) pitch + (size_t) cols;
  memcpy(h, image, image_bytes);
  hipStream_t s = ctx->stream;
  hipError_t e  = hipMemcpyAsync(d, h, image_bytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_extract_features upload");
  }
  prs_extract_batch b;
  b.batch       = 1;
  b.rows        = rows;
  b.cols        = cols;
  b.pitch       = pitch;
  b.images      = d;
  b.stride      = capacity;
  b.keypoints   = reinterpret_cast<prs_kp2*>(d + o_kp);
  b.intensity   = reinterpret_cast<float*>(d + o_int);
  b.descriptors = d + o_desc;
  b.n_features  = reinterpret_cast<int32_t*>(d + o_small);
  b.status      = reinterpret_cast<int32_t*>(d + o_small + 4);
  const int rc  = extract_features_launch(ctx, params, &b);
  if (rc != PRS_OK) {
    return rc;
  }
  e = hipMemcpyAsync(h + o_small, d + o_small, total - o_small, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) {
    e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_extract_features download");
  }
  const int32_t* hs = reinterpret_cast<const int32_t*>(h + o_small);
  const int32_t n = hs[0], status = hs[1];
  if (status < 0) {
    *n_features = 0;
    return ctx_fail(ctx, status, "prs_extract_features: more raw detections or features than the buffers hold");
  }
  memcpy(keypoints, h + o_kp, (size_t) n * sizeof(prs_kp2));
  if (intensity) {
    memcpy(intensity, h + o_int, (size_t) n * sizeof(float));
  }
  memcpy(descriptors, h + o_desc, (size_t) n * PRS_DESC_BYTES);
  *n_features = n;
  return status;
}

}  // extern "C"

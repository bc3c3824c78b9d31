// prs_host.h -- host-side context shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "../../include/proslam_hip.h"

struct prs_context {
  int device            = 0;
  hipStream_t stream    = nullptr;  // stream work is enqueued on
  hipStream_t own       = nullptr;  // stream created (and destroyed) by the context
  std::string last_error;
  bool fused_align      = false;    // PRS_FUSED_ALIGN=1: one fused kernel per frame loop instead of the split search/GN pipeline
  bool matcher_v3       = false;    // PRS_MATCHER_V3=1: always use the first-generation matcher kernel
  bool force_unstaged   = false;    // test hook: PRS_FORCE_UNSTAGED=1 selects the no-LDS-staging variant
  bool merge_fused      = false;    // PRS_MERGE_FUSED=1: pose-based smoother merger as one kernel instead of front | smoother | back
  bool no_prefilter     = false;    // PRS_NO_PREFILTER=1: the search scan scores every candidate in full (diagnostic: what the irrelevance bound buys)
  int prefilter_96_limit = 32;      // PRS_PREFILTER_96_LIMIT: largest irrelevance bound the search scan tests on 96 instead of 128 bits (diagnostic / A-B)
  int bf_mfma           = 1;        // PRS_BF_DENSE_*: the brute-force matcher's dense phase (prs_context_set_bruteforce_dense_phase; PRS_BF_MFMA=0 / auto / 1)
  bool no_lone_gn       = false;    // PRS_NO_LONE_GN=1: small batches use the throughput instantiation of the Gauss-Newton kernel too (diagnostic)
  bool stamps_split     = false;    // PRS_STAMPS_SPLIT=1 (with PRS_STAMPS=1): phase stamps of the split search kernel
  // reusable device scratch for the host-pointer entry points
  void* d_scratch       = nullptr;
  size_t d_scratch_size = 0;
  void* d_slot[4]       = {nullptr, nullptr, nullptr, nullptr};  // kernel-owned scratch (candidates, ...)
  size_t d_slot_size[4] = {0, 0, 0, 0};
  float* d_info_lut     = nullptr;  // information scale by landmark age, 4096 entries (scene clipper)
  void* h_pinned        = nullptr;
  size_t h_pinned_size  = 0;
  // per-kernel HIP-event timing of the split aligner pipeline (prs_context_enable_timing; measurement only)
  bool timing           = false;
  double t_search_ms = 0.0, t_gn_ms = 0.0;
  long long n_search = 0, n_gn = 0;
  double t_search_round[16] = {}, t_gn_round[16] = {};  // the same, by round of the batch (round 15 collects the rest)
  long long n_batches_timed = 0;
  hipEvent_t timing_ev[3 * 64] = {};
  // the batch of the split aligner pipeline that has been enqueued and not yet finished (owned by align.hip)
  void* align_job               = nullptr;
  void (*align_job_free)(void*) = nullptr;
  // diagnostic phase stamps (PRS_STAMPS=1): never enabled in timed runs
  bool stamps_enabled   = false;
  unsigned long long* d_stamps = nullptr;
  size_t d_stamps_size  = 0;
};

namespace prs {

int ctx_fail(prs_context* ctx, int status, const char* what);
int ctx_fail_hip(prs_context* ctx, hipError_t e, const char* what);
inline hipStream_t ctx_stream(prs_context* ctx) {
  return ctx->stream;
}
inline bool ctx_fused_align(const prs_context* ctx) {
  return ctx->fused_align;
}
inline bool ctx_matcher_v3(const prs_context* ctx) {
  return ctx->matcher_v3;
}
inline bool ctx_force_unstaged(const prs_context* ctx) {
  return ctx->force_unstaged;
}
// grows (never shrinks) the context's device scratch; returns nullptr on failure
void* ctx_device_scratch(prs_context* ctx, size_t bytes);
void* ctx_pinned_scratch(prs_context* ctx, size_t bytes);
void* ctx_device_scratch_slot(prs_context* ctx, int slot, size_t bytes);
// device table scale[n] = n > 2 ? 1 + log(n) : 1 for n < 4096 (built once per context)
const float* ctx_info_scale_table(prs_context* ctx);
// diagnostic: device buffer for phase stamps when PRS_STAMPS=1, else nullptr
unsigned long long* ctx_stamps(prs_context* ctx, size_t bytes);
// synchronises and prints mean per-phase cycle counts (n_stamps consecutive stamps per block)
void ctx_report_stamps(prs_context* ctx, int blocks, int n_stamps, const char* legend, bool raw = false, size_t first_block = 0);  // raw: the words are per-phase totals, not cumulative stamps

// exact integer form of `best < max_distance && best / second < max_ratio` (epipolar_impl.cpp:171-173),
// evaluated on the host with the same IEEE float operations the reference performs:
// accept iff best < *best_lim && best <= bmax[second] (index 257 = no second candidate)
void fill_accept_table(const prs_stereo_params* params, int* best_lim, int16_t* bmax258);
int stereo_match_v5_launch(prs_context* ctx, const prs_stereo_params* params, const prs_stereo_batch* batch);
int stereo_match_batch_launch(prs_context* ctx, const prs_stereo_params* params, const prs_stereo_batch* batch);
int align_batch_launch(prs_context* ctx, const prs_pcf_params* finder, const prs_aligner_params* aligner, const prs_align_batch* batch, int mode, int rounds);
int align_batch_finish(prs_context* ctx);
int align_batch_rearm(prs_context* ctx, hipStream_t replay_stream);
bool align_job_active(const prs_context* ctx);
int gn_step_launch(prs_context* ctx, const float* dH, const float* db, float damping, int damping_form, float* dX, int* dok);
int recip_selftest_launch(prs_context* ctx, unsigned long long* d_counts);
int bruteforce_batch_launch(prs_context* ctx, const prs_bruteforce_params* params, const prs_bruteforce_batch* batch);
int selection_order_launch(prs_context* ctx, const uint8_t* response_dev, int n, int32_t* order_dev, int32_t* status_dev);
int extract_features_launch(prs_context* ctx, const prs_extractor_params* params, const prs_extract_batch* batch);
int pose_compose_launch(prs_context* ctx, int batch, const float* prediction, const float* X, float* pose_out);
int motion_predict_launch(prs_context* ctx, int batch, const float* prev2, const float* prev1, float* pred);
int merge_batch_launch(prs_context* ctx, const prs_merger_params* params, const prs_merge_batch* batch);
int scene_clip_launch(prs_context* ctx, const prs_projector* projector, const float* sensor_in_robot, const prs_clip_batch* batch);
int triangulate_launch(prs_context* ctx, const prs_triangulator_params* params, const float* d_uvuv, int64_t n, float* d_xyz4);

} // namespace prs

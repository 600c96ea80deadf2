// capi_detect.cpp -- detector / extractor pipeline behind the C ABI: score map + NMS, sort,
// uniformity selection, sub-pixel, descriptors, compaction + FP64 back-projection; the batch entry
// points (device- and host-fed) and the single-image host-buffer calls that stand behind
// cv::FeatureDetector::detect / cv::DescriptorExtractor::compute (Frame.hpp:152,167) and
// Frontend::detectAndDescribe (Frontend.cpp:221-269).
#include "okvfe_ctx.h"

using namespace okvfe;

extern "C" {

okvfe_status okvfe_harris_score_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                       int32_t* scores_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || !scores_dev || n_images < 0)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_harris_score_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  {
    hipStream_t s = pick_stream(ctx, stream);
    StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
    if (ctx->cfg.score_type != OKVFE_SCORE_HARRIS)
      launch_agast_score(images_dev, ctx->w, ctx->h, n_images, scores_dev, s);
    else
      launch_harris(images_dev, ctx->w, ctx->h, n_images, scores_dev, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  return OKVFE_OK;
}

okvfe_status okvfe_harris_byte_mover_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                            void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 0 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_harris_byte_mover_device: bad argument");
  if (ctx->cfg.score_type != OKVFE_SCORE_HARRIS || !ctx->d_scores || ctx->score_layout.strips < 1)
    return fail(ctx, OKVFE_ERR_UNSUPPORTED, "okvfe_harris_byte_mover_device: the fused score kernel does not apply to this context");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  bool ok;
  {
    StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
    ok = launch_harris_byte_mover(images_dev, ctx->w, ctx->h, n_images, ctx->d_scores, ctx->score_layout, s);
  }
  if (!ok) return fail(ctx, OKVFE_ERR_UNSUPPORTED, "okvfe_harris_byte_mover_device: image base or width not dword aligned");
  ctx->map_free_live = false;  // the buffer holds (pixel bytes in) the fused kernel's layout
  ctx->live_layout = ctx->score_layout;
  HIP_TRY(ctx, hipGetLastError());
  return OKVFE_OK;
}

static okvfe_status upload_image_params(okvfe_ctx* ctx, int n_images, const int32_t* cam_ids,
                                        const float* gravity, hipStream_t s, bool before_detect = false) {
  std::vector<ImageParams> prm(n_images);
  ctx->wide_patches = false;
  ctx->aware_fast = true;
  ctx->all_aware = n_images > 0;
  ctx->none_aware = true;
  for (int i = 0; i < n_images; ++i) {
    ImageParams& p = prm[i];
    p.cam = cam_ids ? cam_ids[i] : -1;
    if (p.cam >= ctx->cfg.num_cameras)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "camera id %d out of range", p.cam);
    const bool aware = gravity != nullptr && p.cam >= 0;
    if (aware) {
      if (!ctx->cam_rays[p.cam])
        return fail(ctx, OKVFE_ERR_NOT_READY,
                    "camera-aware extraction requested for camera %d before okvfe_set_camera[_maps]", p.cam);
      p.mode = kCameraAware;
      ctx->none_aware = false;
      p.dir[0] = gravity[3 * i];
      p.dir[1] = gravity[3 * i + 1];
      p.dir[2] = gravity[3 * i + 2];
      p.fu = ctx->cam_fu[p.cam];
      if (ctx->cam_wide[p.cam]) ctx->wide_patches = true;
      if (ctx->cam_aware_slow[p.cam]) ctx->aware_fast = false;
    } else {
      p.mode = ctx->mode_default;
      ctx->all_aware = false;
      p.dir[0] = 0.0f; p.dir[1] = 1.0f; p.dir[2] = 0.0f;
      p.fu = 1.0f;
    }
    if (p.cam >= 0 && !ctx->cam_has_intrinsics[p.cam]) p.cam = aware ? p.cam : -1;
  }
  // intrinsics are needed for back-projection; a slot with maps only (set_camera_maps) keeps its
  // cam id for the maps and gets invalid back-projections (DeviceCamera zeroed -> fu = 0)
  void* d = nullptr;
  // (single-scale detector: its candidate / fix-up counters are cleared by the same launch)
  ctx->counters_cleared = false;
  okvfe_status st = ring_upload(ctx, &ctx->prm_ring, prm.data(), n_images * sizeof(ImageParams), s, &d,
                                &ctx->prm_slot, before_detect && ctx->n_layers == 1 ? ctx->d_cand_count : nullptr,
                                before_detect && ctx->n_layers == 1 && ctx->d_cand_count ? 2 * ctx->B : 0,
                                &ctx->counters_cleared);
  if (st != OKVFE_OK) return st;
  ctx->d_prm = static_cast<ImageParams*>(d);
  return OKVFE_OK;
}

// ---- batch pipeline ----------------------------------------------------------------------------
// OKVFE_SCORE_TOKEN: see g_score_token.
// The camera-aware-only descriptor kernel carries the fixed-trip box sum alone: every box of the installed pattern has
// to fit its 11 x 11 slots (sigma_half <= 4.75: the built-in pattern; okvfe_set_pattern may install wider samples,
// which take the all-modes kernel).
// 0: every box of the installed pattern fits the fixed-trip slots of the fast descriptor kernels (first-pass samples
// 11 x 11: sigma_half <= 4.75; second-pass samples = points 0 .. n - 65 of a pattern with more than 64 points, 5 x 5:
// sigma_half <= 2.0); 1: the slots of the WIDE instantiations (21 x 21: <= 9.75, 10 x 10: <= 4.25); 2: wider still
// (k_describe.hip: kMaxBox / kSmallBox / kWideBox / kWideSmallBox)
static int pattern_box_class(const okvfe::Pattern& P) {
  const int extra = P.n_points > 64 ? P.n_points - 64 : 0;
  int cls = 0;
  for (int i = 0; i < P.n_points && i < okvfe::kPatternPoints; ++i) {
    const float s = P.sigma_half[i];
    if (!(s <= (i < extra ? 2.0f : 4.75f))) cls = cls < 1 ? 1 : cls;
    if (!(s <= (i < extra ? 4.25f : 9.75f))) cls = 2;
    // half-widths below 0.5 are bilinear point samples: only the all-modes kernel carries that branch (and waits for
    // its patch before it: ADVICE r5)
    if (!(s >= 0.5f)) cls = 2;
  }
  return cls;
}

// describe_rot_kernel (upright / gradient modes on the fast box sums) takes the installed pattern when its circle fits a
// 64-byte row pitch, its samples beyond 64 lanes fit the kernel's table and its long-pair weights fit 16 bits
static bool pattern_rot_ok(const okvfe::Pattern& P) {
  const int extra = P.n_points > 64 ? P.n_points - 64 : 0;
  if (P.border > 29 || extra > okvfe::kAwareMaxExtra || P.n_long > okvfe::kMaxLongPairs) return false;
  for (int l = 0; l < P.n_long; ++l)
    if (P.long_wdx[l] < -32768 || P.long_wdx[l] > 32767 || P.long_wdy[l] < -32768 || P.long_wdy[l] > 32767) return false;
  // the rotation tables must follow the quarter-wave rule exactly (they do for the tables build_pattern computes)
  static const bool sym = [&P] {
    for (int k = 0; k < okvfe::kRot; ++k) {
      if (okvfe::quarter_sin(P.rot_sin, k) != P.rot_sin[k] || okvfe::quarter_cos(P.rot_sin, k) != P.rot_cos[k]) return false;
      // (float tables: sin(pi) and cos(pi / 2) are 1e-16, not 0, in double -- entries sin[512], cos[256], cos[768] are
      // read from the global table by the kernel and exempt here)
      const float fs = okvfe::quarter_sin(P.rot_sinf, k), fc = okvfe::quarter_cos(P.rot_sinf, k);
      if (k != 512 && std::memcmp(&fs, &P.rot_sinf[k], 4) != 0) return false;
      if (k != 256 && k != 768 && std::memcmp(&fc, &P.rot_cosf[k], 4) != 0) return false;
    }
    return true;
  }();
  return sym;
}

extern "C" int32_t okvfe_pattern_kernel_class(const okvfe_ctx* ctx) { return ctx ? pattern_box_class(ctx->host_pattern) : -1; }

// Does describe_aware_kernel (k_describe_aware.hip) serve this call?  -1: no (describe_kernel does); otherwise
// (samples beyond 64) << 8 | their largest box side minus one (0: the pattern has no such samples).
// Every image camera-aware on a camera whose patches fit the kernel's LDS classes, the fixed-scale pattern with boxes
// inside the fixed-trip slots, dword-aligned images.
static int aware_box_for_call(const okvfe_ctx* ctx, const uint8_t* images_dev) {
  static const bool old_aware = lab_env("OKVFE_DESC_R5") != nullptr;  // A/B knob: the round-5 kernels
  const okvfe::Pattern& P = ctx->host_pattern;
  const int cls = pattern_box_class(P);
  const int extra = P.n_points > 64 ? P.n_points - 64 : 0;
  if (old_aware || !ctx->all_aware || !ctx->aware_fast || cls > 1 || extra > okvfe::kAwareMaxExtra || ctx->d_scales ||
      ctx->n_layers != 1 || ctx->w % 4 != 0 || (reinterpret_cast<uintptr_t>(images_dev) & 3) != 0 || ctx->w >= 4096 ||
      ctx->h >= 4096)
    return -1;
  return extra == 0 ? 0 : ((extra << 8) | (cls == 0 ? 4 : 9));  // (samples beyond 64) << 8 | box side - 1
}

namespace {
// The token mutex is held from the wait on the previous holder's event to the record of this
// launch's event, so two host threads can never chain on the same predecessor.
struct TokenScope {
  std::unique_lock<std::mutex> lock;
  bool on = false;
};
okvfe_status heavy_begin(okvfe_ctx* ctx, hipStream_t s, int which, TokenScope* t) {
  t->on = score_token_mode() > which && ctx->cfg.device >= 0 && ctx->cfg.device < kMaxTokenDevices;
  if (!t->on) return OKVFE_OK;
  t->lock = std::unique_lock<std::mutex>(g_token_mutex);
  hipEvent_t prev = g_score_token[ctx->cfg.device];
  if (prev) HIP_TRY(ctx, hipStreamWaitEvent(s, prev, 0));
  return OKVFE_OK;
}
okvfe_status heavy_end(okvfe_ctx* ctx, hipStream_t s, int which, TokenScope* t) {
  if (!t->on) return OKVFE_OK;
  hipEvent_t& ev = ctx->heavy_done[which];
  if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(ev, s));
  g_score_token[ctx->cfg.device] = ev;
  t->lock.unlock();
  return OKVFE_OK;
}

// K1 + K2 of one layer context `L` (score map + NMS candidates), launched for `owner`
void layer_score_nms(okvfe_ctx* L, const uint8_t* images_dev, int n_images, hipStream_t s, bool* fused) {
  int32_t* d_fix_count = L->d_fix_count;
  if (L->cfg.score_type != OKVFE_SCORE_HARRIS) {  // AGAST score map only; the stand-alone NMS follows
    launch_agast_score(images_dev, L->w, L->h, n_images, L->d_scores, s);
    *fused = false;
    L->live_layout = L->score_layout;
    return;
  }
  // map-free: a single-scale context whose selection kernel recomputes the sub-pixel scores from the
  // image needs no map at all -- the fused kernel then writes candidates only (1 of 5 bytes per pixel)
  static const bool lab_keep = lab_env("OKVFE_KEEP_SCORE_MAP") != nullptr;  // A/B knob
  const bool map_free = !L->keep_score_map && !lab_keep && L->n_layers == 1 && !L->layer_child &&
                        select_recomputes_scores(L->cfg.uniformity_radius, L->cfg.max_keypoints, L->kp_cap, L->d_occ,
                                                 L->occ_image_bytes, L->occ_rows, L->occ_cols);
  // a slotted score layout exists only where the fused kernel applies (decided at creation)
  *fused = L->score_layout.strips >= 1 &&
           launch_harris_nms(images_dev, L->w, L->h, n_images, L->d_scores, L->score_layout,
                             L->cfg.absolute_threshold, L->d_cand, L->cand_cap, L->d_cand_count, d_fix_count,
                             L->d_fix_list, s, !map_free);
  L->map_free_live = *fused && map_free;
  L->live_images = images_dev;
  if (!*fused) launch_harris(images_dev, L->w, L->h, n_images, L->d_scores, s);
  // the unfused pair writes and reads a dense map (pitch w) into the same buffer: every later
  // reader of this call (selection, scale filter, okvfe_get_device_outputs) must follow it
  L->live_layout = *fused ? L->score_layout : ScoreLayout{L->w, 0};
}
void layer_nms_finish(okvfe_ctx* L, int n_images, hipStream_t s, bool fused) {
  int32_t* d_fix_count = L->d_fix_count;
  if (fused)
    launch_nms_fixup(L->d_scores, L->live_layout, L->w, L->h, n_images, L->cfg.absolute_threshold, L->d_cand, L->cand_cap,
                     L->d_cand_count, d_fix_count, L->d_fix_list, s, L->map_free_live, L->d_scores);
  else
    launch_nms(L->d_scores, L->w, L->h, n_images, L->cfg.absolute_threshold, L->d_cand, L->cand_cap,
               L->d_cand_count, s);
}
void layer_sort(okvfe_ctx* L, int n_images, hipStream_t s) {
  // the array-bin selection kernel buckets and orders its candidates itself (k_select.hip, round 4)
  if (select_sorts_candidates(L->cfg.uniformity_radius, L->cfg.max_keypoints, L->kp_cap, L->d_occ, L->occ_image_bytes,
                              L->occ_rows, L->occ_cols))
    return;
  launch_sort(L->d_cand, L->cand_cap, L->d_cand_count, n_images, L->cfg.uniformity_radius, L->d_sort_ws, s);
}
void layer_select(okvfe_ctx* L, int n_images, hipStream_t s) {
  // detection + description in one call (single scale): the selection kernel also prepares the
  // extractor's per-keypoint inputs (describe_setup_dev.h)
  if (!L->lane_view) L->aware_extra_box = aware_box_for_call(L, L->live_images);  // (a view: its owner decided)
  const DescribeSetup setup{L->d_pattern, L->d_prm, L->d_rays_ptrs, L->d_jac_ptrs, L->d_kps_tmp, L->d_desc_tmp,
                            L->d_valid_tmp, L->d_scales, L->live_images,
                            L->aware_extra_box > 0 && okvfe::aware_extras_in_setup() ? (L->aware_extra_box & 0xFF) : 0};
  const bool fuse = L->fuse_setup && L->n_layers == 1 && L->d_pattern && L->d_kps_tmp && L->d_prm;
  L->setup_done = launch_select(L->d_scores, L->live_layout, L->w, L->h, n_images, L->d_cand, L->cand_cap,
                                L->d_cand_count, L->cfg.uniformity_radius, L->cfg.max_keypoints, L->d_lut, L->d_occ,
                                L->occ_image_bytes, L->occ_rows, L->occ_cols, L->d_kps_det, L->kp_cap, L->d_det_count,
                                L->d_sort_ws, s, fuse ? &setup : nullptr, L->map_free_live ? L->live_images : nullptr);
}

static bool scale_space_serial() {
  static const bool serial = lab_env("OKVFE_SS_SERIAL") != nullptr;  // A/B knob: the layers one after the other on one stream
  return serial;
}

// scale of layer l relative to layer m, reduced (oracle: the layer ratios of detect_scale_space)
static void layer_ratio(int l, int m, int out[2]) {
  int sn, sd, mn, md;
  layer_scale(l, &sn, &sd);
  layer_scale(m, &mn, &md);
  int rn = sn * md, rd = sd * mn;
  for (int g = 2; g <= 3; ++g)
    while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
  out[0] = rn; out[1] = rd;
}

// The scale space of one call with its layers SIDE BY SIDE (round 6): the kernels of the small layers are a few hundred
// workgroups each -- one after the other on one stream they left most of the GPU idle (scale filter 4 x 0.1 ms,
// refinement 4 x 0.05, NMS of the upper layers ...).  Layer l runs on the stream of its own layer context:
//   sampler (behind the image it samples) -> score map -> NMS            | event "map l"
//   (behind the maps of l - 1 and l + 1) scale filter -> sort -> selection / refinement | event "done l"
// and the caller's stream picks up behind all "done" events for the merge.  Same kernels, same results.
static okvfe_status detect_layers_concurrent(okvfe_ctx* ctx, const uint8_t* images_dev, int n_images, hipStream_t s) {
  const int L = ctx->n_layers;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  if (!ctx->layer_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->layer_fork, hipEventDisableTiming));
  while ((int)ctx->layer_ev.size() < 3 * L + 1) {  // (+ 1: the virtual layer's map)
    hipEvent_t ev = nullptr;
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ctx->layer_ev.push_back(ev);
  }
  static const int own_mask = lab_env("OKVFE_SS_OWN") ? atoi(lab_env("OKVFE_SS_OWN")) : 0xFF;  // bisecting knob: layers on streams of their own
  auto ev_img = [&](int l) { return ctx->layer_ev[3 * l]; };
  auto ev_map = [&](int l) { return ctx->layer_ev[3 * l + 1]; };
  auto ev_done = [&](int l) { return ctx->layer_ev[3 * l + 2]; };
  std::vector<const uint8_t*> img(L);
  img[0] = images_dev;
  for (int l = 1; l < L; ++l) img[l] = ctx->d_layer_img[l];
  const bool brisk_ss = ctx->cfg.score_type == OKVFE_SCORE_BRISK_SCALESPACE;
  HIP_TRY(ctx, hipEventRecord(ctx->layer_fork, s));
  // ---- per layer: image, score map, 2-D maxima
  for (int l = 0; l < L; ++l) {
    okvfe_ctx* ch = ctx->layers[l];
    hipStream_t ls = ((own_mask >> l) & 1) ? ch->stream : s;
    HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->layer_fork, 0));
    if (l == 1) {
      launch_twothird(img[0], ctx->layer_w[0], ctx->layer_h[0], n_images, ctx->d_layer_img[1], ls);
    } else if (l >= 2) {
      if (l - 2 >= 1) HIP_TRY(ctx, hipStreamWaitEvent(ls, ev_img(l - 2), 0));  // (layer 0 is the caller's image)
      launch_halfsample(img[l - 2], ctx->layer_w[l - 2], ctx->layer_h[l - 2], n_images, ctx->d_layer_img[l], ls);
    }
    if (l >= 1) HIP_TRY(ctx, hipEventRecord(ev_img(l), ls));
    static const int dbg_l1 = lab_env("OKVFE_SS_DBG_L1") ? atoi(lab_env("OKVFE_SS_DBG_L1")) : 0;  // bisecting knob
    if (l == 1 && dbg_l1 == 1) (void)hipDeviceSynchronize();
    static const bool dbg_memset = lab_env("OKVFE_SS_DBG_MEMSET") != nullptr;  // bisecting knob: the runtime's memset
    if (dbg_memset)
      HIP_TRY(ctx, hipMemsetAsync(ch->d_cand_count, 0, 2 * (size_t)ch->B * sizeof(int32_t), ls));
    else  // counters cleared by a kernel of our own (stays in the layer's compute queue)
      launch_param_copy(nullptr, nullptr, 0, ch->d_cand_count, 2 * ch->B, ls, nullptr, 0);
    bool f;
    layer_score_nms(ch, img[l], n_images, ls, &f);
    if (l == 1 && dbg_l1 == 2) (void)hipDeviceSynchronize();
    layer_nms_finish(ch, n_images, ls, f);
    HIP_TRY(ctx, hipEventRecord(ev_map(l), ls));
    if (const char* m = lab_env("OKVFE_SS_DBG_A"))  // bisecting knob: device sync behind the layers of this bit mask
      if ((atoi(m) >> l) & 1) (void)hipDeviceSynchronize();
  }
  // the FAST 5-8 map of c0 (the virtual layer below the first octave) is as large as c0's own score map: it runs on the
  // LAST layer's stream, whose own chain is the shortest, instead of behind c0's score kernel on the longest one
  hipEvent_t ev_virtual = ctx->layer_ev[3 * L];
  if (ctx->d_virtual) {
    hipStream_t vs = ((own_mask >> (L - 1)) & 1) ? ctx->layers[L - 1]->stream : s;
    launch_fast58_score(img[0], ctx->layer_w[0], ctx->layer_h[0], n_images, ctx->d_virtual, vs);
    HIP_TRY(ctx, hipEventRecord(ev_virtual, vs));
  }
  static const char* dbg = lab_env("OKVFE_SS_DBG");  // bisecting knob: 1 = device sync between the phases, 2 = after every layer too
  if (dbg) (void)hipDeviceSynchronize();
  // ---- per layer: scale-space maxima against the finished maps below and above, order, selection / refinement
  for (int l = 0; l < L; ++l) {
    okvfe_ctx* ch = ctx->layers[l];
    hipStream_t ls = ((own_mask >> l) & 1) ? ch->stream : s;
    if (dbg && dbg[0] == '2') (void)hipDeviceSynchronize();
    if (l > 0) HIP_TRY(ctx, hipStreamWaitEvent(ls, ev_map(l - 1), 0));
    if (l + 1 < L) HIP_TRY(ctx, hipStreamWaitEvent(ls, ev_map(l + 1), 0));
    if (l == 0 && ctx->d_virtual) HIP_TRY(ctx, hipStreamWaitEvent(ls, ev_virtual, 0));
    const int32_t *below = nullptr, *above = nullptr;
    ScoreLayout lb{0, 0}, la{0, 0};
    int rb[2] = {1, 1}, ra[2] = {1, 1};
    if (l > 0) { below = ctx->layers[l - 1]->d_scores; lb = ctx->layers[l - 1]->live_layout; layer_ratio(l, l - 1, rb); }
    if (l == 0 && ctx->d_virtual) { below = ctx->d_virtual; lb = ScoreLayout{ctx->layer_w[0], 0}; }  // same grid: ratio 1
    if (l + 1 < L) { above = ctx->layers[l + 1]->d_scores; la = ctx->layers[l + 1]->live_layout; layer_ratio(l, l + 1, ra); }
    launch_scale_filter(ch->d_cand, ch->cand_cap, ch->d_cand_count, n_images, below, lb,
                        l > 0 ? ctx->layer_w[l - 1] : (below ? ctx->layer_w[0] : 0),
                        l > 0 ? ctx->layer_h[l - 1] : (below ? ctx->layer_h[0] : 0), rb[0], rb[1], above, la,
                        l + 1 < L ? ctx->layer_w[l + 1] : 0, l + 1 < L ? ctx->layer_h[l + 1] : 0, ra[0], ra[1], ls);
    if (brisk_ss) {
      launch_sort(ch->d_cand, ch->cand_cap, ch->d_cand_count, n_images, 1.0f, ch->d_sort_ws, ls);
      const int32_t* rbelow = l == 0 ? ctx->d_virtual : ctx->layers[l - 1]->d_scores;
      const int wb = l == 0 ? ctx->layer_w[0] : ctx->layer_w[l - 1], hb = l == 0 ? ctx->layer_h[0] : ctx->layer_h[l - 1];
      const int32_t* rabove = l + 1 < L ? ctx->layers[l + 1]->d_scores : nullptr;
      // c_0: the virtual FAST 5-8 layer sits at 2/3 and the result is clamped to [0.7, 1.5] (published refine1D_2)
      const double rel_b = (l & 1) || l == 0 ? 2.0 / 3.0 : 0.75, rel_a = (l & 1) ? 4.0 / 3.0 : 1.5;
      const double rel_lo = l == 0 ? 0.7 : rel_b;
      launch_brisk_refine(ch->d_scores, ch->w, ch->h, n_images, ch->cand_cap, ch->d_cand_count, ch->d_sort_ws,
                          ch->cfg.max_keypoints, rbelow, wb, hb, rb[0], rb[1], rabove,
                          rabove ? ctx->layer_w[l + 1] : 0, rabove ? ctx->layer_h[l + 1] : 0, ra[0], ra[1], rel_b, rel_a,
                          rel_lo, ch->d_kps_det, ch->kp_cap, ch->d_det_count, ls);
    } else {
      layer_sort(ch, n_images, ls);
      layer_select(ch, n_images, ls);
    }
    HIP_TRY(ctx, hipEventRecord(ev_done(l), ls));
  }
  // ---- join and merge into image coordinates
  for (int l = 0; l < L; ++l) HIP_TRY(ctx, hipStreamWaitEvent(s, ev_done(l), 0));
  const okvfe_keypoint* kps[8];
  const int32_t* counts[8];
  float scale[8];
  for (int l = 0; l < L; ++l) {
    kps[l] = ctx->layers[l]->d_kps_det;
    counts[l] = ctx->layers[l]->d_det_count;
    int sn, sd;
    layer_scale(l, &sn, &sd);
    scale[l] = (float)sn / (float)sd;
  }
  launch_merge_layers(kps, counts, scale, L, ctx->cfg.max_keypoints, n_images, ctx->d_kps_det, ctx->kp_cap,
                      ctx->d_det_count, s);
  return OKVFE_OK;
}

// K1..K4: score map + NMS, sort, uniformity selection, sub-pixel -> d_kps_det / d_det_count.
// octaves > 0: the same per layer of the scale space (k_pyramid.hip), with the cross-layer maximum
// test between NMS and selection and the merge into image coordinates at the end.
okvfe_status detect_stage(okvfe_ctx* ctx, const uint8_t* images_dev, int n_images, hipStream_t s) {
  TokenScope token;
  okvfe_status st;
  ctx->setup_done = false;  // set by this call's selection launch only (a failed earlier call must not leak it)
  if (ctx->n_layers == 1) {
    // priority lanes: the score kernel (and the clears in front of it) go to the owner's low-priority score stream,
    // which runs the slices' score kernels back to back; this lane's stream picks up behind its k1_done
    hipStream_t ks = ctx->lane_view && ctx->score_stream ? ctx->score_stream : s;
    if (!ctx->counters_cleared) {  // (cleared together with the parameter upload of the same call otherwise)
      if (ctx->lane_view) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_cand_count, 0, (size_t)ctx->B * sizeof(int32_t), ks));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_fix_count, 0, (size_t)ctx->B * sizeof(int32_t), ks));
      } else {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_cand_count, 0, 2 * (size_t)ctx->B * sizeof(int32_t), s));
      }
    }
    ctx->counters_cleared = false;
    if ((st = heavy_begin(ctx, s, 0, &token)) != OKVFE_OK) return st;
    bool fused;
    // lanes inside one call: the score kernels run one after the other, so that the lanes proceed OUT OF PHASE -- the
    // (vector-ALU-bound) score kernel of lane l beside the selection / descriptor kernels of the lanes before it
    if (ctx->k1_wait && ks == s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->k1_wait, 0));
    {
      StageTimer t(ctx, OKVFE_STAGE_HARRIS, ks);
      layer_score_nms(ctx, images_dev, n_images, ks, &fused);
    }
    if (ctx->k1_done) HIP_TRY(ctx, hipEventRecord(ctx->k1_done, ks));
    if (ks != s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->k1_done, 0));
    if ((st = heavy_end(ctx, s, 0, &token)) != OKVFE_OK) return st;
    {
      StageTimer t(ctx, OKVFE_STAGE_NMS, s);
      layer_nms_finish(ctx, n_images, s, fused);
    }
    {
      StageTimer t(ctx, OKVFE_STAGE_SORT, s);
      layer_sort(ctx, n_images, s);
    }
    {
      StageTimer t(ctx, OKVFE_STAGE_SELECT, s);
      layer_select(ctx, n_images, s);
    }
  } else if (ctx->prof_mask == 0 && score_token_mode() == 0 && !scale_space_serial()) {  // (stage timers and the
    // cross-context chaining of the heavy kernels belong to the one-stream form below)
    if ((st = detect_layers_concurrent(ctx, images_dev, n_images, s)) != OKVFE_OK) return st;
  } else {
    const int L = ctx->n_layers;
    std::vector<const uint8_t*> img(L);
    std::vector<bool> fused(L);
    img[0] = images_dev;
    if ((st = heavy_begin(ctx, s, 0, &token)) != OKVFE_OK) return st;
    {
      StageTimer t(ctx, OKVFE_STAGE_HARRIS, s);
      for (int l = 1; l < L; ++l) {
        if (l == 1)
          launch_twothird(img[0], ctx->layer_w[0], ctx->layer_h[0], n_images, ctx->d_layer_img[1], s);
        else
          launch_halfsample(img[l - 2], ctx->layer_w[l - 2], ctx->layer_h[l - 2], n_images, ctx->d_layer_img[l], s);
        img[l] = ctx->d_layer_img[l];
      }
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        HIP_TRY(ctx, hipMemsetAsync(ch->d_cand_count, 0, 2 * (size_t)ch->B * sizeof(int32_t), s));
        bool f;
        layer_score_nms(ch, img[l], n_images, s, &f);
        fused[l] = f;
      }
      if (ctx->d_virtual) launch_fast58_score(img[0], ctx->layer_w[0], ctx->layer_h[0], n_images, ctx->d_virtual, s);
    }
    if ((st = heavy_end(ctx, s, 0, &token)) != OKVFE_OK) return st;
    {
      StageTimer t(ctx, OKVFE_STAGE_NMS, s);
      for (int l = 0; l < L; ++l) layer_nms_finish(ctx->layers[l], n_images, s, fused[l]);
      // scale-space maxima: every layer against the finished score maps below and above
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        int sn, sd;
        layer_scale(l, &sn, &sd);
        const int32_t *below = nullptr, *above = nullptr;
        ScoreLayout lb{0, 0}, la{0, 0};
        int rb[2] = {1, 1}, ra[2] = {1, 1};
        auto ratio = [&](int m, int out[2]) {  // scale_l / scale_m, reduced
          int mn, md;
          layer_scale(m, &mn, &md);
          int rn = sn * md, rd = sd * mn;
          for (int g = 2; g <= 3; ++g)
            while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
          out[0] = rn; out[1] = rd;
        };
        if (l > 0) { below = ctx->layers[l - 1]->d_scores; lb = ctx->layers[l - 1]->live_layout; ratio(l - 1, rb); }
        if (l == 0 && ctx->d_virtual) { below = ctx->d_virtual; lb = ScoreLayout{ctx->layer_w[0], 0}; }  // same grid: ratio 1
        if (l + 1 < L) { above = ctx->layers[l + 1]->d_scores; la = ctx->layers[l + 1]->live_layout; ratio(l + 1, ra); }
        launch_scale_filter(ch->d_cand, ch->cand_cap, ch->d_cand_count, n_images, below, lb,
                            l > 0 ? ctx->layer_w[l - 1] : (below ? ctx->layer_w[0] : 0),
                            l > 0 ? ctx->layer_h[l - 1] : (below ? ctx->layer_h[0] : 0), rb[0], rb[1], above, la,
                            l + 1 < L ? ctx->layer_w[l + 1] : 0, l + 1 < L ? ctx->layer_h[l + 1] : 0, ra[0], ra[1], s);
      }
    }
    const bool brisk_ss = ctx->cfg.score_type == OKVFE_SCORE_BRISK_SCALESPACE;
    {
      StageTimer t(ctx, OKVFE_STAGE_SORT, s);
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        if (brisk_ss)  // always ordered (score desc, y, x): there is no uniformity radius to switch the sort on
          launch_sort(ch->d_cand, ch->cand_cap, ch->d_cand_count, n_images, 1.0f, ch->d_sort_ws, s);
        else
          layer_sort(ch, n_images, s);
      }
    }
    {
      StageTimer t(ctx, OKVFE_STAGE_SELECT, s);
      for (int l = 0; l < L; ++l) {
        okvfe_ctx* ch = ctx->layers[l];
        if (!brisk_ss) {
          layer_select(ch, n_images, s);
          continue;
        }
        // strongest maxima + continuous scale from the scores of the layers below and above
        int sn, sd;
        layer_scale(l, &sn, &sd);
        auto ratio2 = [&](int m, int out[2]) {
          int mn, md;
          layer_scale(m, &mn, &md);
          int rn = sn * md, rd = sd * mn;
          for (int g = 2; g <= 3; ++g)
            while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
          out[0] = rn; out[1] = rd;
        };
        int rb2[2] = {1, 1}, ra2[2] = {1, 1};
        const int32_t* below = l == 0 ? ctx->d_virtual : ctx->layers[l - 1]->d_scores;
        const int wb = l == 0 ? ctx->layer_w[0] : ctx->layer_w[l - 1], hb = l == 0 ? ctx->layer_h[0] : ctx->layer_h[l - 1];
        if (l > 0) ratio2(l - 1, rb2);
        const int32_t* above = l + 1 < L ? ctx->layers[l + 1]->d_scores : nullptr;
        if (above) ratio2(l + 1, ra2);
        // c_0: the virtual FAST 5-8 layer sits at 2/3 and the result is clamped to [0.7, 1.5] (published refine1D_2)
        const double rel_b = (l & 1) || l == 0 ? 2.0 / 3.0 : 0.75, rel_a = (l & 1) ? 4.0 / 3.0 : 1.5;
        const double rel_lo = l == 0 ? 0.7 : rel_b;
        launch_brisk_refine(ch->d_scores, ch->w, ch->h, n_images, ch->cand_cap, ch->d_cand_count, ch->d_sort_ws,
                            ch->cfg.max_keypoints, below, wb, hb, rb2[0], rb2[1], above,
                            above ? ctx->layer_w[l + 1] : 0, above ? ctx->layer_h[l + 1] : 0, ra2[0], ra2[1], rel_b,
                            rel_a, rel_lo, ch->d_kps_det, ch->kp_cap, ch->d_det_count, s);
      }
      const okvfe_keypoint* kps[8];
      const int32_t* counts[8];
      float scale[8];
      for (int l = 0; l < L; ++l) {
        kps[l] = ctx->layers[l]->d_kps_det;
        counts[l] = ctx->layers[l]->d_det_count;
        int sn, sd;
        layer_scale(l, &sn, &sd);
        scale[l] = (float)sn / (float)sd;
      }
      launch_merge_layers(kps, counts, scale, L, ctx->cfg.max_keypoints, n_images, ctx->d_kps_det, ctx->kp_cap,
                          ctx->d_det_count, s);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_n_images = n_images;
  ctx->last_stream = s;
  return OKVFE_OK;
}

// first image of the last batch whose NMS candidate list overflowed in any layer (-1 = none);
// synchronises the last stream
okvfe_status find_overflow(okvfe_ctx* ctx, int first, int n_images, int* bad, int* count, int* cap) {
  *bad = -1;
  { const okvfe_status js = lanes_join_host(ctx); if (js != OKVFE_OK) return js; }
  if (ctx->last_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
  std::vector<int32_t> counts(n_images);
  const int L = ctx->n_layers;
  for (int l = 0; l < L && *bad < 0; ++l) {
    okvfe_ctx* lc = L == 1 ? ctx : ctx->layers[l];
    HIP_TRY(ctx, hipMemcpy(counts.data(), lc->d_cand_count + first, (size_t)n_images * sizeof(int32_t),
                           hipMemcpyDeviceToHost));
    for (int i = 0; i < n_images; ++i)
      if (counts[i] > lc->cand_cap) {
        *bad = first + i;
        *count = counts[i];
        *cap = lc->cand_cap;
        break;
      }
  }
  return OKVFE_OK;
}

// K6 + compaction + back-projection of the keypoints detect_stage left in d_kps_det
okvfe_status describe_stage(okvfe_ctx* ctx, const uint8_t* images_dev, int n_images, hipStream_t s) {
  const int w = ctx->w, h = ctx->h;
  TokenScope token;
  const bool setup_done = ctx->setup_done;  // consumed here, whatever happens below
  ctx->setup_done = false;
  okvfe_status st = heavy_begin(ctx, s, 1, &token);
  if (st != OKVFE_OK) return st;
  {
    StageTimer t(ctx, OKVFE_STAGE_DESCRIBE, s);
    launch_describe(images_dev, w, h, n_images, ctx->d_pattern, ctx->d_prm,
                    ctx->d_rays_ptrs, ctx->d_jac_ptrs, ctx->d_kps_det, ctx->kp_cap, ctx->d_det_count,
                    ctx->d_kps_tmp, ctx->d_desc_tmp, ctx->d_valid_tmp, ctx->d_scales, ctx->wide_patches, s, setup_done,
                    ctx->all_aware, ctx->lane_view ? ctx->box_class_call : pattern_box_class(ctx->host_pattern),
                    ctx->lane_view || setup_done ? ctx->aware_extra_box : aware_box_for_call(ctx, images_dev),
                    ctx->none_aware && (ctx->lane_view ? ctx->rot_fast_call : pattern_rot_ok(ctx->host_pattern)));
  }
  if ((st = heavy_end(ctx, s, 1, &token)) != OKVFE_OK) return st;
  {
    StageTimer t(ctx, OKVFE_STAGE_COMPACT, s);
    launch_compact(n_images, ctx->d_cams, ctx->d_prm, ctx->d_kps_tmp, ctx->d_desc_tmp, ctx->d_valid_tmp,
                   ctx->d_det_count, ctx->kp_cap, ctx->d_kps, ctx->d_desc, ctx->d_bp, ctx->d_bpv,
                   ctx->d_count, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  if (ctx->lane_view) return OKVFE_OK;  // (the owner releases the parameter slot behind the join)
  const int slot = ctx->prm_slot;
  ctx->prm_slot = -1;
  return ring_release(ctx, &ctx->prm_ring, slot, s);  // the ImageParams slot has no reader after this
}

// ---- lanes inside one call (okvfe_ctx::internal_lanes) ---------------------------------------------------------------
int lanes_for_call(const okvfe_ctx* ctx, int n_images) {
  static const char* force = lab_env("OKVFE_INTERNAL_LANES");  // A/B knob
  int k = force ? atoi(force) : ctx->internal_lanes;
  if (ctx->n_layers != 1 || ctx->lane_view || ctx->child || !ctx->d_cand) return 1;
  // 0 = the library's choice, which is NOT to cut (measured, round 6, 3072 EuRoC stereo frames per call, one caller
  // stream): 1 / 2 / 3 / 4 / 6 lanes = 710 / 688 / 702 / 700 / 677 k stereo-frames/s, with the score kernels chained
  // 710 / 685 / 673 / 684 / 659 k -- every call ends in a join, so the lanes start each call in phase and the slices'
  // kernels are the unsplit kernels in quarters; the +8 % of several CONTEXTS on several streams (766 k at three) comes
  // from lanes that drift out of phase across calls, which one caller stream cannot have
  if (k == 0) k = 1;
  if (k > 8) k = 8;
  while (k > 1 && n_images / k < 64) --k;
  return k < 1 ? 1 : k;
}

okvfe_status ensure_join(okvfe_ctx* ctx) {
  if (!ctx->join_stream) {
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->join_stream, hipStreamNonBlocking));
  }
  if (!ctx->join_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->join_done, hipEventDisableTiming));
  return OKVFE_OK;
}

okvfe_status ensure_lanes(okvfe_ctx* ctx, int k) {
  if (!ctx->lane_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming));
  static const bool prio_env = lab_env("OKVFE_LANES_PRIO") != nullptr;  // A/B knob
  const bool prio = prio_env || ctx->lanes_prio;
  int p_low = 0, p_high = 0;
  if (prio) {
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&p_low, &p_high));  // (numerically: low >= high)
    if (!ctx->score_stream) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->score_stream, hipStreamNonBlocking, p_low));
    ctx->lanes_prio = true;
  }
  while ((int)ctx->lane_ctx.size() < k) {
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));  // (streams live on the context's device, whatever the caller's current one)
    if (prio)
      HIP_TRY(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, p_high));
    else
      HIP_TRY(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipStreamDestroy(st);
      HIP_TRY(ctx, e);
    }
    hipEvent_t k1 = nullptr;
    e = hipEventCreateWithFlags(&k1, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipStreamDestroy(st);
      (void)hipEventDestroy(ev);
      HIP_TRY(ctx, e);
    }
    okvfe_ctx* v = new okvfe_ctx();
    v->k1_done = k1;
    v->lane_view = true;
    v->prof_owner = ctx;
    v->stream = st;
    ctx->lane_ctx.push_back(v);
    ctx->lane_done.push_back(ev);
  }
  return OKVFE_OK;
}

// view `v` = images [first, first + n) of `p`'s buffers, with the state of the running call
void bind_lane(okvfe_ctx* v, okvfe_ctx* p, int first, int n) {
  const size_t f = (size_t)first, K = (size_t)p->kp_cap;
  v->cfg = p->cfg;
  v->w = p->w; v->h = p->h; v->B = n; v->kp_cap = p->kp_cap; v->cand_cap = p->cand_cap; v->ws_stride = p->ws_stride;
  v->occ_rows = p->occ_rows; v->occ_cols = p->occ_cols; v->occ_image_bytes = p->occ_image_bytes;
  v->mode_default = p->mode_default;
  v->score_layout = p->score_layout; v->live_layout = p->live_layout; v->keep_score_map = p->keep_score_map;
  v->n_layers = 1;
  v->d_scores = p->d_scores + f * (size_t)p->score_layout.pitch * p->h;
  v->d_cand = p->d_cand + f * p->cand_cap;
  v->d_cand_count = p->d_cand_count + f;
  v->d_fix_count = p->d_fix_count + f;
  v->d_fix_list = p->d_fix_list + f * kFixListCap;
  v->d_sort_ws = p->d_sort_ws + f * p->ws_stride;
  v->d_occ = p->d_occ ? p->d_occ + f * p->occ_image_bytes : nullptr;
  v->d_lut = p->d_lut; v->d_pattern = p->d_pattern; v->d_scales = p->d_scales;
  v->d_kps_det = p->d_kps_det + f * K; v->d_det_count = p->d_det_count + f;
  v->d_kps_tmp = p->d_kps_tmp + f * K; v->d_desc_tmp = p->d_desc_tmp + f * K * OKVFE_DESC_BYTES;
  v->d_valid_tmp = p->d_valid_tmp + f * K;
  v->d_kps = p->d_kps + f * K; v->d_desc = p->d_desc + f * K * OKVFE_DESC_BYTES;
  v->d_bp = p->d_bp + f * K * 3; v->d_bpv = p->d_bpv + f * K; v->d_count = p->d_count + f;
  v->d_prm = p->d_prm + f;
  v->d_cams = p->d_cams; v->d_rays_ptrs = p->d_rays_ptrs; v->d_jac_ptrs = p->d_jac_ptrs;
  v->wide_patches = p->wide_patches; v->all_aware = p->all_aware; v->aware_fast = p->aware_fast;
  v->aware_extra_box = p->aware_extra_box; v->box_class_call = p->box_class_call;
  v->none_aware = p->none_aware; v->rot_fast_call = p->rot_fast_call;
  v->fuse_setup = p->fuse_setup;
  v->counters_cleared = p->counters_cleared;
  v->prof_mask = p->prof_mask;
  v->prm_slot = -1;
}

// detect + describe of a device-resident batch, in lanes when the call is large enough
okvfe_status detect_describe_split(okvfe_ctx* ctx, const uint8_t* images_dev, int n_images, hipStream_t s,
                                   bool pipelined = false) {
  // (a context set to pipelined lanes splits only the calls that can stay un-joined; everything else runs unsplit)
  const int k = ctx->lanes_pipelined && !pipelined ? 1 : lanes_for_call(ctx, n_images);
  okvfe_status st;
  if (k <= 1) {
    st = detect_stage(ctx, images_dev, n_images, s);
    ctx->fuse_setup = false;
    if (st != OKVFE_OK) {
      ctx->setup_done = false;
      ctx->detected_images = 0;
      return st;
    }
    ctx->detected_images = n_images;
    return describe_stage(ctx, images_dev, n_images, s);
  }
  if ((st = ensure_lanes(ctx, k)) != OKVFE_OK) return st;
  // slices of whole stereo pairs, multiples of 8 images (the kernels' image -> XCD striping)
  int chunk = (n_images + k - 1) / k;
  chunk = (chunk + 7) & ~7;
  const size_t P = (size_t)ctx->w * ctx->h;
  ctx->box_class_call = pattern_box_class(ctx->host_pattern);
  ctx->rot_fast_call = pattern_rot_ok(ctx->host_pattern);
  ctx->aware_extra_box = aware_box_for_call(ctx, images_dev);
  HIP_TRY(ctx, hipEventRecord(ctx->lane_fork, s));  // behind the parameter upload (and whatever the caller queued)
  okvfe_status first_err = OKVFE_OK;
  int used = 0;
  for (int l = 0; l < k; ++l) {
    const int first = l * chunk;
    const int n = std::min(chunk, n_images - first);
    if (n <= 0) break;
    okvfe_ctx* v = ctx->lane_ctx[l];
    bind_lane(v, ctx, first, n);
    static const bool no_chain = lab_env("OKVFE_LANES_NOCHAIN") != nullptr;  // A/B knob
    // (pipelined lanes are never chained: each follows its own previous call, like separate contexts)
    v->k1_wait = l > 0 && !no_chain && !pipelined ? ctx->lane_ctx[l - 1]->k1_done : nullptr;
    v->score_stream = ctx->score_stream;
    HIP_TRY(ctx, hipStreamWaitEvent(v->stream, ctx->lane_fork, 0));
    if (ctx->score_stream && l == 0) HIP_TRY(ctx, hipStreamWaitEvent(ctx->score_stream, ctx->lane_fork, 0));
    st = detect_stage(v, images_dev + first * P, n, v->stream);
    v->fuse_setup = false;
    if (st == OKVFE_OK) st = describe_stage(v, images_dev + first * P, n, v->stream);
    if (st != OKVFE_OK && first_err == OKVFE_OK) {
      first_err = st;
      ctx->err = v->err;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->lane_done[l], v->stream));
    used = l + 1;
  }
  if (pipelined) {
    // no join onto the caller's stream: the join stream waits for every lane (it is what releases the parameter slot)
    // and `join_done` is what a later consumer waits for (pick_stream / lanes_join_host)
    if ((st = ensure_join(ctx)) != OKVFE_OK) return st;
    for (int l = 0; l < used; ++l) HIP_TRY(ctx, hipStreamWaitEvent(ctx->join_stream, ctx->lane_done[l], 0));
    ctx->lanes_pending = true;
    ctx->lanes_used = used;
    ctx->lane_chunk = chunk;
  } else {
    for (int l = 0; l < used; ++l) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->lane_done[l], 0));
  }
  ctx->fuse_setup = false;
  ctx->setup_done = false;
  ctx->counters_cleared = false;
  okvfe_ctx* v0 = ctx->lane_ctx[0];
  ctx->map_free_live = v0->map_free_live;
  ctx->live_layout = v0->live_layout;
  ctx->live_images = images_dev;
  ctx->last_stream = s;
  ctx->last_n_images = n_images;
  const int slot = ctx->prm_slot;
  ctx->prm_slot = -1;
  st = ring_release(ctx, &ctx->prm_ring, slot, pipelined ? ctx->join_stream : s);  // behind the join: every lane has read its parameters
  if (pipelined && st == OKVFE_OK) HIP_TRY(ctx, hipEventRecord(ctx->join_done, ctx->join_stream));
  if (first_err != OKVFE_OK) {
    ctx->detected_images = 0;
    return first_err;
  }
  ctx->detected_images = n_images;
  return st;
}
}  // namespace

okvfe_status okvfe_detect_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                       void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 1 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_batch_device: n_images=%d (max_batch %d)",
                n_images, ctx->B);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  ctx->detected_images = 0;
  okvfe_status st = detect_stage(ctx, images_dev, n_images, pick_stream(ctx, stream));
  if (st == OKVFE_OK) ctx->detected_images = n_images;
  return st;
}

okvfe_status okvfe_describe_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev, int32_t n_images,
                                         const int32_t* cam_ids, const float* gravity_C, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 1 || n_images != ctx->detected_images)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT,
                "okvfe_describe_batch_device: n_images=%d, but the last okvfe_detect_batch_device "
                "covered %d", n_images, ctx->detected_images);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  okvfe_status st = upload_image_params(ctx, n_images, cam_ids, gravity_C, s);
  if (st != OKVFE_OK) return st;
  return describe_stage(ctx, images_dev, n_images, s);
}

okvfe_status okvfe_detect_describe_batch_device(okvfe_ctx* ctx, const uint8_t* images_dev,
                                                int32_t n_images, const int32_t* cam_ids,
                                                const float* gravity_C, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_dev || n_images < 1 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_describe_batch_device: n_images=%d (max_batch %d)",
                n_images, ctx->B);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  // pipelined lanes (okvfe_set_internal_lanes(-k)): no join onto the caller's stream, neither before nor after the call,
  // as long as the slices stay what they were (a lane only ever follows ITS OWN previous work)
  bool piped = false;
  if (ctx->lanes_pipelined) {
    const int k = lanes_for_call(ctx, n_images);
    const int chunk = (((n_images + k - 1) / k) + 7) & ~7;
    piped = k > 1 && (!ctx->lanes_pending || ctx->lane_chunk == chunk);
  }
  hipStream_t s = piped ? pick_stream_raw(ctx, stream) : pick_stream(ctx, stream);
  // (pipelined: the candidate counters are cleared by every lane on its own stream, not by the parameter upload on the
  // caller's -- a lane may still be reading last call's)
  okvfe_status st = upload_image_params(ctx, n_images, cam_ids, gravity_C, s, !piped);
  if (st != OKVFE_OK) return st;
  static const bool no_fuse = lab_env("OKVFE_NO_FUSED_SETUP") != nullptr;  // A/B knob
  ctx->fuse_setup = !no_fuse;
  return detect_describe_split(ctx, images_dev, n_images, s, piped);
}

okvfe_status okvfe_detect_describe_batch_host(okvfe_ctx* ctx, const uint8_t* images_host, int32_t n_images,
                                              const int32_t* cam_ids, const float* gravity_C, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!images_host || n_images < 1 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_describe_batch_host: n_images=%d (max_batch %d)",
                n_images, ctx->B);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const size_t P = (size_t)ctx->w * ctx->h;
  if (!ctx->feed_stream) {
    // the feed state becomes visible only when ALL of it exists: a failed allocation leaves the
    // context as it was (the next call tries again) instead of a stream without buffers
    hipStream_t fs = nullptr;
    uint8_t* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipStreamCreateWithFlags(&fs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      void* q = nullptr;
      e = hipMalloc(&q, P * (size_t)ctx->B);
      buf[i] = static_cast<uint8_t*>(q);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[2 * i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[2 * i + 1], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
      for (hipEvent_t v : ev)
        if (v) (void)hipEventDestroy(v);
      for (uint8_t* b : buf)
        if (b) (void)hipFree(b);
      if (fs) (void)hipStreamDestroy(fs);
      HIP_TRY(ctx, e);
    }
    for (int i = 0; i < 2; ++i) {
      ctx->d_feed[i] = buf[i];
      ctx->feed_copied[i] = ev[2 * i];
      ctx->feed_consumed[i] = ev[2 * i + 1];
    }
    ctx->feed_stream = fs;
  }
  const int slot = (int)(ctx->feed_next++ & 1u);
  // the buffer is rewritten only after the kernels of the batch that used it two calls ago
  if (ctx->feed_busy[slot]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->feed_stream, ctx->feed_consumed[slot], 0));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_feed[slot], images_host, P * (size_t)n_images, hipMemcpyHostToDevice,
                              ctx->feed_stream));
  HIP_TRY(ctx, hipEventRecord(ctx->feed_copied[slot], ctx->feed_stream));
  HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->feed_copied[slot], 0));
  okvfe_status st = upload_image_params(ctx, n_images, cam_ids, gravity_C, s, true);
  if (st != OKVFE_OK) return st;
  ctx->fuse_setup = lab_env("OKVFE_NO_FUSED_SETUP") == nullptr;
  if ((st = detect_describe_split(ctx, ctx->d_feed[slot], n_images, s)) != OKVFE_OK) return st;
  HIP_TRY(ctx, hipEventRecord(ctx->feed_consumed[slot], s));
  ctx->feed_busy[slot] = true;
  return OKVFE_OK;
}

okvfe_status okvfe_get_device_outputs(okvfe_ctx* ctx, okvfe_device_outputs* out) {
  if (!ctx || !out) return OKVFE_ERR_INVALID_ARGUMENT;
  { const okvfe_status js = lanes_join_host(ctx); if (js != OKVFE_OK) return js; }  // (pipelined lanes: the caller is about to read)
  out->max_keypoints = ctx->kp_cap;
  out->counts = ctx->d_count;
  out->keypoints = ctx->d_kps;
  out->descriptors = ctx->d_desc;
  out->backproj = ctx->d_bp;
  out->backproj_valid = ctx->d_bpv;
  // (null after a map-free call: see okvfe_set_keep_score_map)
  out->scores = (ctx->n_layers == 1 && ctx->map_free_live) ? nullptr : ctx->d_scores;
  out->detect_counts = ctx->d_det_count;
  out->candidate_counts = ctx->d_cand_count;
  // the layout of the LAST batch's map (dense if that call took the unfused score + NMS kernels)
  const ScoreLayout& sl = ctx->n_layers > 1 ? ctx->layers[0]->live_layout : ctx->live_layout;
  // no map was written (okvfe_set_keep_score_map): no layout either -- okvfe_harris_score_device is the way to a map
  out->score_pitch = out->scores ? sl.pitch : 0;
  out->score_strips = out->scores ? sl.strips : 0;
  return OKVFE_OK;
}

int32_t okvfe_scale_index(float keypoint_size) { return pattern_scale_index(keypoint_size); }

int32_t okvfe_score_column(const okvfe_ctx* ctx, int32_t x) {
  return ctx ? score_col(ctx->n_layers > 1 ? ctx->layers[0]->live_layout : ctx->live_layout, x) : x;
}

okvfe_status okvfe_check_capacity(okvfe_ctx* ctx, int32_t n_images, int32_t* first_overflowed) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_images < 0 || n_images > ctx->B)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_check_capacity: n_images=%d (max_batch %d)", n_images, ctx->B);
  if (first_overflowed) *first_overflowed = -1;
  if (n_images == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  int bad = -1, count = 0, cap = 0;
  okvfe_status st = find_overflow(ctx, 0, n_images, &bad, &count, &cap);
  if (st != OKVFE_OK) return st;
  if (bad >= 0) {
    if (first_overflowed) *first_overflowed = bad;
    return fail(ctx, OKVFE_ERR_CAPACITY,
                "image %d produced %d NMS maxima, candidate capacity is %d (its keypoint list was left empty)", bad,
                count, cap);
  }
  return OKVFE_OK;
}

}  // extern "C"  (runtime helpers of the single-image calls follow)

namespace {
ResultLayout result_layout(int kp_cap) {
  const BlockLayout B = block_layout(kp_cap);
  ResultLayout L;
  L.o_count = (int32_t)B.o_count;
  L.o_kps = (int32_t)B.o_kps;
  L.o_desc = (int32_t)B.o_desc;
  L.o_bp = (int32_t)B.o_bp;
  L.o_bpv = (int32_t)B.o_bpv;
  L.o_det = (int32_t)B.total;
  L.total = (int32_t)align_up(B.total + (size_t)kp_cap * sizeof(okvfe_keypoint), 256);
  return L;
}

okvfe_status ensure_result_block(okvfe_ctx* ctx) {
  if (ctx->h_result) return OKVFE_OK;
  void* p = nullptr;
  HIP_TRY(ctx, hipHostMalloc(&p, (size_t)result_layout(ctx->kp_cap).total, hipHostMallocDefault));
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) {
    (void)hipGetLastError();
    (void)hipHostFree(p);
    return fail(ctx, OKVFE_ERR_DEVICE, "pinned result block is not device-visible");
  }
  ctx->h_result = static_cast<uint8_t*>(p);
  ctx->h_result_dev = d;
  return OKVFE_OK;
}

// Results of image `index` of the last call -> ctx->h_result, behind everything enqueued on s: one
// kernel, ONE host synchronisation.  final_results = false: only the detector's keypoints exist.
okvfe_status export_and_wait(okvfe_ctx* ctx, int index, bool final_results, hipStream_t s) {
  okvfe_status st = ensure_result_block(ctx);
  if (st != OKVFE_OK) return st;
  ResultSrc src{};
  if (final_results) {
    src.count = ctx->d_count;
    src.kps = ctx->d_kps;
    src.desc = ctx->d_desc;
    src.bp = ctx->d_bp;
    src.bpv = ctx->d_bpv;
  }
  src.det_count = ctx->d_det_count;
  src.det_kps = ctx->d_kps_det;
  src.cand_count = ctx->d_cand_count;  // (scale space: layer 0's; every layer is checked below)
  launch_export_result(src, index, ctx->kp_cap, result_layout(ctx->kp_cap), ctx->h_result_dev, s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(s));
  const int32_t* hdr = reinterpret_cast<const int32_t*>(ctx->h_result);
  if (ctx->n_layers == 1) {
    if (hdr[1] > ctx->cand_cap)
      return fail(ctx, OKVFE_ERR_CAPACITY, "image %d produced %d NMS maxima, candidate capacity is %d", index, hdr[1],
                  ctx->cand_cap);
  } else {
    int bad = -1, cnt = 0, cap_c = 0;
    if ((st = find_overflow(ctx, index, 1, &bad, &cnt, &cap_c)) != OKVFE_OK) return st;
    if (bad >= 0)
      return fail(ctx, OKVFE_ERR_CAPACITY, "image %d produced %d NMS maxima, candidate capacity is %d", index, cnt,
                  cap_c);
  }
  return OKVFE_OK;
}

// h_result -> the caller's arrays (final results)
okvfe_status copy_results_out(okvfe_ctx* ctx, okvfe_keypoint* keypoints, uint8_t* descriptors, double* backproj,
                              uint8_t* backproj_valid, int32_t cap, int32_t* n_out) {
  const ResultLayout L = result_layout(ctx->kp_cap);
  const int n = reinterpret_cast<const int32_t*>(ctx->h_result)[0];
  *n_out = n;
  if (n > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
  if (n > 0) {
    if (keypoints) std::memcpy(keypoints, ctx->h_result + L.o_kps, (size_t)n * sizeof(okvfe_keypoint));
    if (descriptors) std::memcpy(descriptors, ctx->h_result + L.o_desc, (size_t)n * OKVFE_DESC_BYTES);
    if (backproj) std::memcpy(backproj, ctx->h_result + L.o_bp, (size_t)n * 3 * sizeof(double));
    if (backproj_valid) std::memcpy(backproj_valid, ctx->h_result + L.o_bpv, (size_t)n);
  }
  return OKVFE_OK;
}

// the caller's image -> pinned staging -> d_img_stage on the context's stream (a copy kernel reads
// the pinned rows in place: no hand-over between the DMA engine and the compute queue)
okvfe_status stage_image(okvfe_ctx* ctx, const uint8_t* image, size_t stride, bool keep_shadow = false) {
  const size_t P = (size_t)ctx->w * ctx->h;
  if (stride < (size_t)ctx->w) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "stride %zu < width %d", stride, ctx->w);
  // (+16: launch_param_copy moves whole 16-byte chunks, kp_cap * 28 need not be a multiple of 16)
  okvfe_status st = ensure_pinned(ctx, align_up(P, 256) + 256 + align_up((size_t)ctx->kp_cap * sizeof(okvfe_keypoint), 16) + 16);
  if (st != OKVFE_OK) return st;
  if (stride == (size_t)ctx->w) {
    std::memcpy(ctx->h_pinned, image, P);
  } else {
    for (int y = 0; y < ctx->h; ++y) std::memcpy(ctx->h_pinned + (size_t)y * ctx->w, image + (size_t)y * stride, ctx->w);
  }
  if (keep_shadow) {
    // pairing check of okvfe_compute: compared against ordinary memory (reads of the pinned staging
    // buffer run at a fraction of the speed of cached memory)
    ctx->ahead_shadow.resize(P);
    if (stride == (size_t)ctx->w) {
      std::memcpy(ctx->ahead_shadow.data(), image, P);
    } else {
      for (int y = 0; y < ctx->h; ++y)
        std::memcpy(ctx->ahead_shadow.data() + (size_t)y * ctx->w, image + (size_t)y * stride, ctx->w);
    }
  }
  if (ctx->h_pinned_dev) {
    launch_param_copy(ctx->d_img_stage, ctx->h_pinned_dev, P, nullptr, 0, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_img_stage, ctx->h_pinned, P, hipMemcpyHostToDevice, ctx->stream));
  }
  return OKVFE_OK;
}

bool same_rows(const okvfe_ctx* ctx, const uint8_t* image, size_t stride) {  // caller's image == the one detected on?
  const uint8_t* ref = ctx->ahead_shadow.data();
  if (ctx->ahead_shadow.size() != (size_t)ctx->w * ctx->h) return false;
  if (stride == (size_t)ctx->w) return std::memcmp(ref, image, (size_t)ctx->w * ctx->h) == 0;
  for (int y = 0; y < ctx->h; ++y)
    if (std::memcmp(ref + (size_t)y * ctx->w, image + (size_t)y * stride, ctx->w) != 0) return false;
  return true;
}
}  // namespace

extern "C" {

okvfe_status okvfe_download_image_result(okvfe_ctx* ctx, int32_t index, okvfe_keypoint* keypoints,
                                         uint8_t* descriptors, double* backproj,
                                         uint8_t* backproj_valid, int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (index < 0 || index >= ctx->last_n_images || !n_out || cap < 0)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_download_image_result: index %d of %d", index,
                ctx->last_n_images);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  ctx->ahead.valid = false;  // h_result is reused
  okvfe_status st = lanes_join_host(ctx);  // (pipelined lanes: the results are read on the host next)
  if (st != OKVFE_OK) return st;
  st = export_and_wait(ctx, index, true, ctx->last_stream ? ctx->last_stream : ctx->stream);
  if (st != OKVFE_OK) return st;
  return copy_results_out(ctx, keypoints, descriptors, backproj, backproj_valid, cap, n_out);
}

okvfe_status okvfe_detect_describe(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                                   const float gravity_C[3], okvfe_keypoint* keypoints,
                                   uint8_t* descriptors, double* backproj, uint8_t* backproj_valid,
                                   int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_describe: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  ctx->ahead.valid = false;
  okvfe_status st = stage_image(ctx, image, stride);
  if (st != OKVFE_OK) return st;
  const int32_t cam_id = cam;
  st = okvfe_detect_describe_batch_device(ctx, ctx->d_img_stage, 1, &cam_id, (cam >= 0) ? gravity_C : nullptr,
                                          ctx->stream);
  if (st != OKVFE_OK) return st;
  if ((st = export_and_wait(ctx, 0, true, ctx->stream)) != OKVFE_OK) return st;
  return copy_results_out(ctx, keypoints, descriptors, backproj, backproj_valid, cap, n_out);
}

okvfe_status okvfe_detect(okvfe_ctx* ctx, const uint8_t* image, size_t stride, okvfe_keypoint* keypoints,
                          int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  ctx->ahead.valid = false;
  okvfe_status st = stage_image(ctx, image, stride);
  if (st != OKVFE_OK) return st;
  hipStream_t s = ctx->stream;
  if ((st = detect_stage(ctx, ctx->d_img_stage, 1, s)) != OKVFE_OK) return st;
  if ((st = export_and_wait(ctx, 0, false, s)) != OKVFE_OK) return st;
  const int n = reinterpret_cast<const int32_t*>(ctx->h_result)[2];
  *n_out = n;
  if (n > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
  if (n > 0 && keypoints)
    std::memcpy(keypoints, ctx->h_result + result_layout(ctx->kp_cap).o_det, (size_t)n * sizeof(okvfe_keypoint));
  return OKVFE_OK;
}

okvfe_status okvfe_detect_ahead(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                                const float gravity_C[3], okvfe_keypoint* keypoints, int32_t cap, int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_detect_ahead: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  ctx->ahead.valid = false;
  okvfe_status st = stage_image(ctx, image, stride, true);
  if (st != OKVFE_OK) return st;
  const int32_t cam_id = cam;
  const bool aware = cam >= 0 && gravity_C != nullptr;
  st = okvfe_detect_describe_batch_device(ctx, ctx->d_img_stage, 1, &cam_id, aware ? gravity_C : nullptr, ctx->stream);
  if (st != OKVFE_OK) return st;
  if ((st = export_and_wait(ctx, 0, true, ctx->stream)) != OKVFE_OK) return st;
  const int n = reinterpret_cast<const int32_t*>(ctx->h_result)[2];
  *n_out = n;
  if (n > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
  if (n > 0 && keypoints)
    std::memcpy(keypoints, ctx->h_result + result_layout(ctx->kp_cap).o_det, (size_t)n * sizeof(okvfe_keypoint));
  ctx->ahead.valid = true;
  ctx->ahead.image = image;
  ctx->ahead.stride = stride;
  ctx->ahead.cam = cam;
  ctx->ahead.aware = aware;
  for (int i = 0; i < 3; ++i) ctx->ahead.g[i] = aware ? gravity_C[i] : 0.0f;
  return OKVFE_OK;
}

okvfe_status okvfe_compute(okvfe_ctx* ctx, const uint8_t* image, size_t stride, int32_t cam,
                           const float gravity_C[3], okvfe_keypoint* keypoints, int32_t n_in,
                           uint8_t* descriptors, double* backproj, uint8_t* backproj_valid,
                           int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!image || !n_out || n_in < 0 || (n_in > 0 && !keypoints))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_compute: bad argument");
  if (n_in > ctx->kp_cap)
    return fail(ctx, OKVFE_ERR_CAPACITY, "okvfe_compute: %d keypoints exceed max_keypoints %d", n_in, ctx->kp_cap);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  const bool aware = cam >= 0 && gravity_C != nullptr;
  // answered ahead?  Same image (pointer, stride AND content), same extraction set-up, and exactly
  // the keypoints okvfe_detect_ahead returned: the descriptors are already in the result block
  if (ctx->ahead.valid) {
    const okvfe_ctx::Ahead& a = ctx->ahead;
    const ResultLayout L = result_layout(ctx->kp_cap);
    const int32_t* hdr = reinterpret_cast<const int32_t*>(ctx->h_result);
    const bool hit = a.image == image && a.stride == stride && a.cam == cam && a.aware == aware &&
                     (!aware || std::memcmp(a.g, gravity_C, sizeof(a.g)) == 0) && hdr[2] == n_in &&
                     (n_in == 0 || std::memcmp(keypoints, ctx->h_result + L.o_det, (size_t)n_in * sizeof(okvfe_keypoint)) == 0) &&
                     same_rows(ctx, image, stride);
    if (hit) return copy_results_out(ctx, keypoints, descriptors, backproj, backproj_valid, n_in, n_out);
    ctx->ahead.valid = false;
  }
  okvfe_status st = stage_image(ctx, image, stride);
  if (st != OKVFE_OK) return st;
  hipStream_t s = ctx->stream;
  const int32_t cam_id = cam;
  st = upload_image_params(ctx, 1, &cam_id, aware ? gravity_C : nullptr, s);
  if (st != OKVFE_OK) return st;
  const int w = ctx->w, h = ctx->h;
  // keypoints ride behind the image in the pinned staging buffer; the count is a kernel argument
  const size_t o_kp = align_up((size_t)w * h, 256) + 256;
  if (n_in > 0) std::memcpy(ctx->h_pinned + o_kp, keypoints, (size_t)n_in * sizeof(okvfe_keypoint));
  if (ctx->h_pinned_dev) {
    launch_param_copy(ctx->d_kps_det, static_cast<uint8_t*>(ctx->h_pinned_dev) + o_kp,
                      (size_t)n_in * sizeof(okvfe_keypoint), ctx->d_cand_count, 1, s, ctx->d_det_count, n_in);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    std::memcpy(ctx->h_pinned + o_kp - 256, &n_in, sizeof(int32_t));
    if (n_in > 0)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_kps_det, ctx->h_pinned + o_kp, (size_t)n_in * sizeof(okvfe_keypoint),
                                  hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_det_count, ctx->h_pinned + o_kp - 256, sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_cand_count, 0, sizeof(int32_t), s));
  }
  launch_describe(ctx->d_img_stage, w, h, 1, ctx->d_pattern, ctx->d_prm, ctx->d_rays_ptrs,
                  ctx->d_jac_ptrs, ctx->d_kps_det, ctx->kp_cap, ctx->d_det_count, ctx->d_kps_tmp,
                  ctx->d_desc_tmp, ctx->d_valid_tmp, ctx->d_scales, ctx->wide_patches, s, false,
                  ctx->all_aware, pattern_box_class(ctx->host_pattern), aware_box_for_call(ctx, ctx->d_img_stage),
                  ctx->none_aware && pattern_rot_ok(ctx->host_pattern));
  launch_compact(1, ctx->d_cams, ctx->d_prm, ctx->d_kps_tmp, ctx->d_desc_tmp, ctx->d_valid_tmp, ctx->d_det_count,
                 ctx->kp_cap, ctx->d_kps, ctx->d_desc, ctx->d_bp, ctx->d_bpv, ctx->d_count, s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_n_images = 1;
  ctx->last_stream = s;
  {
    const int slot = ctx->prm_slot;
    ctx->prm_slot = -1;
    if ((st = ring_release(ctx, &ctx->prm_ring, slot, s)) != OKVFE_OK) return st;
  }
  if ((st = export_and_wait(ctx, 0, true, s)) != OKVFE_OK) return st;
  return copy_results_out(ctx, keypoints, descriptors, backproj, backproj_valid, n_in, n_out);
}

}  // extern "C"

#ifdef OKVFE_LAB
// lab build only: buffers of one scale-space layer after the last call (what: 0 = score map, 1 = layer image,
// 2 = candidate records, 3 = candidate counts); returns the bytes the buffer holds (copies min(bytes, that))
extern "C" long long okvfe_lab_dump_layer(okvfe_ctx* ctx, int layer, int what, void* host, size_t bytes) {
  if (!ctx || layer < 0 || layer >= (int)ctx->layers.size()) return -1;
  (void)hipDeviceSynchronize();
  okvfe_ctx* ch = ctx->layers[layer];
  const void* src = nullptr;
  size_t n = 0;
  if (what == 0) { src = ch->d_scores; n = (size_t)ch->live_layout.pitch * ch->h * ch->B * 4; }
  if (what == 1) { src = layer == 0 ? nullptr : ctx->d_layer_img[layer]; n = (size_t)ch->w * ch->h * ch->B; }
  if (what == 2) { src = ch->d_cand; n = (size_t)ch->cand_cap * ch->B * sizeof(okvfe::Candidate); }
  if (what == 3) { src = ch->d_cand_count; n = (size_t)ch->B * 4; }
  if (!src) return 0;
  if (host && hipMemcpy(host, src, bytes < n ? bytes : n, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (long long)n;
}
#endif

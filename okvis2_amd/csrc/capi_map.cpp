// capi_map.cpp -- landmark-to-frame association behind the C ABI: matchToMap incl. landmark
// projection and descriptor-view pooling (Frontend.cpp:1219-1359, 1552-1589), the un-initialised
// variant (:1616-1719), verifyRecognisedPlace (:330-355), and the DBoW2 query path with the FBrisk
// trait (FBrisk.cpp:64-67; Frontend.cpp:756-766).
#include "okvfe_ctx.h"

using namespace okvfe;

extern "C" {

okvfe_status okvfe_match_to_map(okvfe_ctx* ctx, const uint8_t* desc, const okvfe_keypoint* kps, const uint8_t* use,
                                int32_t n_kps, const double* projections_l2, const int32_t* desc_begin,
                                int32_t n_landmarks, const uint8_t* pool, double reprojection_threshold,
                                int32_t* best_landmark, int32_t* best_dist) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_kps < 0 || n_landmarks < 0 || !desc_begin || !(reprojection_threshold >= 0.0) ||
      (n_kps > 0 && (!desc || !kps || !use || !best_landmark || !best_dist)) ||
      (n_landmarks > 0 && (!projections_l2 || !pool)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map: bad argument");
  for (int l = 0; l < n_landmarks; ++l)
    if (desc_begin[l + 1] < desc_begin[l] || desc_begin[l] < 0)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map: desc_begin not monotone at %d", l);
  if (n_kps == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const int n_pool = desc_begin[n_landmarks];
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_d = take((size_t)n_kps * 48), o_k = take((size_t)n_kps * sizeof(okvfe_keypoint)), o_u = take(n_kps);
  const size_t o_p = take((size_t)n_landmarks * 16), o_b = take((size_t)(n_landmarks + 1) * 4), o_pool = take((size_t)n_pool * 48);
  const size_t o_lm = take((size_t)n_kps * 4), o_bd = take((size_t)n_kps * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_d, desc, (size_t)n_kps * 48));
  HIP_TRY(ctx, up(o_k, kps, (size_t)n_kps * sizeof(okvfe_keypoint)));
  HIP_TRY(ctx, up(o_u, use, n_kps));
  HIP_TRY(ctx, up(o_p, projections_l2, (size_t)n_landmarks * 16));
  HIP_TRY(ctx, up(o_b, desc_begin, (size_t)(n_landmarks + 1) * 4));
  HIP_TRY(ctx, up(o_pool, pool, (size_t)n_pool * 48));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_match_to_map(base + o_d, reinterpret_cast<okvfe_keypoint*>(base + o_k), base + o_u, n_kps,
                      reinterpret_cast<double*>(base + o_p), reinterpret_cast<int32_t*>(base + o_b), n_landmarks,
                      base + o_pool, reprojection_threshold * reprojection_threshold, ctx->cfg.match_threshold,
                      reinterpret_cast<int32_t*>(base + o_lm), reinterpret_cast<int32_t*>(base + o_bd), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(best_landmark, base + o_lm, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_bd, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map_landmarks(okvfe_ctx* ctx, int32_t cam, const okvfe_landmark_table* T,
                                          const okvfe_pose* T_WC1, double reprojection_threshold, int32_t exclusive,
                                          const uint8_t* desc, const okvfe_keypoint* kps, const uint8_t* use,
                                          int32_t n_kps, okvfe_landmark_pool* pool_out, int32_t* best_landmark,
                                          int32_t* best_dist) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!T || !T_WC1 || n_kps < 0 || !(reprojection_threshold >= 0.0) || T->n_landmarks < 0 || T->n_observations < 0 ||
      T->n_poses < 0 || !T->obs_begin || (n_kps > 0 && (!desc || !kps || !use || !best_landmark || !best_dist)) ||
      (T->n_landmarks > 0 && (!T->hp_W || !T->quality)) ||
      (T->n_observations > 0 && (!T->obs_pose || !T->obs_desc || !T->obs_backproj || !T->poses)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_landmarks: bad argument");
  if (cam < 0 || cam >= (int)ctx->h_cams.size() || !(ctx->h_cams[cam].fu > 0.0))
    return fail(ctx, OKVFE_ERR_NOT_READY, "okvfe_match_to_map_landmarks: camera slot %d has no intrinsics (okvfe_set_camera)", cam);
  const int nl = T->n_landmarks, no = T->n_observations;
  for (int l = 0; l < nl; ++l)
    if (T->obs_begin[l + 1] < T->obs_begin[l] || T->obs_begin[l] < 0 || T->obs_begin[l + 1] > no)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_landmarks: obs_begin not monotone at %d", l);
  for (int o = 0; o < no; ++o)
    if (T->obs_pose[o] < 0 || T->obs_pose[o] >= T->n_poses)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_landmarks: observation %d: pose index out of range", o);
  for (int k = 0; k < n_kps; ++k) {
    best_landmark[k] = -1;
    best_dist[k] = ctx->cfg.match_threshold;
  }
  if (nl == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  // inputs
  const size_t o_hp = take((size_t)nl * 32), o_q = take((size_t)nl * 8), o_ob = take((size_t)(nl + 1) * 4),
               o_op = take((size_t)no * 4), o_od = take((size_t)no * 48), o_obp = take((size_t)no * 24),
               o_poses = take((size_t)T->n_poses * sizeof(okvfe_pose));
  const size_t o_d = take((size_t)n_kps * 48), o_k = take((size_t)n_kps * sizeof(okvfe_keypoint)), o_u = take(n_kps);
  // pooling results
  const size_t o_st = take((size_t)nl * 4), o_nd = take((size_t)nl * 4), o_rows = take((size_t)nl * 12),
               o_proj = take((size_t)nl * 16), o_e = take((size_t)nl * 48), o_r = take((size_t)nl * 48);
  // packed 3-D set + matcher outputs
  const size_t o_idx = take((size_t)nl * 4), o_p3 = take((size_t)nl * 16), o_b3 = take((size_t)(nl + 1) * 4),
               o_pool3 = take((size_t)nl * 2 * 48), o_n3 = take(8), o_lm = take((size_t)n_kps * 4),
               o_bd = take((size_t)n_kps * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_hp, T->hp_W, (size_t)nl * 32));
  HIP_TRY(ctx, up(o_q, T->quality, (size_t)nl * 8));
  HIP_TRY(ctx, up(o_ob, T->obs_begin, (size_t)(nl + 1) * 4));
  HIP_TRY(ctx, up(o_op, T->obs_pose, (size_t)no * 4));
  HIP_TRY(ctx, up(o_od, T->obs_desc, (size_t)no * 48));
  HIP_TRY(ctx, up(o_obp, T->obs_backproj, (size_t)no * 24));
  HIP_TRY(ctx, up(o_poses, T->poses, (size_t)T->n_poses * sizeof(okvfe_pose)));
  HIP_TRY(ctx, up(o_d, desc, (size_t)n_kps * 48));
  HIP_TRY(ctx, up(o_k, kps, (size_t)n_kps * sizeof(okvfe_keypoint)));
  HIP_TRY(ctx, up(o_u, use, n_kps));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // pageable sources
  const DeviceCamera& dc = ctx->h_cams[cam];
  const double focal = dc.fu + dc.fv;  // the SUM, as at Frontend.cpp:1213-1215
  auto I = [&](size_t o) { return reinterpret_cast<int32_t*>(base + o); };
  auto D = [&](size_t o) { return reinterpret_cast<double*>(base + o); };
  launch_prepare_landmarks(D(o_hp), D(o_q), I(o_ob), nl, I(o_op), D(o_obp),
                           reinterpret_cast<const okvfe_pose*>(base + o_poses), *T_WC1, ctx->d_cams + cam, ctx->w,
                           ctx->h, reprojection_threshold, exclusive ? 1 : 0, std::cos(10.0 / focal), std::cos(0.6),
                           I(o_st), I(o_nd), I(o_rows), D(o_proj), D(o_e), D(o_r), s);
  launch_compact_landmarks(I(o_st), I(o_nd), I(o_rows), D(o_proj), base + o_od, nl, 1, I(o_idx), D(o_p3), I(o_b3),
                           base + o_pool3, I(o_n3), s);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n3[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(n3, base + o_n3, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // the matcher's grid depends on the number of 3-D landmarks only
  if (n_kps > 0 && n3[0] > 0) {
    launch_match_to_map(base + o_d, reinterpret_cast<okvfe_keypoint*>(base + o_k), base + o_u, n_kps, D(o_p3),
                        I(o_b3), n3[0], base + o_pool3, reprojection_threshold * reprojection_threshold,
                        ctx->cfg.match_threshold, I(o_lm), I(o_bd), s);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<int32_t> idx(n3[0]);
    HIP_TRY(ctx, hipMemcpyAsync(best_landmark, base + o_lm, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_bd, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(idx.data(), base + o_idx, (size_t)n3[0] * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    for (int k = 0; k < n_kps; ++k)
      if (best_landmark[k] >= 0) best_landmark[k] = idx[best_landmark[k]];  // packed -> table index
  }
  if (pool_out) {
    if (pool_out->status) HIP_TRY(ctx, hipMemcpyAsync(pool_out->status, base + o_st, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
    if (pool_out->n_desc) HIP_TRY(ctx, hipMemcpyAsync(pool_out->n_desc, base + o_nd, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
    if (pool_out->obs_rows) HIP_TRY(ctx, hipMemcpyAsync(pool_out->obs_rows, base + o_rows, (size_t)nl * 12, hipMemcpyDeviceToHost, s));
    if (pool_out->projection) HIP_TRY(ctx, hipMemcpyAsync(pool_out->projection, base + o_proj, (size_t)nl * 16, hipMemcpyDeviceToHost, s));
    if (pool_out->e_W) HIP_TRY(ctx, hipMemcpyAsync(pool_out->e_W, base + o_e, (size_t)nl * 48, hipMemcpyDeviceToHost, s));
    if (pool_out->r_W) HIP_TRY(ctx, hipMemcpyAsync(pool_out->r_W, base + o_r, (size_t)nl * 48, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
  }
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map_uninitialised(okvfe_ctx* ctx, const uint8_t* desc, const double* backproj,
                                              const uint8_t* use, const int32_t* previous_landmark,
                                              int32_t n_kps, const int32_t* desc_begin, int32_t n_landmarks,
                                              const uint8_t* pool, const double* e0_W, const double* r0_W,
                                              const okvfe_pose* T_WC1, double focal_length,
                                              int32_t* best_landmark, int32_t* best_dist, double* hps_W,
                                              uint8_t* hp_set, int32_t* already_matched) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_kps < 0 || n_landmarks < 0 || !desc_begin || !T_WC1 || !(focal_length > 0.0) || !already_matched ||
      (n_kps > 0 && (!desc || !backproj || !use || !previous_landmark || !best_landmark || !best_dist || !hps_W || !hp_set)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised: bad argument");
  for (int l = 0; l < n_landmarks; ++l)
    if (desc_begin[l + 1] < desc_begin[l] || desc_begin[l] < 0)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised: desc_begin not monotone at %d", l);
  *already_matched = 0;
  if (n_kps == 0) return OKVFE_OK;
  const int n_pool = desc_begin[n_landmarks];
  if (n_pool > 0 && (!pool || !e0_W || !r0_W))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised: null pool");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_pair = take(sizeof(PairParams)), o_d = take((size_t)n_kps * 48), o_bp = take((size_t)n_kps * 24),
               o_u = take(n_kps), o_prev = take((size_t)n_kps * 4), o_b = take((size_t)(n_landmarks + 1) * 4),
               o_pool = take((size_t)n_pool * 48), o_e = take((size_t)n_pool * 24), o_r = take((size_t)n_pool * 24),
               o_lm = take((size_t)n_kps * 4), o_bd = take((size_t)n_kps * 4), o_hp = take((size_t)n_kps * 32),
               o_hs = take(n_kps), o_ctr = take(4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  PairParams pp{};
  std::memcpy(pp.C1, T_WC1->C, sizeof(pp.C1));
  std::memcpy(pp.r1, T_WC1->r, sizeof(pp.r1));
  const double sigma = 1.0 / focal_length;  // Frontend.cpp:1636
  pp.cos26 = std::cos(2.6 * sigma);
  pp.cos6 = std::cos(6.0 * sigma);
  auto up = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  };
  HIP_TRY(ctx, up(o_pair, &pp, sizeof(pp)));
  HIP_TRY(ctx, up(o_d, desc, (size_t)n_kps * 48));
  HIP_TRY(ctx, up(o_bp, backproj, (size_t)n_kps * 24));
  HIP_TRY(ctx, up(o_u, use, n_kps));
  HIP_TRY(ctx, up(o_prev, previous_landmark, (size_t)n_kps * 4));
  HIP_TRY(ctx, up(o_b, desc_begin, (size_t)(n_landmarks + 1) * 4));
  HIP_TRY(ctx, up(o_pool, pool, (size_t)n_pool * 48));
  HIP_TRY(ctx, up(o_e, e0_W, (size_t)n_pool * 24));
  HIP_TRY(ctx, up(o_r, r0_W, (size_t)n_pool * 24));
  HIP_TRY(ctx, hipMemsetAsync(base + o_ctr, 0, 4, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_match_to_map_uninit(reinterpret_cast<PairParams*>(base + o_pair), base + o_d,
                             reinterpret_cast<double*>(base + o_bp), base + o_u,
                             reinterpret_cast<int32_t*>(base + o_prev), n_kps, reinterpret_cast<int32_t*>(base + o_b),
                             n_landmarks, base + o_pool, reinterpret_cast<double*>(base + o_e),
                             reinterpret_cast<double*>(base + o_r), ctx->cfg.match_threshold,
                             reinterpret_cast<int32_t*>(base + o_lm), reinterpret_cast<int32_t*>(base + o_bd),
                             reinterpret_cast<double*>(base + o_hp), base + o_hs,
                             reinterpret_cast<int32_t*>(base + o_ctr), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(best_landmark, base + o_lm, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_bd, (size_t)n_kps * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(hps_W, base + o_hp, (size_t)n_kps * 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(hp_set, base + o_hs, n_kps, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(already_matched, base + o_ctr, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}
okvfe_status okvfe_verify_place_match(okvfe_ctx* ctx, const uint8_t* landmark_desc, const int32_t* desc_begin,
                                      int32_t n_landmarks, const uint8_t* frame_desc, int32_t n_kps,
                                      int32_t* k_min, uint32_t* dist_min) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_landmarks < 0 || n_kps < 0 || !desc_begin || (n_landmarks > 0 && (!k_min || !dist_min)) ||
      (n_kps > 0 && !frame_desc))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_match: bad argument");
  for (int l = 0; l < n_landmarks; ++l)
    if (desc_begin[l + 1] < desc_begin[l] || desc_begin[l] < 0)
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_match: desc_begin not monotone at %d", l);
  if (n_landmarks == 0) return OKVFE_OK;
  const int n_pool = desc_begin[n_landmarks];
  if (n_pool > 0 && !landmark_desc) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_match: null pool");
  const uint32_t thr = (uint32_t)ctx->cfg.match_threshold;
  if (n_kps == 0 || n_pool == 0) {  // Frontend.cpp:333-335: a camera without keypoints is skipped
    for (int l = 0; l < n_landmarks; ++l) { k_min[l] = 0; dist_min[l] = thr; }
    return OKVFE_OK;
  }
  if ((int64_t)3 * n_kps >= (int64_t)1 << 31) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "too many keypoints");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_pool = take((size_t)n_pool * 48), o_b = take((size_t)(n_landmarks + 1) * 4),
               o_f = take((size_t)n_kps * 48), o_k = take((size_t)n_landmarks * 4), o_d = take((size_t)n_landmarks * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_pool, landmark_desc, (size_t)n_pool * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_b, desc_begin, (size_t)(n_landmarks + 1) * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_f, frame_desc, (size_t)n_kps * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_verify_place(base + o_pool, reinterpret_cast<int32_t*>(base + o_b), n_landmarks, base + o_f, n_kps, thr,
                      reinterpret_cast<int32_t*>(base + o_k), reinterpret_cast<uint32_t*>(base + o_d), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(k_min, base + o_k, (size_t)n_landmarks * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(dist_min, base + o_d, (size_t)n_landmarks * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

okvfe_status okvfe_fbrisk_transform(okvfe_ctx* ctx, const uint8_t* descriptors, int32_t n,
                                    const uint8_t* node_descriptors, int32_t n_nodes, const int32_t* child_begin,
                                    const int32_t* child_index, const int32_t* node_word, int32_t* word_ids,
                                    int32_t* leaf_nodes) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n < 0 || n_nodes < 1 || !node_descriptors || !child_begin || !node_word || (n > 0 && (!descriptors || !word_ids)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: bad argument");
  // the tree must be a tree: children lists monotone, indices in range and pointing downwards
  if (child_begin[0] != 0) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: child_begin[0] != 0");
  for (int i = 0; i < n_nodes; ++i)
    if (child_begin[i + 1] < child_begin[i])
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: child_begin not monotone at %d", i);
  const int n_child = child_begin[n_nodes];
  if (n_child > 0 && !child_index) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: null children");
  {
    std::vector<int> depth(n_nodes, -1);
    depth[0] = 0;
    for (int i = 0; i < n_nodes; ++i)  // nodes are numbered so that a parent precedes its children
      for (int c = child_begin[i]; c < child_begin[i + 1]; ++c) {
        const int id = child_index[c];
        if (id <= i || id >= n_nodes || depth[i] < 0)
          return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_fbrisk_transform: node %d has child %d (not a tree in id order)", i, id);
        depth[id] = depth[i] + 1;
      }
  }
  if (n == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_d = take((size_t)n * 48), o_n = take((size_t)n_nodes * 48), o_cb = take((size_t)(n_nodes + 1) * 4),
               o_ci = take((size_t)n_child * 4), o_w = take((size_t)n_nodes * 4), o_wo = take((size_t)n * 4),
               o_no = take((size_t)n * 4);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_d, descriptors, (size_t)n * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_n, node_descriptors, (size_t)n_nodes * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_cb, child_begin, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice, s));
  if (n_child) HIP_TRY(ctx, hipMemcpyAsync(base + o_ci, child_index, (size_t)n_child * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_w, node_word, (size_t)n_nodes * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_voc_transform(base + o_d, n, base + o_n, n_nodes, reinterpret_cast<int32_t*>(base + o_cb),
                       reinterpret_cast<int32_t*>(base + o_ci), reinterpret_cast<int32_t*>(base + o_w),
                       reinterpret_cast<int32_t*>(base + o_wo), reinterpret_cast<int32_t*>(base + o_no), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(word_ids, base + o_wo, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  if (leaf_nodes) HIP_TRY(ctx, hipMemcpyAsync(leaf_nodes, base + o_no, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}
okvfe_status okvfe_bow_vector(const int32_t* word_ids, int32_t n_features, const double* word_weight,
                              int32_t n_words, int32_t weighting, int32_t normalise_l1, int32_t* ids_out,
                              double* values_out, int32_t cap, int32_t* n_out) {
  if (n_features < 0 || n_words < 1 || !word_weight || !n_out || weighting < 0 || weighting > 3 ||
      (n_features > 0 && !word_ids) || cap < 0 || (cap > 0 && (!ids_out || !values_out)))
    return OKVFE_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n_features; ++i)
    if (word_ids[i] < 0 || word_ids[i] >= n_words) return OKVFE_ERR_INVALID_ARGUMENT;
  // value per word in feature order (the sums are sequential additions of the same weight, as
  // BowVector::addWeight performs them), then the words in ascending order
  const bool sums = weighting == 0 || weighting == 1;  // TF_IDF, TF
  std::vector<double> acc(n_words, 0.0);
  std::vector<uint8_t> seen(n_words, 0);
  for (int i = 0; i < n_features; ++i) {
    const int id = word_ids[i];
    const double wgt = word_weight[id];
    if (!(wgt > 0)) continue;
    if (!seen[id]) {
      seen[id] = 1;
      acc[id] = wgt;
    } else if (sums) {
      acc[id] = acc[id] + wgt;
    }
  }
  int n = 0;
  for (int id = 0; id < n_words; ++id) n += seen[id];
  *n_out = n;
  if (n > cap) return OKVFE_ERR_CAPACITY;
  int k = 0;
  for (int id = 0; id < n_words; ++id)
    if (seen[id]) {
      ids_out[k] = id;
      values_out[k] = acc[id];
      ++k;
    }
  if (normalise_l1) {
    double norm = 0.0;
    for (int i = 0; i < n; ++i) norm = norm + std::fabs(values_out[i]);
    if (norm > 0.0)
      for (int i = 0; i < n; ++i) values_out[i] = values_out[i] / norm;
  } else if (sums && n > 0) {
    const double nd = (double)n;
    for (int i = 0; i < n; ++i) values_out[i] = values_out[i] / nd;
  }
  return OKVFE_OK;
}

okvfe_status okvfe_bow_query_l1(okvfe_ctx* ctx, const int32_t* db_begin, const int32_t* db_ids,
                                const double* db_values, int32_t n_entries, const int32_t* q_ids,
                                const double* q_values, int32_t n_q, double* scores) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n_entries < 0 || n_q < 0 || !db_begin || (n_entries > 0 && !scores) || (n_q > 0 && (!q_ids || !q_values)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: bad argument");
  if (db_begin[0] != 0) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: db_begin[0] != 0");
  for (int e = 0; e < n_entries; ++e) {
    if (db_begin[e + 1] < db_begin[e])
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: db_begin not monotone at %d", e);
    for (int i = db_begin[e] + 1; i < db_begin[e + 1]; ++i)
      if (db_ids[i] <= db_ids[i - 1])
        return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: entry %d is not in ascending word order", e);
  }
  for (int j = 1; j < n_q; ++j)
    if (q_ids[j] <= q_ids[j - 1])
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: query is not in ascending word order");
  if (n_entries == 0) return OKVFE_OK;
  const int m = db_begin[n_entries];
  if (m > 0 && (!db_ids || !db_values)) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_bow_query_l1: null database");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_b = take((size_t)(n_entries + 1) * 4), o_i = take((size_t)m * 4), o_v = take((size_t)m * 8),
               o_qi = take((size_t)n_q * 4), o_qv = take((size_t)n_q * 8), o_s = take((size_t)n_entries * 8);
  okvfe_status st = ensure_scratch(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_b, db_begin, (size_t)(n_entries + 1) * 4, hipMemcpyHostToDevice, s));
  if (m) {
    HIP_TRY(ctx, hipMemcpyAsync(base + o_i, db_ids, (size_t)m * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_v, db_values, (size_t)m * 8, hipMemcpyHostToDevice, s));
  }
  if (n_q) {
    HIP_TRY(ctx, hipMemcpyAsync(base + o_qi, q_ids, (size_t)n_q * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_qv, q_values, (size_t)n_q * 8, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));  // pageable host buffers: the copies above have completed
  launch_bow_query_l1(reinterpret_cast<int32_t*>(base + o_b), reinterpret_cast<int32_t*>(base + o_i),
                      reinterpret_cast<double*>(base + o_v), n_entries, reinterpret_cast<int32_t*>(base + o_qi),
                      reinterpret_cast<double*>(base + o_qv), n_q, reinterpret_cast<double*>(base + o_s), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(scores, base + o_s, (size_t)n_entries * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}
// ---- device-resident, batched map matchers (frame f = gather block f) --------------------------
namespace {
okvfe_status map_args_ok(okvfe_ctx* ctx, const char* who, const void* blocks, int n_frames, const okvfe_map_device* map) {
  if (!blocks || !map || n_frames < 1 || map->n_landmarks < 0 || !map->desc_begin ||
      (map->n_landmarks > 0 && !map->pool))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "%s: bad argument", who);
  return OKVFE_OK;
}
}  // namespace

okvfe_status okvfe_match_to_map_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                              const uint8_t* use_dev, const okvfe_map_device* map,
                                              double reprojection_threshold, int32_t* best_landmark_dev,
                                              int32_t* best_dist_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  okvfe_status st = map_args_ok(ctx, "okvfe_match_to_map_blocks_device", blocks_dev, n_frames, map);
  if (st != OKVFE_OK) return st;
  if (!best_landmark_dev || !best_dist_dev || !(reprojection_threshold >= 0.0) ||
      (map->n_landmarks > 0 && !map->projections))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  // workspace of the region order: one per stream, grown on demand (growing frees the old buffer, which
  // synchronises the device once)
  okvfe_ctx::MapPerm* mp = nullptr;
  for (auto& m : ctx->map_perm)
    if (m.stream == s) mp = &m;
  if (!mp) {
    // at most eight streams keep a workspace: an application that creates a stream per frame would otherwise leave one
    // behind per handle; the oldest entry is recycled (its stream is drained first: a launch may still read it)
    constexpr size_t kMaxPermStreams = 8;
    if (ctx->map_perm.size() >= kMaxPermStreams) {
      okvfe_ctx::MapPerm old = ctx->map_perm.front();
      ctx->map_perm.erase(ctx->map_perm.begin());
      HIP_TRY(ctx, hipDeviceSynchronize());
      if (old.d) HIP_TRY(ctx, hipFree(old.d));
    }
    ctx->map_perm.push_back(okvfe_ctx::MapPerm{s, nullptr, 0});
    mp = &ctx->map_perm.back();
  }
  if ((size_t)n_frames > mp->frames) {
    if (mp->d) HIP_TRY(ctx, hipFree(mp->d));
    mp->d = nullptr;
    mp->frames = 0;
    void* q = nullptr;
    HIP_TRY(ctx, hipMalloc(&q, (size_t)n_frames * ctx->kp_cap * sizeof(int32_t)));
    mp->d = static_cast<int32_t*>(q);
    mp->frames = (size_t)n_frames;
  }
  {
    StageTimer t(ctx, OKVFE_STAGE_MAP, s);
    launch_match_to_map_blocks(offs, static_cast<const uint8_t*>(blocks_dev), n_frames, ctx->kp_cap, use_dev,
                               map->projections, (size_t)map->n_landmarks * 2, map->desc_begin, map->n_landmarks,
                               map->pool, reprojection_threshold * reprojection_threshold, ctx->cfg.match_threshold,
                               best_landmark_dev, best_dist_dev, mp->d, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_match_to_map_uninitialised_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                                            const uint8_t* use_dev, const int32_t* previous_landmark_dev,
                                                            const okvfe_map_device* map, const okvfe_pose* T_WC1,
                                                            double focal_length, int32_t* best_landmark_dev,
                                                            int32_t* best_dist_dev, double* hps_W_dev, uint8_t* hp_set_dev,
                                                            int32_t* already_matched_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  okvfe_status st = map_args_ok(ctx, "okvfe_match_to_map_uninitialised_blocks_device", blocks_dev, n_frames, map);
  if (st != OKVFE_OK) return st;
  if (!T_WC1 || !(focal_length > 0.0) || !best_landmark_dev || !best_dist_dev || !hps_W_dev || !hp_set_dev ||
      !already_matched_dev || (map->n_landmarks > 0 && (!map->e0_W || !map->r0_W)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_to_map_uninitialised_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  // one pose record per frame, through the pinned parameter ring (one asynchronous copy, no host sync)
  std::vector<PairParams> pp((size_t)n_frames);
  const double sigma = 1.0 / focal_length;  // Frontend.cpp:1636
  const double c26 = std::cos(2.6 * sigma), c6 = std::cos(6.0 * sigma);
  for (int f = 0; f < n_frames; ++f) {
    pp[(size_t)f] = PairParams{};
    std::memcpy(pp[(size_t)f].C1, T_WC1[f].C, sizeof(pp[(size_t)f].C1));
    std::memcpy(pp[(size_t)f].r1, T_WC1[f].r, sizeof(pp[(size_t)f].r1));
    pp[(size_t)f].cos26 = c26;
    pp[(size_t)f].cos6 = c6;
  }
  void* d_pairs = nullptr;
  int slot = -1;
  st = ring_upload(ctx, &ctx->pair_ring, pp.data(), pp.size() * sizeof(PairParams), s, &d_pairs, &slot);
  if (st != OKVFE_OK) return st;
  hipError_t e = hipMemsetAsync(already_matched_dev, 0, (size_t)n_frames * sizeof(int32_t), s);
  if (e == hipSuccess) {
    StageTimer t(ctx, OKVFE_STAGE_MAP, s);
    launch_match_to_map_uninit_blocks(static_cast<const PairParams*>(d_pairs), offs,
                                      static_cast<const uint8_t*>(blocks_dev), n_frames, ctx->kp_cap, use_dev,
                                      previous_landmark_dev, map->desc_begin, map->n_landmarks, map->pool, map->e0_W,
                                      map->r0_W, ctx->cfg.match_threshold, best_landmark_dev, best_dist_dev, hps_W_dev,
                                      hp_set_dev, already_matched_dev, s);
    e = hipGetLastError();
  }
  const okvfe_status rel = ring_release(ctx, &ctx->pair_ring, slot, s);  // on every path: the slot has a reader or not
  HIP_TRY(ctx, e);
  ctx->last_stream = s;
  return rel;
}

okvfe_status okvfe_verify_place_blocks_device(okvfe_ctx* ctx, const void* blocks_dev, int32_t n_frames,
                                              const okvfe_map_device* map, int32_t* k_min_dev,
                                              uint32_t* dist_min_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  okvfe_status st = map_args_ok(ctx, "okvfe_verify_place_blocks_device", blocks_dev, n_frames, map);
  if (st != OKVFE_OK) return st;
  if (!k_min_dev || !dist_min_dev)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_verify_place_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  {
    StageTimer t(ctx, OKVFE_STAGE_MAP, s);
    launch_verify_place_blocks(map->pool, map->desc_begin, map->n_landmarks, offs,
                               static_cast<const uint8_t*>(blocks_dev), n_frames, ctx->kp_cap,
                               (uint32_t)ctx->cfg.match_threshold, k_min_dev, dist_min_dev, s);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}
}  // extern "C"

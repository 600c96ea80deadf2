// capi_match.cpp -- the Hamming matchers behind the C ABI: matchStereo (Frontend.cpp:2016-2076),
// matchMotionStereo (:1789-1905), candidate lists / arg-min, and the gather blocks that carry one
// image's results between GPUs and into the device-resident matchers.
#include "okvfe_ctx.h"

using namespace okvfe;

extern "C" {

okvfe_status okvfe_match_stereo_batch_device(okvfe_ctx* ctx, const okvfe_stereo_pair* pairs,
                                             int32_t n_pairs, okvfe_stereo_match* matches_dev,
                                             void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!pairs || !matches_dev || n_pairs < 1)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo_batch_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  std::vector<PairParams> pp(n_pairs);
  for (int i = 0; i < n_pairs; ++i) {
    if (pairs[i].image0 < 0 || pairs[i].image0 >= ctx->B || pairs[i].image1 < 0 || pairs[i].image1 >= ctx->B ||
        !(pairs[i].f0 > 0.0) || !(pairs[i].f1 > 0.0))
      return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "pair %d: image index or focal length out of range", i);
    pp[i] = to_pair_params(pairs[i]);
  }
  // pipelined lanes (okvfe_set_internal_lanes(-k)): the pairs of a slice are matched on the slice's lane stream, behind
  // its detect + describe chain, when every pair lies inside one slice and the slices' pairs are contiguous runs
  std::vector<int> lane_first;
  bool piped = ctx->lanes_pipelined && ctx->lanes_pending && ctx->n_layers == 1 && ctx->lane_chunk > 0 &&
               ctx->lanes_used > 1 && (int)ctx->lane_ctx.size() >= ctx->lanes_used && ctx->join_stream;
  if (piped) {
    lane_first.assign(ctx->lanes_used + 1, n_pairs);
    int cur = -1;
    for (int i = 0; i < n_pairs && piped; ++i) {
      const int l0 = pairs[i].image0 / ctx->lane_chunk, l1 = pairs[i].image1 / ctx->lane_chunk;
      if (l0 != l1 || l0 < cur || l0 >= ctx->lanes_used) piped = false;
      while (piped && cur < l0) lane_first[++cur] = i;
    }
    while (piped && cur < ctx->lanes_used - 1) lane_first[++cur] = n_pairs;
  }
  hipStream_t s = piped ? pick_stream_raw(ctx, stream) : pick_stream(ctx, stream);
  okvfe_status st;
  int cls_slot = -1;
  if (ctx->n_layers > 1) {
    // one table per distinct (f0, f1); usually one for the whole call
    std::vector<double> tables;
    std::vector<std::pair<double, double>> seen;
    std::vector<int> which(n_pairs);
    for (int i = 0; i < n_pairs; ++i) {
      int j = 0;
      for (; j < (int)seen.size(); ++j)
        if (seen[j].first == pairs[i].f0 && seen[j].second == pairs[i].f1) break;
      if (j == (int)seen.size()) {
        seen.emplace_back(pairs[i].f0, pairs[i].f1);
        tables.resize(tables.size() + kClassTableDoubles);
        fill_class_table(tables.data() + (size_t)j * kClassTableDoubles, pairs[i].f0, pairs[i].f1, false);
      }
      which[i] = j;
    }
    void* d_tab = nullptr;
    if ((st = ring_upload(ctx, &ctx->cls_ring, tables.data(), tables.size() * sizeof(double), s, &d_tab,
                          &cls_slot)) != OKVFE_OK)
      return st;
    for (int i = 0; i < n_pairs; ++i)
      pp[i].cls = static_cast<const double*>(d_tab) + (size_t)which[i] * kClassTableDoubles;
  }
  void* d_pairs = nullptr;
  int slot = -1;
  st = ring_upload(ctx, &ctx->pair_ring, pp.data(), (size_t)n_pairs * sizeof(PairParams), s, &d_pairs, &slot);
  if (st != OKVFE_OK) return st;
  if (piped) {
    HIP_TRY(ctx, hipEventRecord(ctx->lane_fork, s));  // behind the pair upload
    for (int l = 0; l < ctx->lanes_used; ++l) {
      const int first = lane_first[l], n = lane_first[l + 1] - first;
      hipStream_t ls = ctx->lane_ctx[l]->stream;
      HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->lane_fork, 0));
      if (n > 0) {
        StageTimer t(ctx, OKVFE_STAGE_MATCH, ls);
        launch_match_stereo(static_cast<const PairParams*>(d_pairs) + first, n, ctx->d_kps, ctx->d_desc, ctx->d_bp,
                            ctx->d_bpv, ctx->d_count, ctx->kp_cap, ctx->cfg.match_threshold,
                            matches_dev + (size_t)first * ctx->kp_cap, ls);
      }
      HIP_TRY(ctx, hipEventRecord(ctx->lane_done[l], ls));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->join_stream, ctx->lane_done[l], 0));
    }
    if ((st = ring_release(ctx, &ctx->pair_ring, slot, ctx->join_stream)) != OKVFE_OK) return st;
    HIP_TRY(ctx, hipEventRecord(ctx->join_done, ctx->join_stream));
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_stream = s;
    return OKVFE_OK;
  }
  {
    StageTimer t(ctx, OKVFE_STAGE_MATCH, s);
    launch_match_stereo(static_cast<const PairParams*>(d_pairs), n_pairs, ctx->d_kps, ctx->d_desc, ctx->d_bp,
                        ctx->d_bpv, ctx->d_count, ctx->kp_cap, ctx->cfg.match_threshold, matches_dev, s);
  }
  if ((st = ring_release(ctx, &ctx->pair_ring, slot, s)) != OKVFE_OK) return st;
  if ((st = ring_release(ctx, &ctx->cls_ring, cls_slot, s)) != OKVFE_OK) return st;
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_match_stereo(okvfe_ctx* ctx, const uint8_t* desc0, const okvfe_keypoint* kp0,
                                const double* backproj0, const uint8_t* valid0, int32_t n0,
                                const uint8_t* desc1, const okvfe_keypoint* kp1, const double* backproj1,
                                const uint8_t* valid1, int32_t n1, const okvfe_pose* T_WC0,
                                const okvfe_pose* T_WC1, double f0, double f1, okvfe_stereo_match* matches) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (n0 < 0 || n1 < 0 || !T_WC0 || !T_WC1 || !(f0 > 0.0) || !(f1 > 0.0) ||
      (n0 > 0 && (!desc0 || !backproj0 || !valid0 || !matches)) || (n1 > 0 && (!desc1 || !backproj1 || !valid1)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo: bad argument");
  // keypoint sizes select the triangulation sigma (Frontend.cpp:2031-2035): sizes must be
  // 12 * scale(octave); only a scale-space detector produces anything but 12
  bool multi = false;
  okvfe_status st = check_size_classes(ctx, kp0, n0, &multi);
  if (st == OKVFE_OK) st = check_size_classes(ctx, kp1, n1, &multi);
  if (st != OKVFE_OK) return st;
  if (multi && (!kp0 || !kp1)) return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo: keypoints needed");
  if (n0 == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), 256); return o; };
  const size_t o_pair = take(sizeof(PairParams)), o_cls = take(kClassTableDoubles * sizeof(double));
  const size_t o_d0 = take((size_t)n0 * 48), o_b0 = take((size_t)n0 * 24), o_v0 = take(n0),
               o_k0 = take((size_t)n0 * sizeof(okvfe_keypoint));
  const size_t o_d1 = take((size_t)n1 * 48), o_b1 = take((size_t)n1 * 24), o_v1 = take(n1),
               o_k1 = take((size_t)n1 * sizeof(okvfe_keypoint));
  const size_t o_out = take((size_t)n0 * sizeof(okvfe_stereo_match));
  if ((st = ensure_scratch(ctx, off)) != OKVFE_OK) return st;
  if ((st = ensure_pinned(ctx, off)) != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  uint8_t* hb = ctx->h_pinned;  // the same layout in pinned host memory
  okvfe_stereo_pair sp{};
  sp.image0 = 0; sp.image1 = 0; sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1; sp.f0 = f0; sp.f1 = f1;
  PairParams pp = to_pair_params(sp);
  if (multi) {
    fill_class_table(reinterpret_cast<double*>(hb + o_cls), f0, f1, false);
    pp.cls = reinterpret_cast<const double*>(base + o_cls);
    std::memcpy(hb + o_k0, kp0, (size_t)n0 * sizeof(okvfe_keypoint));
    std::memcpy(hb + o_k1, kp1, (size_t)n1 * sizeof(okvfe_keypoint));
  }
  std::memcpy(hb + o_pair, &pp, sizeof(pp));
  std::memcpy(hb + o_d0, desc0, (size_t)n0 * 48);
  std::memcpy(hb + o_b0, backproj0, (size_t)n0 * 24);
  std::memcpy(hb + o_v0, valid0, (size_t)n0);
  if (n1 > 0) {
    std::memcpy(hb + o_d1, desc1, (size_t)n1 * 48);
    std::memcpy(hb + o_b1, backproj1, (size_t)n1 * 24);
    std::memcpy(hb + o_v1, valid1, (size_t)n1);
  }
  // One kernel moves every input (it reads the pinned block in place), the matcher writes its rows
  // straight into the pinned block, one synchronisation: a frame's matchStereo was seven pageable
  // host-to-device copies, a synchronisation, the kernel, a copy back and another synchronisation
  // (75 us for a 17 us kernel at B = 1)
  okvfe_stereo_match* out_dev = reinterpret_cast<okvfe_stereo_match*>(base + o_out);
  if (ctx->h_pinned_dev) {
    launch_param_copy(base, ctx->h_pinned_dev, o_out, nullptr, 0, s);
    out_dev = reinterpret_cast<okvfe_stereo_match*>(static_cast<uint8_t*>(ctx->h_pinned_dev) + o_out);
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(base, hb, o_out, hipMemcpyHostToDevice, s));
  }
  launch_match_stereo_arrays(reinterpret_cast<PairParams*>(base + o_pair), base + o_d0,
                             reinterpret_cast<double*>(base + o_b0), base + o_v0, nullptr, n0, base + o_d1,
                             reinterpret_cast<double*>(base + o_b1), base + o_v1, nullptr, n1, n0,
                             ctx->cfg.match_threshold, out_dev, s,
                             multi ? reinterpret_cast<const okvfe_keypoint*>(base + o_k0) : nullptr,
                             multi ? reinterpret_cast<const okvfe_keypoint*>(base + o_k1) : nullptr);
  HIP_TRY(ctx, hipGetLastError());
  if (!ctx->h_pinned_dev)
    HIP_TRY(ctx, hipMemcpyAsync(hb + o_out, base + o_out, (size_t)n0 * sizeof(okvfe_stereo_match), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  std::memcpy(matches, hb + o_out, (size_t)n0 * sizeof(okvfe_stereo_match));
  return OKVFE_OK;
}

okvfe_status okvfe_match_motion_stereo(okvfe_ctx* ctx, const okvfe_camera* camera, const uint8_t* desc0,
                                       const okvfe_keypoint* kp0, const double* backproj0, const uint8_t* valid0,
                                       const uint8_t* skip0, int32_t n0, const uint8_t* desc1,
                                       const okvfe_keypoint* kp1, const double* backproj1, const uint8_t* valid1,
                                       const uint8_t* matched1, int32_t n1, const okvfe_pose* T_WC0,
                                       const okvfe_pose* T_WC1, okvfe_motion_match* matches) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!camera || n0 < 0 || n1 < 0 || !T_WC0 || !T_WC1 ||
      (n0 > 0 && (!desc0 || !kp0 || !backproj0 || !valid0 || !matches)) ||
      (n1 > 0 && (!desc1 || !kp1 || !backproj1 || !valid1)))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_motion_stereo: bad argument");
  bool multi = false;
  {
    okvfe_status cst = check_size_classes(ctx, kp0, n0, &multi);
    if (cst == OKVFE_OK) cst = check_size_classes(ctx, kp1, n1, &multi);
    if (cst != OKVFE_OK) return cst;
  }
  if (n0 == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const size_t a = 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + std::max<size_t>(bytes, 1), a); return o; };
  const size_t o_pair = take(sizeof(PairParams)), o_cam = take(sizeof(DeviceCamera)),
               o_cls = take(kClassTableDoubles * sizeof(double));
  const size_t o_d0 = take((size_t)n0 * 48), o_k0 = take((size_t)n0 * sizeof(okvfe_keypoint)), o_b0 = take((size_t)n0 * 24),
               o_v0 = take(n0), o_s0 = take(n0);
  const size_t o_d1 = take((size_t)n1 * 48), o_k1 = take((size_t)n1 * sizeof(okvfe_keypoint)), o_b1 = take((size_t)n1 * 24),
               o_v1 = take(n1), o_m1 = take(n1);
  const size_t o_out = take((size_t)n0 * sizeof(okvfe_motion_match));
  okvfe_status st = ensure_scratch(ctx, off);
  if (st == OKVFE_OK) st = ensure_pinned(ctx, off);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  uint8_t* hb = ctx->h_pinned;  // the same layout in pinned host memory (see okvfe_match_stereo)
  okvfe_stereo_pair sp{};
  sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1;
  sp.f0 = sp.f1 = 0.5 * (camera->fu + camera->fv);  // sigma = size0 / f0 * 0.125 (Frontend.cpp:1834)
  PairParams pp = to_pair_params(sp);
  const DeviceCamera dc = to_device_camera(*camera);
  auto put = [&](size_t o, const void* src, size_t bytes) {
    if (bytes) std::memcpy(hb + o, src, bytes);
  };
  if (multi) {
    fill_class_table(reinterpret_cast<double*>(hb + o_cls), sp.f0, sp.f1, true);
    pp.cls = reinterpret_cast<const double*>(base + o_cls);
  }
  put(o_pair, &pp, sizeof(pp));
  put(o_cam, &dc, sizeof(dc));
  put(o_d0, desc0, (size_t)n0 * 48);
  put(o_k0, kp0, (size_t)n0 * sizeof(okvfe_keypoint));
  put(o_b0, backproj0, (size_t)n0 * 24);
  put(o_v0, valid0, n0);
  if (skip0) put(o_s0, skip0, n0);
  if (n1 > 0) {
    put(o_d1, desc1, (size_t)n1 * 48);
    put(o_k1, kp1, (size_t)n1 * sizeof(okvfe_keypoint));
    put(o_b1, backproj1, (size_t)n1 * 24);
    put(o_v1, valid1, n1);
    if (matched1) put(o_m1, matched1, n1);
  }
  okvfe_motion_match* out_dev = reinterpret_cast<okvfe_motion_match*>(base + o_out);
  if (ctx->h_pinned_dev) {
    launch_param_copy(base, ctx->h_pinned_dev, o_out, nullptr, 0, s);
    out_dev = reinterpret_cast<okvfe_motion_match*>(static_cast<uint8_t*>(ctx->h_pinned_dev) + o_out);
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(base, hb, o_out, hipMemcpyHostToDevice, s));
  }
  launch_match_motion(reinterpret_cast<PairParams*>(base + o_pair), reinterpret_cast<DeviceCamera*>(base + o_cam),
                      camera->width, camera->height, base + o_d0, reinterpret_cast<okvfe_keypoint*>(base + o_k0),
                      reinterpret_cast<double*>(base + o_b0), base + o_v0, skip0 ? base + o_s0 : nullptr, n0,
                      base + o_d1, reinterpret_cast<okvfe_keypoint*>(base + o_k1),
                      reinterpret_cast<double*>(base + o_b1), base + o_v1, matched1 ? base + o_m1 : nullptr, n1,
                      ctx->cfg.match_threshold, out_dev, s);
  HIP_TRY(ctx, hipGetLastError());
  if (!ctx->h_pinned_dev)
    HIP_TRY(ctx, hipMemcpyAsync(hb + o_out, base + o_out, (size_t)n0 * sizeof(okvfe_motion_match), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  std::memcpy(matches, hb + o_out, (size_t)n0 * sizeof(okvfe_motion_match));
  return OKVFE_OK;
}
okvfe_status okvfe_hamming_candidates(okvfe_ctx* ctx, const uint8_t* A, int32_t nA, const uint8_t* B,
                                      int32_t nB, int32_t threshold, okvfe_candidate* out, int32_t cap,
                                      int32_t* n_out) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (nA < 0 || nB < 0 || !n_out || cap < 0 || (nA > 0 && !A) || (nB > 0 && !B) || (cap > 0 && !out))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_hamming_candidates: bad argument");
  *n_out = 0;
  if (nA == 0 || nB == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const size_t a = 256;
  const size_t o_A = 0;
  const size_t o_B = align_up(o_A + (size_t)nA * 48, a);
  const size_t o_rows = align_up(o_B + (size_t)nB * 48, a);
  const size_t o_out = align_up(o_rows + (size_t)nA * 4, a);
  const size_t total = o_out + (size_t)std::max(cap, 1) * sizeof(okvfe_candidate);
  okvfe_status st = ensure_scratch(ctx, total);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_A, A, (size_t)nA * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(base + o_B, B, (size_t)nB * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  int32_t* d_rows = reinterpret_cast<int32_t*>(base + o_rows);
  launch_hamming_count(base + o_A, nA, base + o_B, nB, threshold, d_rows, s);
  std::vector<int32_t> rows(nA);
  HIP_TRY(ctx, hipMemcpyAsync(rows.data(), d_rows, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  int64_t total_c = 0;
  for (int i = 0; i < nA; ++i) {
    const int32_t c = rows[i];
    rows[i] = (int32_t)std::min<int64_t>(total_c, INT32_MAX);
    total_c += c;
  }
  *n_out = (int32_t)std::min<int64_t>(total_c, INT32_MAX);
  HIP_TRY(ctx, hipMemcpyAsync(d_rows, rows.data(), (size_t)nA * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_hamming_emit(base + o_A, nA, base + o_B, nB, threshold, d_rows,
                      reinterpret_cast<okvfe_candidate*>(base + o_out), cap, s);
  HIP_TRY(ctx, hipGetLastError());
  const int32_t ncopy = (int32_t)std::min<int64_t>(total_c, cap);
  if (ncopy > 0)
    HIP_TRY(ctx, hipMemcpyAsync(out, base + o_out, (size_t)ncopy * sizeof(okvfe_candidate), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (total_c > cap) return fail(ctx, OKVFE_ERR_CAPACITY, "%lld candidates, caller capacity %d", (long long)total_c, cap);
  return OKVFE_OK;
}

okvfe_status okvfe_hamming_argmin(okvfe_ctx* ctx, const uint8_t* A, int32_t nA, const uint8_t* B, int32_t nB,
                                  uint32_t threshold, int32_t* best_j, uint32_t* best_dist) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (nA < 0 || nB < 0 || (nA > 0 && (!A || !best_j || !best_dist)) || (nB > 0 && !B))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_hamming_argmin: bad argument");
  if (nA == 0) return OKVFE_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = ctx->stream;
  const size_t a = 256;
  const size_t o_A = 0;
  const size_t o_B = align_up(o_A + (size_t)nA * 48, a);
  const size_t o_j = align_up(o_B + (size_t)std::max(nB, 1) * 48, a);
  const size_t o_d = align_up(o_j + (size_t)nA * 4, a);
  const size_t total = o_d + (size_t)nA * 4;
  okvfe_status st = ensure_scratch(ctx, total);
  if (st != OKVFE_OK) return st;
  uint8_t* base = static_cast<uint8_t*>(ctx->scratch);
  HIP_TRY(ctx, hipMemcpyAsync(base + o_A, A, (size_t)nA * 48, hipMemcpyHostToDevice, s));
  if (nB > 0) HIP_TRY(ctx, hipMemcpyAsync(base + o_B, B, (size_t)nB * 48, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  launch_hamming_argmin(base + o_A, nA, base + o_B, nB, threshold, reinterpret_cast<int32_t*>(base + o_j),
                        reinterpret_cast<uint32_t*>(base + o_d), s);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(best_j, base + o_j, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(best_dist, base + o_d, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return OKVFE_OK;
}

size_t okvfe_gather_block_bytes(const okvfe_ctx* ctx) { return ctx ? block_layout(ctx->kp_cap).total : 0; }

okvfe_status okvfe_pack_gather_blocks_device(okvfe_ctx* ctx, int32_t first_index, int32_t n, void* blocks_dev,
                                             void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (first_index < 0 || n < 1 || first_index + n > ctx->B || !blocks_dev)
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_pack_gather_blocks_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  launch_pack_blocks(offs, first_index, n, ctx->kp_cap, ctx->d_count, ctx->d_kps, ctx->d_desc, ctx->d_bp,
                     ctx->d_bpv, static_cast<uint8_t*>(blocks_dev), s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return OKVFE_OK;
}

okvfe_status okvfe_pack_gather_block_device(okvfe_ctx* ctx, int32_t index, void* block_dev, void* stream) {
  return okvfe_pack_gather_blocks_device(ctx, index, 1, block_dev, stream);
}

okvfe_status okvfe_match_stereo_blocks_batch_device(okvfe_ctx* ctx, const void* blocks0_dev,
                                                    const void* blocks1_dev, int32_t n_frames,
                                                    const okvfe_pose* T_WC0, const okvfe_pose* T_WC1, double f0,
                                                    double f1, okvfe_stereo_match* matches_dev, void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!blocks0_dev || !blocks1_dev || n_frames < 1 || !T_WC0 || !T_WC1 || !matches_dev || !(f0 > 0.0) || !(f1 > 0.0))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_stereo_blocks_batch_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  okvfe_stereo_pair sp{};
  sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1; sp.f0 = f0; sp.f1 = f1;
  PairParams pp = to_pair_params(sp);
  int cls_slot = -1;
  if (ctx->n_layers > 1) {
    double table[kClassTableDoubles];
    fill_class_table(table, f0, f1, false);
    void* d_tab = nullptr;
    okvfe_status st = ring_upload(ctx, &ctx->cls_ring, table, sizeof(table), s, &d_tab, &cls_slot);
    if (st != OKVFE_OK) return st;
    pp.cls = static_cast<const double*>(d_tab);
  }
  // the pair record travels by value as a kernel argument: nothing to keep alive
  launch_match_stereo_blocks(pp, offs, static_cast<const uint8_t*>(blocks0_dev),
                             static_cast<const uint8_t*>(blocks1_dev), n_frames, ctx->kp_cap,
                             ctx->cfg.match_threshold, matches_dev, s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return ring_release(ctx, &ctx->cls_ring, cls_slot, s);
}
okvfe_status okvfe_match_motion_stereo_blocks_device(okvfe_ctx* ctx, int32_t cam, const void* block0_dev,
                                                     const void* block1_dev, const uint8_t* skip0_dev,
                                                     const uint8_t* matched1_dev, const okvfe_pose* T_WC0,
                                                     const okvfe_pose* T_WC1, okvfe_motion_match* matches_dev,
                                                     void* stream) {
  if (!ctx) return OKVFE_ERR_INVALID_ARGUMENT;
  if (!block0_dev || !block1_dev || !T_WC0 || !T_WC1 || !matches_dev || cam < 0 ||
      cam >= (int)ctx->h_cams.size())
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "okvfe_match_motion_stereo_blocks_device: bad argument");
  const DeviceCamera& dc = ctx->h_cams[cam];
  if (!(dc.fu > 0.0))
    return fail(ctx, OKVFE_ERR_INVALID_ARGUMENT, "camera slot %d has no intrinsics (okvfe_set_camera)", cam);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t s = pick_stream(ctx, stream);
  const BlockLayout L = block_layout(ctx->kp_cap);
  const int offs[6] = {(int)L.o_count, (int)L.o_kps, (int)L.o_desc, (int)L.o_bp, (int)L.o_bpv, (int)L.total};
  okvfe_stereo_pair sp{};
  sp.T_WC0 = *T_WC0; sp.T_WC1 = *T_WC1;
  sp.f0 = sp.f1 = 0.5 * (dc.fu + dc.fv);  // sigma = size0 / f0 * 0.125 (Frontend.cpp:1834)
  PairParams pp = to_pair_params(sp);
  int cls_slot = -1;
  if (ctx->n_layers > 1) {
    double table[kClassTableDoubles];
    fill_class_table(table, sp.f0, sp.f1, true);
    void* d_tab = nullptr;
    okvfe_status st = ring_upload(ctx, &ctx->cls_ring, table, sizeof(table), s, &d_tab, &cls_slot);
    if (st != OKVFE_OK) return st;
    pp.cls = static_cast<const double*>(d_tab);
  }
  launch_match_motion_blocks(pp, ctx->d_cams + cam, ctx->w, ctx->h, offs,
                             static_cast<const uint8_t*>(block0_dev), static_cast<const uint8_t*>(block1_dev),
                             skip0_dev, matched1_dev, ctx->kp_cap, ctx->cfg.match_threshold, matches_dev, s);
  HIP_TRY(ctx, hipGetLastError());
  ctx->last_stream = s;
  return ring_release(ctx, &ctx->cls_ring, cls_slot, s);
}

okvfe_status okvfe_match_stereo_blocks_device(okvfe_ctx* ctx, const void* block0_dev, const void* block1_dev,
                                              const okvfe_pose* T_WC0, const okvfe_pose* T_WC1, double f0,
                                              double f1, okvfe_stereo_match* matches_dev, void* stream) {
  return okvfe_match_stereo_blocks_batch_device(ctx, block0_dev, block1_dev, 1, T_WC0, T_WC1, f0, f1, matches_dev,
                                                stream);
}

}  // extern "C"

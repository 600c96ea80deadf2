// wire_format.cpp -- host-side data formats either side of the hot path (SURVEY.md §8 f, N2):
//
//   okvfe_format_keypoint_lines / okvfe_parse_keypoint_lines
//       the per-keypoint text records of OKVIS2 map files,
//       "FRAME:KEYPOINT <stateId> <cameraIdx> <x> <y> <size> BRISK2 <96 hex digits>"
//       written by okvis::Component::save (okvis_ceres/src/Component.cpp:448-460, stream
//       precision 17 from :409) and read back by Component::load (:235-256), which then injects
//       them through MultiFrame::resetKeypoints / resetDescriptors (:260-265).  Lets front-end
//       output feed an unmodified back-end and lets saved maps serve as real-data fixtures.
//   okvfe_fbrisk_mean
//       DBoW2::FBrisk::meanValue (okvis_frontend/src/FBrisk.cpp:25-58): bitwise majority of
//       48-byte descriptors (bit set iff count > n/2).  FBrisk::distance is okvfe_popcnt_xor.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string_view>
#include <vector>

#include "../../include/okvfe.h"

extern "C" {

okvfe_status okvfe_format_keypoint_lines(uint64_t state_id, uint64_t camera_idx,
                                         const okvfe_keypoint* keypoints, const uint8_t* descriptors,
                                         int32_t n, char* out, size_t cap, size_t* written) {
  if (n < 0 || !written || (n > 0 && (!keypoints || !descriptors))) return OKVFE_ERR_INVALID_ARGUMENT;
  size_t pos = 0;
  bool fits = true;
  for (int k = 0; k < n; ++k) {
    char line[256];
    // operator<<(float) at precision 17 prints the float widened to double with %.17g
    int len = std::snprintf(line, sizeof(line), "FRAME:KEYPOINT %llu %llu %.17g %.17g %.17g BRISK2 ",
                            (unsigned long long)state_id, (unsigned long long)camera_idx,
                            (double)keypoints[k].x, (double)keypoints[k].y, (double)keypoints[k].size);
    const uint8_t* d = descriptors + (size_t)k * OKVFE_DESC_BYTES;
    for (int i = 0; i < OKVFE_DESC_BYTES; ++i) len += std::snprintf(line + len, sizeof(line) - len, "%02x", d[i]);
    line[len++] = '\n';
    if (out && pos + (size_t)len <= cap)
      std::memcpy(out + pos, line, (size_t)len);
    else
      fits = false;
    pos += (size_t)len;
  }
  *written = pos;  // bytes needed, even when they did not fit
  return (out && !fits) ? OKVFE_ERR_CAPACITY : OKVFE_OK;  // out == NULL: size query
}

okvfe_status okvfe_parse_keypoint_lines(const char* text, size_t len, uint64_t* state_id,
                                        uint64_t* camera_idx, okvfe_keypoint* keypoints,
                                        uint8_t* descriptors, int32_t cap, int32_t* n_out,
                                        size_t* consumed) {
  if (!text || !n_out || cap < 0) return OKVFE_ERR_INVALID_ARGUMENT;
  int n = 0;
  size_t pos = 0;
  uint64_t sid = 0, cam = 0;
  okvfe_status st = OKVFE_OK;
  while (pos < len) {
    size_t eol = pos;
    while (eol < len && text[eol] != '\n') ++eol;
    const std::string_view line(text + pos, eol - pos);
    // Component::load stops the keypoint block at the first line of another type (:257-270)
    if (line.substr(0, 15) != "FRAME:KEYPOINT ") break;
    char buf[320];
    if (line.size() >= sizeof(buf)) return OKVFE_ERR_INVALID_ARGUMENT;
    std::memcpy(buf, line.data(), line.size());
    buf[line.size()] = 0;
    unsigned long long s = 0, c = 0;
    float x = 0, y = 0, size = 0;
    char kind[16] = {0}, hex[128] = {0};
    if (std::sscanf(buf + 15, "%llu %llu %f %f %f %15s %127s", &s, &c, &x, &y, &size, kind, hex) != 7)
      return OKVFE_ERR_INVALID_ARGUMENT;
    if (std::strcmp(kind, "BRISK2") != 0) return OKVFE_ERR_UNSUPPORTED;  // "only BRISK 2" (:243-245)
    if (std::strlen(hex) != 2 * OKVFE_DESC_BYTES) return OKVFE_ERR_INVALID_ARGUMENT;
    if (n == 0) {
      sid = s;
      cam = c;
    } else if (s != sid || c != cam) {
      break;  // next frame / camera: the caller parses it with another call
    }
    if (n < cap) {
      if (keypoints) {
        okvfe_keypoint& kp = keypoints[n];
        kp.x = x; kp.y = y; kp.size = size;
        kp.angle = -1.0f; kp.response = 0.0f; kp.octave = 0; kp.class_id = -1;  // cv::KeyPoint defaults
      }
      if (descriptors) {
        for (int i = 0; i < OKVFE_DESC_BYTES; ++i) {
          const char two[3] = {hex[2 * i], hex[2 * i + 1], 0};
          char* end = nullptr;
          const unsigned long v = std::strtoul(two, &end, 16);
          if (end != two + 2) return OKVFE_ERR_INVALID_ARGUMENT;
          descriptors[(size_t)n * OKVFE_DESC_BYTES + i] = (uint8_t)v;
        }
      }
    } else {
      st = OKVFE_ERR_CAPACITY;
    }
    ++n;
    pos = eol < len ? eol + 1 : eol;
  }
  if (state_id) *state_id = sid;
  if (camera_idx) *camera_idx = cam;
  if (consumed) *consumed = pos;
  *n_out = n;
  return st;
}

okvfe_status okvfe_fbrisk_mean(const uint8_t* descriptors, int32_t n, uint8_t* mean) {
  if (n < 0 || !mean || (n > 0 && !descriptors)) return OKVFE_ERR_INVALID_ARGUMENT;
  std::vector<uint64_t> sum(OKVFE_DESC_BYTES * 8, 0);
  for (int k = 0; k < n; ++k)
    for (int i = 0; i < OKVFE_DESC_BYTES; ++i)
      for (int b = 0; b < 8; ++b) sum[i * 8 + b] += (descriptors[(size_t)k * OKVFE_DESC_BYTES + i] >> b) & 1u;
  const uint64_t s = (uint64_t)n / 2;
  for (int i = 0; i < OKVFE_DESC_BYTES; ++i) {
    uint8_t m = 0;
    for (int b = 0; b < 8; ++b)
      if (sum[i * 8 + b] > s) m |= (uint8_t)(1u << b);
    mean[i] = m;
  }
  return OKVFE_OK;
}

}  // extern "C"

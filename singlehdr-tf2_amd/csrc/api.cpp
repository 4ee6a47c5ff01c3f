// libshdr: error state and version (host only).
#include <stdarg.h>
#include <stdio.h>

#include "shdr_internal.h"

namespace shdr {
static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}
}  // namespace shdr

namespace shdr {
int g_env_epoch = 0;
}
extern "C" void shdr_config_reload(void) { __atomic_add_fetch(&shdr::g_env_epoch, 1, __ATOMIC_ACQ_REL); }

extern "C" const char* shdr_last_error(void) { return shdr::g_last_error; }
extern "C" const char* shdr_version(void) { return "libshdr 0.1 gfx950"; }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8: the checksum of TensorFlow's tensor-bundle
// checkpoint format (.index blocks and tensor payloads; tf_utils.py:149-169 writes such checkpoints).  Host only.
namespace {
struct Crc32cTable {
  uint32_t t[8][256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
}  // namespace

extern "C" uint32_t shdr_crc32c(const void* data, uint64_t n, uint32_t crc) {
  static const Crc32cTable tab;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {
    const uint32_t lo = c ^ ((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24);
    c = tab.t[7][lo & 0xff] ^ tab.t[6][(lo >> 8) & 0xff] ^ tab.t[5][(lo >> 16) & 0xff] ^ tab.t[4][lo >> 24] ^
        tab.t[3][p[4]] ^ tab.t[2][p[5]] ^ tab.t[1][p[6]] ^ tab.t[0][p[7]];
    p += 8;
    n -= 8;
  }
  while (n--) c = tab.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return ~c;
}

// Radiance .hdr scanline run-length encoding ("new" adaptive RLE, each of the 4 RGBE components of a scanline coded
// separately: a count byte > 128 repeats the next byte (count - 128) times, a count <= 128 copies that many literal
// bytes).  This is the on-disk form cv2.imwrite("*.hdr") produces (test_real_refinement.py:150).  Host only.
namespace {
int64_t rle_component(const uint8_t* data, int n, uint8_t* out) {   // data stride 4 (one RGBE component)
  constexpr int kMinRun = 4;
  int64_t w = 0;
  int cur = 0;
  while (cur < n) {
    int run_start = cur, run_len = 0, prev_len = 0;
    while (run_len < kMinRun && run_start < n) {       // find the next run of at least kMinRun equal bytes
      run_start += run_len;
      prev_len = run_len;
      run_len = 1;
      while (run_start + run_len < n && run_len < 127 && data[4 * run_start] == data[4 * (run_start + run_len)]) ++run_len;
    }
    if (prev_len > 1 && prev_len == run_start - cur) {   // the bytes before it are themselves one short run
      out[w++] = (uint8_t)(128 + prev_len);
      out[w++] = data[4 * cur];
      cur = run_start;
    }
    while (cur < run_start) {                            // literals up to the run
      int lit = run_start - cur;
      if (lit > 128) lit = 128;
      out[w++] = (uint8_t)lit;
      for (int i = 0; i < lit; ++i) out[w++] = data[4 * (cur + i)];
      cur += lit;
    }
    if (run_len >= kMinRun) {
      out[w++] = (uint8_t)(128 + run_len);
      out[w++] = data[4 * run_start];
      cur += run_len;
    }
  }
  return w;
}
}  // namespace

extern "C" int64_t shdr_rgbe_rle_encode(const uint8_t* rgbe, int width, int height, uint8_t* out, int64_t capacity) {
  if (!rgbe || !out || width <= 0 || height <= 0) {
    shdr::set_error("rgbe_rle_encode: bad arguments");
    return -1;
  }
  const bool rle = width >= 8 && width <= 32767;         // outside this range the format stores flat pixels
  const int64_t per_line = rle ? 4 + 4 * ((int64_t)width + width / 127 + 2) : 4 * (int64_t)width;
  if (capacity < per_line * height) {
    shdr::set_error("rgbe_rle_encode: output buffer too small (%lld < %lld)", (long long)capacity, (long long)(per_line * height));
    return -1;
  }
  int64_t w = 0;
  for (int y = 0; y < height; ++y) {
    const uint8_t* line = rgbe + (int64_t)y * width * 4;
    if (!rle) {
      for (int64_t i = 0; i < 4 * (int64_t)width; ++i) out[w++] = line[i];
      continue;
    }
    out[w++] = 2; out[w++] = 2; out[w++] = (uint8_t)(width >> 8); out[w++] = (uint8_t)(width & 255);
    for (int c = 0; c < 4; ++c) w += rle_component(line + c, width, out + w);
  }
  return w;
}

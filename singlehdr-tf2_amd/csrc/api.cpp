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

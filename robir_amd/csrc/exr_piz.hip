// Host-side decoder for PIZ-compressed OpenEXR chunks (the reference reads its relighting environment maps with
// imageio/FreeImage: EnvmapMaterialNetwork.load_light, model/sg_envmap_material.py:266-268; two of its three shipped maps
// are PIZ).  PIZ = value-range bitmap + canonical Huffman coding with run lengths + a 2-D Haar-like wavelet on 16-bit
// words; restated from the published OpenEXR format description (OpenEXR 2.x "PIZ compression": hufUncompress,
// wav2Decode, reverse LUT).  No GPU work here -- this file only rides in the same library so that the Python loader
// (robir_amd/exr.py) does not decode 1.5 M Huffman symbols in the interpreter.
#include "../../include/robir_hip.h"
#include "common.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int HUF_ENCBITS = 16, HUF_DECBITS = 14;
constexpr int HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1;   // symbols 0..65535 + the run-length symbol
constexpr int HUF_DECSIZE = 1 << HUF_DECBITS;
constexpr int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;
constexpr int BITMAP_SIZE = 8192;

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  unsigned __int128 c = 0;   // up to 58 + 7 buffered bits
  int lc = 0;
  bool fill(int need) {   // most-significant bit first
    while (lc < need) {
      if (p >= end) return false;
      c = (c << 8) | *p++;
      lc += 8;
    }
    return true;
  }
  bool get(int n, unsigned& v) {
    if (!fill(n)) return false;
    lc -= n;
    v = (unsigned)((c >> lc) & ((1ull << n) - 1ull));
    return true;
  }
};

// code lengths (6 bits each, zero runs compressed) -> canonical codes: hcode[s] = len | code << 6
bool huf_unpack_table(const uint8_t*& p, const uint8_t* end, int im, int iM, std::vector<uint64_t>& hcode) {
  BitReader br{p, end};
  for (; im <= iM; ++im) {
    unsigned l;
    if (!br.get(6, l)) return false;
    hcode[im] = l;
    if (l == (unsigned)LONG_ZEROCODE_RUN) {
      unsigned z;
      if (!br.get(8, z)) return false;
      int zerun = (int)z + SHORTEST_LONG_RUN;
      if (im + zerun > iM + 1) return false;
      while (zerun--) hcode[im++] = 0;
      --im;
    } else if (l >= (unsigned)SHORT_ZEROCODE_RUN) {
      int zerun = (int)l - SHORT_ZEROCODE_RUN + 2;
      if (im + zerun > iM + 1) return false;
      while (zerun--) hcode[im++] = 0;
      --im;
    }
  }
  p = br.p;   // the data stream starts at the next whole byte
  uint64_t n[59];
  std::memset(n, 0, sizeof(n));
  for (int i = 0; i < HUF_ENCSIZE; ++i) n[hcode[i]] += 1;
  uint64_t c = 0;
  for (int i = 58; i > 0; --i) {   // longest codes get the smallest values
    const uint64_t nc = (c + n[i]) >> 1;
    n[i] = c;
    c = nc;
  }
  for (int i = 0; i < HUF_ENCSIZE; ++i) {
    const int l = (int)hcode[i];
    if (l > 0) hcode[i] = (uint64_t)l | (n[l]++ << 6);
  }
  return true;
}

struct Dec {
  int len = 0;   // code of at most HUF_DECBITS bits: its length (0: a longer code starts with this prefix)
  int lit = 0;   // ... and symbol
};

bool huf_uncompress(const uint8_t* src, long n_src, uint16_t* out, long n_out) {
  if (n_src < 20) return n_out == 0;
  auto rd = [&](int o) { return (uint32_t)src[o] | (uint32_t)src[o + 1] << 8 | (uint32_t)src[o + 2] << 16 | (uint32_t)src[o + 3] << 24; };
  const int im = (int)rd(0), iM = (int)rd(4);
  const long n_bits = (long)rd(12);
  if (im < 0 || im >= HUF_ENCSIZE || iM < 0 || iM >= HUF_ENCSIZE) return false;
  const uint8_t* p = src + 20;
  const uint8_t* end = src + n_src;
  std::vector<uint64_t> hcode(HUF_ENCSIZE, 0);
  if (!huf_unpack_table(p, end, im, iM, hcode)) return false;
  if (n_bits > 8 * (long)(end - p)) return false;
  std::vector<Dec> dec(HUF_DECSIZE);
  // longer codes: canonical numbering gives each length a contiguous code range, symbols in increasing order
  std::vector<int> long_syms[59];
  uint64_t long_first[59];
  for (int l = 0; l < 59; ++l) long_first[l] = ~0ull;
  for (int s = im; s <= iM; ++s) {
    const uint64_t hc = hcode[s];
    const int l = (int)(hc & 63);
    const uint64_t code = hc >> 6;
    if (l == 0) continue;
    if (code >> l) return false;
    if (l > HUF_DECBITS) {
      if (long_syms[l].empty()) long_first[l] = code;
      if (code != long_first[l] + long_syms[l].size()) return false;
      long_syms[l].push_back(s);
    } else {
      const uint64_t first = code << (HUF_DECBITS - l);
      for (uint64_t i = 0; i < (1ull << (HUF_DECBITS - l)); ++i) {
        dec[first + i].len = l;
        dec[first + i].lit = s;
      }
    }
  }
  const int rlc = iM;
  BitReader br{p, p + (n_bits + 7) / 8};
  long bits_left = n_bits, no = 0;
  auto emit = [&](int sym) -> bool {
    if (sym == rlc) {
      if (bits_left < 8) return false;
      unsigned cs;
      if (!br.get(8, cs)) return false;
      bits_left -= 8;
      if (no == 0 || no + (long)cs > n_out) return false;
      const uint16_t s = out[no - 1];
      while (cs--) out[no++] = s;
      return true;
    }
    if (no >= n_out) return false;
    out[no++] = (uint16_t)sym;
    return true;
  };
  while (bits_left > 0) {
    const int avail = (int)(bits_left < 58 ? bits_left : 58);
    br.fill(avail);   // (may stop short at the end of the stream)
    const int have = br.lc < avail ? br.lc : avail;
    if (have <= 0) return false;
    // top `have` bits of the buffer, left-aligned to 14 bits for the table
    const uint64_t window = (uint64_t)(br.c >> (br.lc - have)) & ((1ull << have) - 1ull);
    const uint64_t idx = have >= HUF_DECBITS ? window >> (have - HUF_DECBITS) : (window << (HUF_DECBITS - have)) & (HUF_DECSIZE - 1);
    const Dec& d = dec[idx];
    if (d.len && d.len <= have) {
      br.lc -= d.len;
      bits_left -= d.len;
      if (!emit(d.lit)) return false;
      continue;
    }
    bool found = false;
    for (int l = HUF_DECBITS + 1; l <= have && !found; ++l) {
      if (long_syms[l].empty()) continue;
      const uint64_t k = (window >> (have - l)) - long_first[l];
      if (k < long_syms[l].size()) {
        br.lc -= l;
        bits_left -= l;
        if (!emit(long_syms[l][k])) return false;
        found = true;
      }
    }
    if (!found) return false;
  }
  return no == n_out;
}

inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int16_t ls = (int16_t)l, hs = (int16_t)h;
  const int hi = hs;
  const int ai = ls + (hi & 1) + (hi >> 1);
  a = (uint16_t)(int16_t)ai;
  b = (uint16_t)(int16_t)(ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  const int aa = (d + bb - 0x8000) & 0xffff;
  b = (uint16_t)bb;
  a = (uint16_t)aa;
}

void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1 << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1, p2;
  while (p <= n) p <<= 1;
  p >>= 1;
  p2 = p;
  p >>= 1;
  auto wd = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { w14 ? wdec14(l, h, a, b) : wdec16(l, h, a, b); };
  while (p >= 1) {
    uint16_t* py = in;
    uint16_t* const ey = in + (long)oy * (ny - p2);
    const long oy1 = (long)oy * p, oy2 = (long)oy * p2, ox1 = (long)ox * p, ox2 = (long)ox * p2;
    uint16_t i00, i01, i10, i11;
    for (; py <= ey; py += oy2) {
      uint16_t* px = py;
      uint16_t* const ex = py + (long)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1;
        uint16_t* p10 = px + oy1;
        uint16_t* p11 = p10 + ox1;
        wd(*px, *p10, i00, i10);
        wd(*p01, *p11, i01, i11);
        wd(i00, i01, *px, *p01);
        wd(i10, i11, *p10, *p11);
      }
      if (nx & p) {
        uint16_t* p10 = px + oy1;
        wd(*px, *p10, i00, *p10);
        *px = i00;
      }
    }
    if (ny & p) {
      uint16_t* px = py;
      uint16_t* const ex = py + (long)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1;
        wd(*px, *p01, i00, *p01);
        *px = i00;
      }
    }
    p2 = p;
    p >>= 1;
  }
}

}  // namespace

// One PIZ chunk -> 16-bit words, channel-major ([channel][line][pixel][word of the pixel]).
// chan: n_ch rows of (pixels per line, lines, 16-bit words per pixel: 1 = HALF, 2 = FLOAT/UINT).
extern "C" int rb_exr_piz_decode(const unsigned char* src, long n_src, const int* chan, int n_ch, unsigned short* out,
                                 long n_out) {
  RB_REQUIRE(src && chan && out && n_ch > 0 && n_src >= 4 && n_out > 0, "bad arguments");
  long total = 0;
  for (int c = 0; c < n_ch; ++c) total += (long)chan[3 * c] * chan[3 * c + 1] * chan[3 * c + 2];
  RB_REQUIRE(total == n_out, "output size does not match the channel list");
  const uint8_t* p = src;
  const uint8_t* const end = src + n_src;
  const int min_nz = p[0] | p[1] << 8, max_nz = p[2] | p[3] << 8;
  p += 4;
  RB_REQUIRE(max_nz < BITMAP_SIZE, "PIZ: bad bitmap range");
  std::vector<uint8_t> bitmap(BITMAP_SIZE, 0);
  if (min_nz <= max_nz) {
    RB_REQUIRE(p + (max_nz - min_nz + 1) <= end, "PIZ: truncated bitmap");
    std::memcpy(bitmap.data() + min_nz, p, (size_t)(max_nz - min_nz + 1));
    p += max_nz - min_nz + 1;
  }
  std::vector<uint16_t> lut(65536, 0);
  int k = 0;
  for (int i = 0; i < 65536; ++i)
    if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
  const uint16_t max_value = (uint16_t)(k - 1);
  RB_REQUIRE(p + 4 <= end, "PIZ: truncated chunk");
  const long len = (long)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24);
  p += 4;
  RB_REQUIRE(len >= 0 && p + len <= end, "PIZ: bad Huffman block length");
  RB_REQUIRE(huf_uncompress(p, len, out, n_out), "PIZ: corrupt Huffman stream");
  uint16_t* q = out;
  for (int c = 0; c < n_ch; ++c) {
    const int nx = chan[3 * c], ny = chan[3 * c + 1], size = chan[3 * c + 2];
    for (int j = 0; j < size; ++j) wav2_decode(q + j, nx, size, ny, nx * size, max_value);
    q += (long)nx * ny * size;
  }
  for (long i = 0; i < n_out; ++i) out[i] = lut[out[i]];
  return 0;
}

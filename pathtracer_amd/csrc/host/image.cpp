// image.cpp — PPM and PNG decoders for diffuse maps (see image.h).
#include "image.h"

#include <cstdio>
#include <cstring>

namespace pt {
namespace {

bool read_file(const std::string &path, std::vector<uint8_t> &buf) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) return false;
  std::fseek(fp, 0, SEEK_END);
  const long n = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  bool ok = n >= 0;
  if (ok) {
    buf.resize((size_t)n);
    ok = n == 0 || std::fread(buf.data(), 1, (size_t)n, fp) == (size_t)n;
  }
  std::fclose(fp);
  return ok;
}

// ---------------------------------------------------------------- PPM (P6 binary, P3 plain)
bool decode_ppm(const std::vector<uint8_t> &f, Image &out, std::string &err) {
  size_t i = 2;
  auto next_int = [&](long &v) {
    for (;;) {
      while (i < f.size() && (f[i] == ' ' || f[i] == '\t' || f[i] == '\n' || f[i] == '\r')) ++i;
      if (i < f.size() && f[i] == '#') {
        while (i < f.size() && f[i] != '\n') ++i;
        continue;
      }
      break;
    }
    if (i >= f.size() || f[i] < '0' || f[i] > '9') return false;
    v = 0;
    while (i < f.size() && f[i] >= '0' && f[i] <= '9') {
      v = v * 10 + (f[i] - '0');
      if (v > (1L << 30)) return false;
      ++i;
    }
    return true;
  };
  const bool binary = f[1] == '6';
  long w, h, maxv;
  if (!next_int(w) || !next_int(h) || !next_int(maxv) || w <= 0 || h <= 0 || maxv <= 0 || maxv > 255 ||
      (uint64_t)w * (uint64_t)h > (1ull << 28)) {
    err = "bad PPM header";
    return false;
  }
  out.width = (uint32_t)w;
  out.height = (uint32_t)h;
  const size_t n = (size_t)w * h * 3;
  out.rgb.resize(n);
  if (binary) {
    ++i;  // the single whitespace byte after maxval
    if (f.size() < i + n) {
      err = "truncated PPM";
      return false;
    }
    std::memcpy(out.rgb.data(), f.data() + i, n);
  } else {
    for (size_t k = 0; k < n; ++k) {
      long v;
      if (!next_int(v) || v > maxv) {
        err = "truncated PPM";
        return false;
      }
      out.rgb[k] = (uint8_t)v;
    }
  }
  if (maxv != 255)
    for (uint8_t &b : out.rgb) b = (uint8_t)((b * 255 + maxv / 2) / maxv);
  return true;
}

// ---------------------------------------------------------------- inflate
struct Bits {
  const uint8_t *p;
  size_t n, pos = 0;
  uint32_t acc = 0;
  int cnt = 0;
  bool fail = false;
  uint32_t get(int k) {  // k <= 16, least significant bit first
    while (cnt < k) {
      if (pos >= n) {
        fail = true;
        return 0;
      }
      acc |= (uint32_t)p[pos++] << cnt;
      cnt += 8;
    }
    const uint32_t v = acc & ((1u << k) - 1u);
    acc >>= k;
    cnt -= k;
    return v;
  }
};

struct Huffman {  // canonical code, decoded bit by bit (textures are small; speed is irrelevant)
  uint16_t count[16] = {0};
  uint16_t symbol[288] = {0};
  bool build(const uint8_t *len, int n) {
    std::memset(count, 0, sizeof(count));
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
      left = (left << 1) - count[l];
      if (left < 0) return false;  // over-subscribed
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; ++i)
      if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    return true;
  }
  int decode(Bits &b) const {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
      code |= (int)b.get(1);
      if (b.fail) return -1;
      const int c = count[l];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflate_block(Bits &b, const Huffman &lit, const Huffman &dist, std::vector<uint8_t> &out) {
  for (;;) {
    const int s = lit.decode(b);
    if (s < 0) return false;
    if (s < 256) {
      out.push_back((uint8_t)s);
    } else if (s == 256) {
      return true;
    } else {
      if (s > 285) return false;
      const int len = kLenBase[s - 257] + (int)b.get(kLenExtra[s - 257]);
      const int ds = dist.decode(b);
      if (ds < 0 || ds > 29) return false;
      const size_t d = kDistBase[ds] + b.get(kDistExtra[ds]);
      if (b.fail || d > out.size()) return false;
      for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
    }
  }
}

}  // namespace

bool inflate_zlib(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err) {
  if (n < 6 || (src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) {
    err = "not a zlib stream";
    return false;
  }
  Bits b{src + 2, n - 2};
  for (bool last = false; !last;) {
    last = b.get(1) != 0;
    const uint32_t type = b.get(2);
    if (b.fail) break;
    if (type == 0) {
      b.acc = 0;
      b.cnt = 0;  // to the byte boundary
      if (b.pos + 4 > b.n) { b.fail = true; break; }
      const uint32_t len = b.p[b.pos] | (b.p[b.pos + 1] << 8), nlen = b.p[b.pos + 2] | (b.p[b.pos + 3] << 8);
      b.pos += 4;
      if ((len ^ 0xffffu) != nlen || b.pos + len > b.n) { b.fail = true; break; }
      out.insert(out.end(), b.p + b.pos, b.p + b.pos + len);
      b.pos += len;
    } else if (type == 1) {
      uint8_t l[288], dl[30];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      for (int i = 0; i < 30; ++i) dl[i] = 5;
      Huffman lit, dist;
      lit.build(l, 288);
      dist.build(dl, 30);
      if (!inflate_block(b, lit, dist, out)) { b.fail = true; break; }
    } else if (type == 2) {
      const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.get(3);
      Huffman lencode;
      if (b.fail || nlen > 286 || ndist > 30 || !lencode.build(cl, 19)) { b.fail = true; break; }
      uint8_t l[320] = {0};
      int idx = 0;
      while (idx < nlen + ndist) {
        const int s = lencode.decode(b);
        if (s < 0) { b.fail = true; break; }
        if (s < 16) {
          l[idx++] = (uint8_t)s;
        } else {
          int rep, val = 0;
          if (s == 16) {
            if (idx == 0) { b.fail = true; break; }
            val = l[idx - 1];
            rep = 3 + (int)b.get(2);
          } else if (s == 17) {
            rep = 3 + (int)b.get(3);
          } else {
            rep = 11 + (int)b.get(7);
          }
          if (idx + rep > nlen + ndist) { b.fail = true; break; }
          while (rep--) l[idx++] = (uint8_t)val;
        }
      }
      if (b.fail) break;
      Huffman lit, dist;
      if (!lit.build(l, nlen) || !dist.build(l + nlen, ndist) || !inflate_block(b, lit, dist, out)) { b.fail = true; break; }
    } else {
      b.fail = true;
    }
    if (b.fail) break;
  }
  if (b.fail) {
    err = "corrupt deflate data";
    return false;
  }
  return true;
}

namespace {

// ---------------------------------------------------------------- PNG
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool decode_png(const std::vector<uint8_t> &f, Image &out, std::string &err) {
  size_t i = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, color = -1, interlace = 0;
  std::vector<uint8_t> idat, palette;
  bool end = false;
  while (!end && i + 12 <= f.size()) {
    const uint32_t len = be32(&f[i]);
    const char *type = reinterpret_cast<const char *>(&f[i + 4]);
    if ((uint64_t)i + 12 + len > f.size()) break;
    const uint8_t *data = &f[i + 8];
    if (!std::memcmp(type, "IHDR", 4) && len == 13) {
      w = be32(data);
      h = be32(data + 4);
      depth = data[8];
      color = data[9];
      interlace = data[12];
    } else if (!std::memcmp(type, "PLTE", 4)) {
      palette.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    i += 12 + (size_t)len;
  }
  int channels = 0;
  switch (color) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
  }
  if (!end || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28) || channels == 0) {
    err = "bad PNG structure";
    return false;
  }
  if (depth != 8 || interlace != 0) {
    err = "PNG with a bit depth other than 8 or with interlacing is not supported";
    return false;
  }
  std::vector<uint8_t> raw;
  if (!inflate_zlib(idat.data(), idat.size(), raw, err)) return false;
  const size_t stride = (size_t)w * channels;
  if (raw.size() < (stride + 1) * h) {
    err = "PNG data shorter than its header says";
    return false;
  }
  // undo the per-row filters in place (PNG specification, section 9)
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  out.width = w;
  out.height = h;
  out.rgb.resize((size_t)w * h * 3);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t *row = &raw[(stride + 1) * y];
    const int filter = row[0];
    if (filter > 4) {
      err = "bad PNG filter";
      return false;
    }
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= (size_t)channels ? cur[x - channels] : 0, b = prev[x], c = x >= (size_t)channels ? prev[x - channels] : 0;
      int pred = 0;
      if (filter == 1) pred = a;
      else if (filter == 2) pred = b;
      else if (filter == 3) pred = (a + b) >> 1;
      else if (filter == 4) {
        const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      }
      cur[x] = (uint8_t)(row[1 + x] + pred);
    }
    uint8_t *dst = &out.rgb[(size_t)y * w * 3];
    for (uint32_t x = 0; x < w; ++x) {
      const uint8_t *px = &cur[(size_t)x * channels];
      if (color == 0 || color == 4) {
        dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = px[0];
      } else if (color == 3) {
        if ((size_t)px[0] * 3 + 2 >= palette.size()) {
          err = "PNG palette index out of range";
          return false;
        }
        std::memcpy(dst + 3 * x, &palette[(size_t)px[0] * 3], 3);
      } else {
        std::memcpy(dst + 3 * x, px, 3);  // alpha, when present, is dropped as the RGB24 conversion drops it
      }
    }
    prev.swap(cur);
  }
  return true;
}

}  // namespace

bool load_image(const std::string &path, Image &out, std::string &err) {
  std::vector<uint8_t> f;
  if (!read_file(path, f)) {
    err = "cannot read image " + path;
    return false;
  }
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  bool ok;
  if (f.size() >= 8 && !std::memcmp(f.data(), png_sig, 8)) ok = decode_png(f, out, err);
  else if (f.size() >= 2 && f[0] == 'P' && (f[1] == '6' || f[1] == '3')) ok = decode_ppm(f, out, err);
  else {
    err = "unsupported image format (PNG and PPM are decoded)";
    ok = false;
  }
  if (!ok) err = path + ": " + err;
  return ok;
}

}  // namespace pt

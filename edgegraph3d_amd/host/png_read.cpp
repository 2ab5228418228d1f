// PNG reader for the edge images of the step before the path. The reference loads them
// with cv::imread (edge_graph_3d_utilities.cpp:332) and compares every pixel with
// EDGE_COLOR = (255,255,255) (global_defines.hpp:47; is_edge,
// convert_edge_images_pixel_to_segment.cpp:205-207); OpenCV is not available here, so this is an own
// decoder of what the format needs: all colour types and bit depths, the five scanline filters,
// zlib streams split over any number of IDAT chunks (inflate from the system's libz); no
// interlacing (the dtu006 edge maps are 1-bit greyscale, non-interlaced). Output: one byte per pixel,
// 1 where an 8-bit BGR load of the image would be pure white.
#include <zlib.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eg3d_host.h"

namespace {
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

// Limits of what is decoded (a hostile file must not be able to take the process down): the file itself and the
// decoded image are bounded before anything is allocated, and no C++ exception leaves the extern "C" functions.
static const size_t kMaxFileBytes = (size_t)1 << 30;     // 1 GiB of PNG
static const size_t kMaxDecodedBytes = (size_t)1 << 30;  // 1 GiB of decoded scanlines (e.g. 16384 x 16384 RGBA 8-bit)

static int png_read_edge_mask_impl(const char* path, int* width, int* height, uint8_t** mask_out) {
  if (!path || !width || !height || !mask_out) return -1;
  *mask_out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return -2;
  std::vector<unsigned char> file;
  {
    unsigned char buf[1 << 16];
    size_t n;
    bool too_big = false;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) {
      if (file.size() + n > kMaxFileBytes) {
        too_big = true;
        break;
      }
      try {
        file.insert(file.end(), buf, buf + n);
      } catch (...) {
        fclose(f);
        return -6;
      }
    }
    fclose(f);
    if (too_big) return -4;
  }
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return -3;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat, plte;
  size_t i = 8;
  bool end = false;
  while (!end && i + 12 <= file.size()) {
    const uint32_t n = be32(&file[i]);
    if (n > file.size() || i + 12 + (size_t)n > file.size()) return -3;
    const unsigned char* type = &file[i + 4];
    const unsigned char* body = &file[i + 8];
    if (!memcmp(type, "IHDR", 4) && n >= 13) {
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(body, body + n);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + n);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    i += 12 + (size_t)n;
  }
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return -3;
  }
  if (!w || !h || w > (1u << 15) || h > (1u << 15) || interlace != 0 || idat.empty()) return -4;
  if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return -3;
  if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) return -3;
  if (ctype == 3 && (depth == 16 || plte.empty())) return -3;
  const size_t bits_pp = (size_t)channels * depth;
  const size_t stride = ((size_t)w * bits_pp + 7) / 8;
  const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  if ((stride + 1) > kMaxDecodedBytes / (size_t)h) return -4;  // a ~100-byte header must not buy gigabytes
  std::vector<unsigned char> raw((stride + 1) * (size_t)h);
  {
    // zlib counts avail_in / avail_out in 32 bits: both sizes are bounded far below that (limits above)
    static_assert(kMaxFileBytes <= UINT_MAX && kMaxDecodedBytes <= UINT_MAX, "one inflate call must cover the stream");
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return -5;
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool full = zs.total_out == raw.size();
    inflateEnd(&zs);
    if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || !full) return -5;
  }
  // unfilter in place
  std::vector<unsigned char> zero(stride, 0);
  for (uint32_t r = 0; r < h; r++) {
    unsigned char* line = &raw[(stride + 1) * (size_t)r + 1];
    const unsigned char* prev = r ? &raw[(stride + 1) * (size_t)(r - 1) + 1] : zero.data();
    const int ft = raw[(stride + 1) * (size_t)r];
    for (size_t x = 0; x < stride; x++) {
      const int a = x >= bpp ? line[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
      int p;
      switch (ft) {
        case 0: p = 0; break;
        case 1: p = a; break;
        case 2: p = b; break;
        case 3: p = (a + b) / 2; break;
        case 4: p = paeth(a, b, c); break;
        default: return -3;
      }
      line[x] = (unsigned char)(line[x] + p);
    }
  }
  uint8_t* mask = (uint8_t*)malloc((size_t)w * h);
  if (!mask) return -6;
  const int maxv = depth >= 8 ? 255 : (1 << depth) - 1;
  for (uint32_t r = 0; r < h; r++) {
    const unsigned char* line = &raw[(stride + 1) * (size_t)r + 1];
    for (uint32_t x = 0; x < w; x++) {
      int s[4] = {0, 0, 0, 0};
      for (int ch = 0; ch < channels; ch++) {
        const size_t k = (size_t)x * channels + ch;
        if (depth == 8)
          s[ch] = line[k];
        else if (depth == 16)
          s[ch] = line[2 * k];  // the 8-bit load keeps the high byte
        else {
          const size_t bit = k * depth;
          s[ch] = (line[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
        }
      }
      bool white;
      if (ctype == 3) {
        const size_t e = (size_t)s[0] * 3;
        white = e + 2 < plte.size() && plte[e] == 255 && plte[e + 1] == 255 && plte[e + 2] == 255;
      } else if (ctype == 0 || ctype == 4) {
        white = s[0] == maxv;  // grey replicated to B, G, R, scaled to 8 bits
      } else {
        white = s[0] == 255 && s[1] == 255 && s[2] == 255;
      }
      mask[(size_t)r * w + x] = white ? 1 : 0;
    }
  }
  *width = (int)w;
  *height = (int)h;
  *mask_out = mask;
  return 0;
}

extern "C" int eg3d_png_read_edge_mask(const char* path, int* width, int* height, uint8_t** mask_out) {
  try {
    return png_read_edge_mask_impl(path, width, height, mask_out);
  } catch (...) {  // std::bad_alloc and friends: an error code, never std::terminate through the C ABI
    if (mask_out && *mask_out) {
      free(*mask_out);
      *mask_out = nullptr;
    }
    return -6;
  }
}

extern "C" int eg3d_plg_build_from_png(const char* path, int* width, int* height, eg3d_plg_view* out) {
  uint8_t* mask = nullptr;
  int w = 0, h = 0;
  const int rc = eg3d_png_read_edge_mask(path, &w, &h, &mask);
  if (rc != 0) return rc;
  const int rb = eg3d_plg_build_from_mask(mask, w, h, out);
  free(mask);
  if (width) *width = w;
  if (height) *height = h;
  return rb;
}

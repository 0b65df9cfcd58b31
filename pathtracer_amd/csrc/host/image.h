// image.h — decoding of diffuse maps.  The reference hands map_Kd files to FFmpeg and converts
// the first frame to RGB24 (Image, object.cpp:31-142); FFmpeg is not available here, so the two
// lossless formats that cover OBJ material libraries in practice are decoded directly: PNG
// (8-bit grey, grey+alpha, RGB, RGBA and palette, non-interlaced) and binary/plain PPM.
#ifndef PT_HOST_IMAGE_H
#define PT_HOST_IMAGE_H

#include <cstdint>
#include <string>
#include <vector>

namespace pt {

struct Image {
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> rgb;  // width*height*3, rows top to bottom (AV_PIX_FMT_RGB24, object.cpp:89-95)
};

// Returns false and fills `err` when the file cannot be read or is not one of the formats above
// (the reference aborts in that case, object.cpp:40-45).
bool load_image(const std::string &path, Image &out, std::string &err);

// zlib stream -> bytes (RFC 1950/1951); used by the PNG decoder.
bool inflate_zlib(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err);

}  // namespace pt
#endif

// math.h — host-side vector type with the reference's arithmetic (math.cpp:20-183):
// float components, one rounding per operation, Euler rotation evaluated in double.
#ifndef PT_HOST_MATH_H
#define PT_HOST_MATH_H

#include <cmath>
#include <limits>

namespace pt {

using scalar = float;                                             // math.cpp:10
constexpr scalar INF = std::numeric_limits<scalar>::infinity();   // math.cpp:11

struct vec {
  scalar x = 0, y = 0, z = 0;
  constexpr vec() = default;
  constexpr vec(scalar n) : x(n), y(n), z(n) {}
  constexpr vec(scalar a, scalar b, scalar c) : x(a), y(b), z(c) {}

  vec operator-() const { return vec(-x, -y, -z); }
  vec operator+(const vec &v) const { return vec(x + v.x, y + v.y, z + v.z); }
  vec operator-(const vec &v) const { return vec(x - v.x, y - v.y, z - v.z); }
  vec operator*(const vec &v) const { return vec(x * v.x, y * v.y, z * v.z); }
  vec operator/(const vec &v) const { return vec(x / v.x, y / v.y, z / v.z); }
  vec &operator+=(const vec &v) { x += v.x; y += v.y; z += v.z; return *this; }
  vec &operator-=(const vec &v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
  bool operator==(const vec &v) const { return x == v.x && y == v.y && z == v.z; }
  bool operator!=(const vec &v) const { return !(*this == v); }
  scalar operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }

  scalar dot(const vec &v) const { return x * v.x + y * v.y + z * v.z; }     // math.cpp:108
  scalar norm() const { return std::sqrt(x * x + y * y + z * z); }            // math.cpp:109
  vec normalize() const {                                                      // math.cpp:111-114
    scalar n = norm();
    return n == 0 ? *this : vec(x / n, y / n, z / n);
  }
  vec cross(const vec &v) const {                                              // math.cpp:124-126
    return vec(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x);
  }
  // math.cpp:130-143: angles are floats, every product is double, one narrowing per component
  vec rotate(scalar a, scalar b, scalar g) const {
    const double ca = std::cos((double)a), sa = std::sin((double)a);
    const double cb = std::cos((double)b), sb = std::sin((double)b);
    const double cg = std::cos((double)g), sg = std::sin((double)g);
    const double X = x, Y = y, Z = z;
    return vec((scalar)((cb * cg * X) + ((sa * sb * cg - ca * sg) * Y) + ((ca * sb * cg + sa * sg) * Z)),
               (scalar)((cb * sg * X) + ((sa * sb * sg + ca * cg) * Y) + ((ca * sb * sg - sa * cg) * Z)),
               (scalar)((-sb * X) + (sa * cb * Y) + (ca * cb * Z)));
  }
  vec rotate(const vec &e) const { return rotate(e.x, e.y, e.z); }
};

inline vec operator*(scalar n, const vec &v) { return vec(v.x * n, v.y * n, v.z * n); }  // math.cpp:168-171
inline vec operator*(const vec &v, scalar n) { return vec(v.x * n, v.y * n, v.z * n); }  // math.cpp:172-175
inline vec operator/(const vec &v, scalar n) { return vec(v.x / n, v.y / n, v.z / n); }  // math.cpp:180-183

}  // namespace pt
#endif

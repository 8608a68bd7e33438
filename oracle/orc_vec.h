/* orc_vec.h -- ORACLE (test infrastructure, not product code).
 *
 * Small vector / matrix types with the evaluation order the oracle assigns to Taichi's vector
 * operations: component-wise arithmetic, dot and matrix products
 * summed left to right, normalized() = v * (1 / sqrt(x*x + y*y + z*z)).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.
 */
#ifndef ORC_VEC_H
#define ORC_VEC_H

#include "../include/vrt_detmath.h"

namespace orc {

static const float EPS = 1e-6f;  /* math_utils.py:5 */
static const float INF = DM_INF; /* math_utils.py:6 */
static const float PI = DM_PI;   /* np.pi rounded to f32 */

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct V4 {
    float x, y, z, w;
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct I3 {
    int x, y, z;
    int& operator[](int i) { return (&x)[i]; }
    int operator[](int i) const { return (&x)[i]; }
};
struct M4 {
    float m[4][4]; /* m[row][col] */
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(float s) { return V3{s, s, s}; }
inline V2 v2(float x, float y) { return V2{x, y}; }
inline V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }

inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator+(V3 a, float s) { return V3{a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(V3 a, float s) { return V3{a.x - s, a.y - s, a.z - s}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
inline V3& operator*=(V3& a, float s) { a = a * s; return a; }
inline V3& operator/=(V3& a, V3 b) { a = a / b; return a; }
inline V3& operator/=(V3& a, float s) { a = a / s; return a; }

inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, V2 b) { return V2{a.x * b.x, a.y * b.y}; }
inline V2 operator*(V2 a, float s) { return V2{a.x * s, a.y * s}; }
inline V2 operator/(V2 a, float s) { return V2{a.x / s, a.y / s}; }
inline V2 operator+(V2 a, float s) { return V2{a.x + s, a.y + s}; }

inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(V3 a) { return dm_sqrt(dot(a, a)); }
inline float length(V2 a) { return dm_sqrt(dot(a, a)); }
inline float distance(V3 a, V3 b) { return length(a - b); }
/* Taichi Vector.normalized(): invlen = 1 / (norm + 0); return invlen * v */
inline V3 normalized(V3 a) { float inv = 1.0f / length(a); return inv * a; }
inline V2 normalized(V2 a) { float inv = 1.0f / length(a); return V2{inv * a.x, inv * a.y}; }
inline V3 vabs(V3 a) { return V3{dm_abs(a.x), dm_abs(a.y), dm_abs(a.z)}; }
inline V3 vmin(V3 a, V3 b) { return V3{dm_min(a.x, b.x), dm_min(a.y, b.y), dm_min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{dm_max(a.x, b.x), dm_max(a.y, b.y), dm_max(a.z, b.z)}; }
inline V3 vmax(V3 a, float s) { return V3{dm_max(a.x, s), dm_max(a.y, s), dm_max(a.z, s)}; }
inline V3 vclamp(V3 a, float lo, float hi) {
    return V3{dm_clamp(a.x, lo, hi), dm_clamp(a.y, lo, hi), dm_clamp(a.z, lo, hi)};
}
inline V3 vfloor(V3 a) { return V3{dm_floor(a.x), dm_floor(a.y), dm_floor(a.z)}; }
inline float max3(float a, float b, float c) { return dm_max(dm_max(a, b), c); }
inline float min3(float a, float b, float c) { return dm_min(dm_min(a, b), c); }
inline float sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
inline float saturate(float x) { return dm_min(dm_max(x, 0.0f), 1.0f); } /* math_utils.py:10 */
inline V3 saturate(V3 a) { return V3{saturate(a.x), saturate(a.y), saturate(a.z)}; }
inline float sqr(float x) { return x * x; }                               /* math_utils.py:14 */
/* ti.math.mix(x, y, a) = x * (1 - a) + y * a */
inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline V3 mix(V3 x, V3 y, float a) { return x * (1.0f - a) + y * a; }
inline float fract(float x) { return x - dm_floor(x); }
inline V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }
inline bool is_vec_zero(V3 x) { return dot(x, x) < 1e-7f; } /* math_utils.py:18 */

inline V4 mul(const M4& M, V4 v) {
    V4 r;
    for (int i = 0; i < 4; i++)
        r[i] = M.m[i][0] * v.x + M.m[i][1] * v.y + M.m[i][2] * v.z + M.m[i][3] * v.w;
    return r;
}
inline V3 xyz(V4 v) { return V3{v.x, v.y, v.z}; }

} /* namespace orc */
#endif

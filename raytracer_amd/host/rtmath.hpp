// rtmath.hpp — minimal fp32 vector/matrix maths for the host side of the trace path.
//
// The reference host code does its arithmetic with glm 0.9.9.7 (vendored under the
// reference's include/glm).  Only a handful of glm entry points are on the hot path
// (Camera.cpp:9-17, Tracer.cpp:110-130, BVH.cpp:22-51), so instead of vendoring glm this
// header restates exactly those with the same fp32 operation order, so that the ray
// corners handed to the launcher are bit-identical to what the reference would upload.
// The restatement is pinned by oracle/ref_glm_camera.cpp (built against the reference's
// own glm headers in the build container) → tests/golden/camera_rays.json.
//
// Everything here is plain IEEE fp32; the host library is compiled with
// -ffp-contract=off so no multiply-add is fused.
#ifndef RT_AMD_RTMATH_HPP
#define RT_AMD_RTMATH_HPP

#include <cmath>

namespace rtm {

struct vec2 {
    float x, y;
    vec2() : x(0.f), y(0.f) {}
    vec2(float a, float b) : x(a), y(b) {}
};

struct vec3 {
    float x, y, z;
    vec3() : x(0.f), y(0.f), z(0.f) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct vec4 {
    float x, y, z, w;
    vec4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct ivec2 {
    int x, y;
    ivec2() : x(0), y(0) {}
    ivec2(int a, int b) : x(a), y(b) {}
};

inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 &operator+=(vec3 &a, const vec3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline vec3 &operator/=(vec3 &a, float s) { a.x /= s; a.y /= s; a.z /= s; return a; }

inline vec2 operator+(const vec2 &a, const vec2 &b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator*(const vec2 &a, float s) { return vec2(a.x * s, a.y * s); }

inline vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator*(const vec4 &a, const vec4 &b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator/(const vec4 &a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }

// glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z (left to right).
inline float dot(const vec3 &a, const vec3 &b) {
    vec3 t = a * b;
    return t.x + t.y + t.z;
}
// glm compute_cross.
inline vec3 cross(const vec3 &x, const vec3 &y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// glm compute_normalize: v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt.
inline vec3 normalize(const vec3 &v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// Column-major 4x4, m[c][r] like glm.
struct mat4 {
    vec4 c[4];
    mat4() {}
    explicit mat4(float d) {
        c[0] = vec4(d, 0, 0, 0);
        c[1] = vec4(0, d, 0, 0);
        c[2] = vec4(0, 0, d, 0);
        c[3] = vec4(0, 0, 0, d);
    }
    mat4(const vec4 &a, const vec4 &b, const vec4 &cc, const vec4 &d) { c[0] = a; c[1] = b; c[2] = cc; c[3] = d; }
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};

// glm operator*(mat4, mat4): Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3].
inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int j = 0; j < 4; ++j)
        r[j] = a[0] * b[j][0] + a[1] * b[j][1] + a[2] * b[j][2] + a[3] * b[j][3];
    return r;
}
inline mat4 operator*(const mat4 &m, float s) { return mat4(m[0] * s, m[1] * s, m[2] * s, m[3] * s); }

// glm operator*(vec4 row, mat4): component j = m[j][0]*v0 + m[j][1]*v1 + m[j][2]*v2 + m[j][3]*v3.
inline vec4 operator*(const vec4 &v, const mat4 &m) {
    vec4 r;
    for (int j = 0; j < 4; ++j)
        r[j] = m[j][0] * v[0] + m[j][1] * v[1] + m[j][2] * v[2] + m[j][3] * v[3];
    return r;
}

inline mat4 transpose(const mat4 &m) {
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r[i][j] = m[j][i];
    return r;
}

// Inverse by cofactors.  The camera rays must agree bit for bit with the ones the reference computes with its
// vendored glm (tests/golden/camera_rays.json), so every product, difference and sum below is formed in the order
// glm's 4x4 inverse forms it (glm is MIT / Happy Bunny licensed, (c) G-Truc Creation; the arithmetic order is what
// is followed here, the code is this repository's):
//   minors of columns {2,3}, {1,3}, {1,2} over the row pairs (2,3) (1,3) (1,2) (0,3) (0,2) (0,1);
//   cofactor column c = lead[a] * M(b,d) - lead[b] * M(a,d) + lead[d] * M(a,b) for the rows a < b < d other than c,
//   with lead[r] = (m[1][r], m[0][r], m[0][r], m[0][r]); signs (-1)^(i + c); determinant from column 0 of m and
//   row 0 of the cofactor matrix, summed as (x + y) + (z + w); the result is cofactors * (1 / determinant).
inline mat4 inverse(const mat4 &m) {
    auto minor2 = [&m](int ca, int cb, int r, int s) { return m[ca][r] * m[cb][s] - m[cb][r] * m[ca][s]; };
    static const int pairs[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
    auto pairIndex = [](int r, int s) {
        for (int k = 0; k < 6; ++k)
            if (pairs[k][0] == r && pairs[k][1] == s) return k;
        return 0;
    };
    vec4 M[6], lead[4];
    for (int k = 0; k < 6; ++k) {
        const int r = pairs[k][0], s = pairs[k][1];
        const float m23 = minor2(2, 3, r, s);
        M[k] = vec4(m23, m23, minor2(1, 3, r, s), minor2(1, 2, r, s));
    }
    for (int r = 0; r < 4; ++r) lead[r] = vec4(m[1][r], m[0][r], m[0][r], m[0][r]);
    mat4 cof;
    for (int c = 0; c < 4; ++c) {
        int o[3], n = 0;
        for (int r = 0; r < 4; ++r)
            if (r != c) o[n++] = r;
        const vec4 col = lead[o[0]] * M[pairIndex(o[1], o[2])] - lead[o[1]] * M[pairIndex(o[0], o[2])] +
                         lead[o[2]] * M[pairIndex(o[0], o[1])];
        for (int i = 0; i < 4; ++i) cof[c][i] = ((i + c) & 1) ? col[i] * -1.0f : col[i] * 1.0f;
    }
    const vec4 d = m[0] * vec4(cof[0][0], cof[1][0], cof[2][0], cof[3][0]);
    const float det = (d.x + d.y) + (d.z + d.w);
    return cof * (1.0f / det);
}

// glm::radians<float>: degrees * float(0.0174532925199432957...)
inline float radians(float deg) { return deg * static_cast<float>(0.01745329251994329576923690768489); }

// glm::lookAt (right-handed), ext/matrix_transform.inl lookAtRH.
inline mat4 lookAt(const vec3 &eye, const vec3 &center, const vec3 &up) {
    const vec3 f(normalize(center - eye));
    const vec3 s(normalize(cross(f, up)));
    const vec3 u(cross(s, f));
    mat4 r(1.0f);
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

// glm::perspective (RH, depth -1..1), ext/matrix_clip_space.inl perspectiveRH_NO.
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) {
    const float tanHalfFovy = std::tan(fovy / 2.0f);
    mat4 r(0.0f);
    r[0][0] = 1.0f / (aspect * tanHalfFovy);
    r[1][1] = 1.0f / (tanHalfFovy);
    r[2][2] = -(zFar + zNear) / (zFar - zNear);
    r[2][3] = -1.0f;
    r[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return r;
}

} // namespace rtm

#endif

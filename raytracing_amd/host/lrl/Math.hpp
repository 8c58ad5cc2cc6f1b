// Small vector types for the host-side scene library.  Stand-ins for the reference's external
// <_Math.h> (Math::vec2/vec3/vec4, absent from /root/reference; use sites BVH/BVH.cpp:31,283-310).
// Layout-compatible with GLSL vec2/vec3/vec4 members of the std430 records in include/lrl_types.h.
#pragma once
#include <cmath>
#include <cstdint>
#include <initializer_list>

namespace lrl {
namespace Math {

constexpr double Pi = 3.14159265358979323846;

template <class T> struct vec2 {
    T data[2];
    vec2() : data{0, 0} {}
    vec2(T a, T b) : data{a, b} {}
    T& operator[](int i) { return data[i]; }
    T const& operator[](int i) const { return data[i]; }
};

template <class T> struct vec3 {
    T data[3];
    vec3() : data{0, 0, 0} {}
    vec3(double s) : data{T(s), T(s), T(s)} {}           // scalar broadcast (Color{1,-1,...} use sites)
    vec3(double a, double b, double c) : data{T(a), T(b), T(c)} {}
    template <class U> explicit vec3(vec3<U> const& o) : data{T(o.data[0]), T(o.data[1]), T(o.data[2])} {}
    T& operator[](int i) { return data[i]; }
    T const& operator[](int i) const { return data[i]; }
    vec3 operator+(vec3 const& o) const { return {data[0] + o.data[0], data[1] + o.data[1], data[2] + o.data[2]}; }
    vec3 operator-(vec3 const& o) const { return {data[0] - o.data[0], data[1] - o.data[1], data[2] - o.data[2]}; }
    vec3 operator*(T s) const { return {data[0] * s, data[1] * s, data[2] * s}; }
    T dot(vec3 const& o) const { return data[0] * o.data[0] + data[1] * o.data[1] + data[2] * o.data[2]; }
    vec3 cross(vec3 const& o) const {
        return {data[1] * o.data[2] - data[2] * o.data[1], data[2] * o.data[0] - data[0] * o.data[2],
                data[0] * o.data[1] - data[1] * o.data[0]};
    }
};

template <class T> struct vec4 {
    T data[4];
    vec4() : data{0, 0, 0, 0} {}
    vec4(double a, double b, double c, double d = 0) : data{T(a), T(b), T(c), T(d)} {}
    T& operator[](int i) { return data[i]; }
    T const& operator[](int i) const { return data[i]; }
};

template <class T> struct mat3 {
    T data[3][3];
    static mat3 id() { mat3 m{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.data[i][j] = T(i == j); return m; }
    vec3<T> operator*(vec3<T> const& v) const {
        return {data[0][0] * v[0] + data[0][1] * v[1] + data[0][2] * v[2],
                data[1][0] * v[0] + data[1][1] * v[1] + data[1][2] * v[2],
                data[2][0] * v[0] + data[2][1] * v[1] + data[2][2] * v[2]};
    }
    mat3 operator*(mat3 const& o) const {
        mat3 r{};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            T s = 0; for (int k = 0; k < 3; ++k) s += data[i][k] * o.data[k][j];
            r.data[i][j] = s;
        }
        return r;
    }
};

}  // namespace Math
}  // namespace lrl

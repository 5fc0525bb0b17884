// pf_glm.h -- the handful of glm types the kernel.h API surface uses, as layout-identical PODs.
// Define PFSLAM_HAVE_GLM (and put glm on the include path) to use the real library instead.
#pragma once
#ifdef PFSLAM_HAVE_GLM
#include <glm/glm.hpp>
#else
namespace glm {
struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float a, float b) : x(a), y(b) {} explicit vec2(float a) : x(a), y(a) {} };
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
};
struct ivec2 { int x, y; ivec2() : x(0), y(0) {} ivec2(int a, int b) : x(a), y(b) {} };
} // namespace glm
#endif

// Shared host/device helpers for librag_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <utility>

#include "rag_amd.h"

namespace ragmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// thread-local last-error text, returned by ragmi_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

// compile-time unrolled loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ATen's linear-interpolation source index (UpSample.h: area_pixel_compute_source_index +
// guard_index_and_lambda), evaluated in fp32.  `scale` is (in-1)/(out-1) for
// align_corners, in/out otherwise, computed on the host in fp32.
struct LinIdx {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ LinIdx lin_index(int dst, int in_size, int out_size, float scale, int align_corners) {
  LinIdx r;
  if (in_size == out_size) {
    r.i0 = r.i1 = dst;
    r.w0 = 1.f;
    r.w1 = 0.f;
    return r;
  }
  float src;
  if (align_corners) {
    src = scale * (float)dst;
  } else {
    src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
  }
  int i0 = (int)src;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  float lam = src - (float)i0;
  lam = lam < 0.f ? 0.f : (lam > 1.f ? 1.f : lam);
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  r.w1 = lam;
  r.w0 = 1.f - lam;
  return r;
}
inline float lin_scale(int in_size, int out_size, int align_corners) {
  if (align_corners) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}

}  // namespace ragmi

#define RAGMI_REQUIRE(cond, code, ...)                    \
  do {                                                    \
    if (!(cond)) return ::ragmi::fail((code), __VA_ARGS__); \
  } while (0)

// Shared host/device helpers for librag_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <type_traits>
#include <utility>

#include "rag_amd.h"

namespace ragmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activation storage types.  RAGMI_F32: float.  RAGMI_BF16: bf16 in HBM (raw 16-bit), fp32 everywhere on chip
// (LDS tiles, MFMA operands, accumulators, BN/ReLU) and rounded to nearest-even once, at the store.
typedef unsigned short bf16_t;
template <class T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return __uint_as_float((unsigned)(*p) << 16); }
__device__ __forceinline__ bf16_t to_bf16(float v) {
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // keep a NaN a (quiet) NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
  return (bf16_t)(u >> 16);
}
template <class T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = to_bf16(v); }
// 4 consecutive elements (16 B of float / 8 B of bf16); pointer must be aligned to 4 elements
template <class T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <class T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
  t.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = t;
}
inline bool dtype_ok(int dtype) { return dtype == RAGMI_F32 || dtype == RAGMI_BF16; }
// the 3x3x3 convolution entry points also take RAGMI_F32X3: fp32 storage, bf16x3 split products where the shape is eligible
inline bool conv_dtype_ok(int dtype) { return dtype_ok(dtype) || dtype == RAGMI_F32X3; }
inline size_t dtype_size(int dtype) { return dtype == RAGMI_BF16 ? 2 : 4; }
// 4-element vector paths need the pointer aligned to 4 elements
inline bool aligned4(const void* p, int dtype) { return (reinterpret_cast<uintptr_t>(p) & (4 * dtype_size(dtype) - 1)) == 0; }

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

// Host-side launch state of ONE kernel instantiation, kept per device and behind a mutex: hipFuncSetAttribute applies to the
// current device only (a process that serves a second GPU must set it there too), and the forward (main thread) and backward
// (autograd worker thread) paths launch concurrently.  Read-only after the first call per (device, LDS size): the library stays
// stateless in the sense of include/rag_amd.h (no pointers kept, nothing that changes results).
struct LaunchState {
  static constexpr int MAX_DEV = 16, MAX_SIZES = 8;
  std::mutex mu;
  bool attr_done[MAX_DEV] = {};
  int cus[MAX_DEV] = {};
  size_t cached_lds[MAX_DEV][MAX_SIZES] = {};
  int cached_slots[MAX_DEV][MAX_SIZES] = {};
  int ncached[MAX_DEV] = {};

  // device slot of the cache, or -1 when the current device cannot be identified or lies past the table: such a device is never
  // aliased onto another one's entries — its attribute is set and its occupancy queried on every call instead
  static int device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (dev >= 0 && dev < MAX_DEV) ? dev : -1;
  }
  // raise the dynamic-LDS limit of `fn` on the current device (once per cached device); false on a runtime error
  bool ensure_attr(const void* fn, size_t max_dynamic_lds) {
    const int dev = device_slot();
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && attr_done[dev]) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_dynamic_lds) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    if (dev >= 0) attr_done[dev] = true;
    return true;
  }
  // workgroups of `threads` threads and `lds` dynamic bytes resident on the whole current device (occupancy x CUs); 0 on error
  int slots(const void* fn, int threads, size_t lds, size_t max_dynamic_lds) {
    if (!ensure_attr(fn, max_dynamic_lds)) return 0;
    const int dev = device_slot();
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0)
      for (int i = 0; i < ncached[dev]; ++i)
        if (cached_lds[dev][i] == lds) return cached_slots[dev][i];
    int ncu = dev >= 0 ? cus[dev] : 0;
    if (ncu == 0) {
      int cur = 0;
      hipDeviceProp_t prop;
      ncu = (hipGetDevice(&cur) == hipSuccess && hipGetDeviceProperties(&prop, cur) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
      if (dev >= 0) cus[dev] = ncu;
    }
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    (void)hipGetLastError();
    const int n = per_cu * ncu;
    if (dev >= 0 && ncached[dev] < MAX_SIZES) { cached_lds[dev][ncached[dev]] = lds; cached_slots[dev][ncached[dev]] = n; ++ncached[dev]; }
    return n;
  }
};

// ATen's linear-interpolation source index (UpSample.h: area_pixel_compute_source_index +
// guard_index_and_lambda), evaluated in fp32.  `scale` is (in-1)/(out-1) for
// align_corners, in/out otherwise, computed on the host in fp32.
struct LinIdx {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ LinIdx lin_index(int dst, int in_size, int out_size, float scale, int align_corners) {
  // ATen's arithmetic, bit for bit (checked against torch CPU on this image, round 4: 100 % identical outputs in 1-D):
  //   align_corners=True : src = fl(scale * dst); i0 = (int)src; lambda = src - i0   (the product is ROUNDED before it is used twice)
  //   align_corners=False: src = fma(scale, dst + 0.5, -0.5)                          (one rounding: the CPU build contracts it)
  // hipcc's default -ffp-contract=fast would fuse the first product into the subtraction too — a more accurate lambda (by up to
  // 8e-6 at an index of ~400) but not the reference's, and fused in one kernel and not in another (the standalone upsample kernel
  // and the fused head differed by 4e-5 at the headline shape for this reason alone) — so contraction is off in here and the one
  // fused operation is written out.
#pragma clang fp contract(off)
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
    src = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
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
// one linear interpolation w0 * a + w1 * b in ATen's rounding: the second product rounded, the first fused into the sum (what its
// CPU kernels compile to: identical bits in the 1-D check above); x innermost, then y, then z in every kernel that mirrors ATen
__device__ __forceinline__ float lerp2(float w0, float a, float w1, float b) {
#pragma clang fp contract(off)
  return __builtin_fmaf(w0, a, w1 * b);
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

// clip_grad_norm_ + SGD(momentum, weight decay) over the flat parameter / gradient buffers of the training step as TWO
// launches (reference: approaches/rag.py:64-70 — torch.optim.SGD(lr, momentum=0.9, weight_decay=3e-3) — and rag.py:215-216,
// clip_grad_norm_(parameters, 5) then optimizer.step()).  torch's multi-tensor path issues several launches per group of
// ~500 small tensors (~1.5 ms of a 15 ms step); with every parameter a view of one buffer the update is a single
// HBM-bound pass: 5 floats of traffic per element (p, g, buf read; p, buf written; g written back clipped).
#include "common.h"

namespace ragmi {

constexpr int SGD_PARTS = 256;

// part[k] = sum of g[i]^2 over the k-th grid-stride slice (fp64 accumulation, fixed order: deterministic)
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = g[i];
    s += v * v;
  }
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct SgdArgs {
  float* p;
  float* g;
  float* buf;
  const double* part;
  float* norm_out;
  int64_t n;
  float lr, momentum, weight_decay, max_norm;
  int first_step;
};

// every workgroup re-reduces the SGD_PARTS partial sums (same order everywhere -> the same coefficient), then updates its slice:
//   g *= min(1, max_norm / (||g|| + 1e-6));  d = g + wd * p;  buf = first ? d : momentum * buf + d;  p -= lr * buf
__global__ __launch_bounds__(256) void sgd_clip_kernel(SgdArgs a) {
  __shared__ double red[4];
  double s = a.part[threadIdx.x];          // SGD_PARTS == blockDim.x
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float total = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) a.norm_out[0] = total;
  float coef = 1.f;
  if (a.max_norm > 0.f) coef = fminf(a.max_norm / (total + 1e-6f), 1.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    const float pv = a.p[i];
    const float gv = a.g[i] * coef;
    const float d = gv + a.weight_decay * pv;
    const float b = a.first_step ? d : a.momentum * a.buf[i] + d;
    a.g[i] = gv;
    a.buf[i] = b;
    a.p[i] = pv - a.lr * b;
  }
}

}  // namespace ragmi

extern "C" int64_t ragmi_sgd_workspace_bytes(void) { return (int64_t)ragmi::SGD_PARTS * (int64_t)sizeof(double); }

extern "C" int ragmi_sgd_clip_step(void* param, void* grad, void* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                                   float max_norm, int first_step, void* workspace, void* norm_out, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(param && grad && momentum_buf && workspace, RAGMI_EINVAL, "sgd_clip_step: null pointer");
  RAGMI_REQUIRE(n > 0, RAGMI_EINVAL, "sgd_clip_step: empty parameter buffer");
  RAGMI_REQUIRE(lr >= 0.f && momentum >= 0.f && weight_decay >= 0.f, RAGMI_EINVAL, "sgd_clip_step: negative hyper-parameter");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(SGD_PARTS), dim3(256), 0, st, (const float*)grad, n, (double*)workspace);
  SgdArgs a{(float*)param, (float*)grad, (float*)momentum_buf, (const double*)workspace, (float*)norm_out, n, lr, momentum,
            weight_decay, max_norm, first_step ? 1 : 0};
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(sgd_clip_kernel, dim3(blocks), dim3(256), 0, st, a);
  return check_launch("sgd_clip_step");
}

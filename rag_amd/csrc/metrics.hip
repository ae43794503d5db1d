// Loss + evaluation metrics of the stereo training / eval loops as ONE pass over (disp_est, disp_gt) with device-side
// accumulators.  Reference: approaches/rag.py:210-211, 418-430 (mask = 0 < gt < max_disp, masked
// smooth-L1) and utilstool/metrics.py:21-65 (EPE, D1, Thres-tau, each averaged over the images whose mask keeps at
// least 10 % of the gt > 0 pixels).  The reference does six boolean gathers and six .item() syncs per batch.
#include "common.h"

namespace ragmi {

constexpr int MET_N = 8;   // per image: n_mask, n_gt_pos, sum smooth-L1, sum |e|, n_D1, n_thr1, n_thr2, n_thr3

__global__ __launch_bounds__(256) void zero_floats_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void stereo_metrics_kernel(const float* __restrict__ est, const float* __restrict__ gt, int64_t hw,
                                                             float maxdisp, float* __restrict__ acc) {
  const int b = blockIdx.y;
  const float* pe = est + (int64_t)b * hw;
  const float* pg = gt + (int64_t)b * hw;
  float v[MET_N];
#pragma unroll
  for (int k = 0; k < MET_N; ++k) v[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    const float g = pg[i], d = pe[i];
    if (g > 0.f) v[1] += 1.f;
    if (g > 0.f && g < maxdisp) {
      const float e = fabsf(g - d);
      v[0] += 1.f;
      v[2] += e < 1.f ? 0.5f * e * e : e - 0.5f;                 // smooth-L1, beta = 1
      v[3] += e;
      if (e > 3.f && e / fabsf(g) > 0.05f) v[4] += 1.f;           // D1: > 3 px and > 5 %
      if (e > 1.f) v[5] += 1.f;
      if (e > 2.f) v[6] += 1.f;
      if (e > 3.f) v[7] += 1.f;
    }
  }
  __shared__ float red[4][MET_N];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < MET_N; ++k) {
    float s = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < MET_N) atomicAdd(acc + b * MET_N + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// out[0..5] = loss, EPE, D1, Thres1, Thres2, Thres3;  out[6] = number of masked pixels in the batch;  out[7] = images kept
__global__ void stereo_metrics_finalize_kernel(const float* __restrict__ acc, int B, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double n_all = 0.0, l_all = 0.0, m[5] = {0, 0, 0, 0, 0};
  int kept = 0;
  for (int b = 0; b < B; ++b) {
    const float* a = acc + b * MET_N;
    n_all += a[0];
    l_all += a[2];
    // metrics.py:30: an image is skipped when mask.mean() / (gt > 0).mean() < 0.1 (0/0 = nan compares false: kept, then nan)
    const bool skip = a[1] > 0.f ? (a[0] / a[1] < 0.1f) : false;
    if (skip) continue;
    ++kept;
    for (int k = 0; k < 5; ++k) m[k] += (double)a[3 + k] / (double)a[0];
  }
  out[0] = (float)(l_all / n_all);
  for (int k = 0; k < 5; ++k) out[1 + k] = kept ? (float)(m[k] / kept) : 0.f;
  out[6] = (float)n_all;
  out[7] = (float)kept;
}

// d loss / d est = gout * [mask] * clamp(est - gt, -1, 1) / n_mask   (n_mask = out[6] of the forward)
__global__ __launch_bounds__(256) void masked_smooth_l1_bwd_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                                   const float* __restrict__ out, const float* __restrict__ gout,
                                                                   float* __restrict__ dest, int64_t n, float maxdisp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float g = gt[i], k = gout[0] / out[6];
  float r = 0.f;
  if (g > 0.f && g < maxdisp) r = k * fminf(fmaxf(est[i] - g, -1.f), 1.f);
  dest[i] = r;
}

}  // namespace ragmi

extern "C" int ragmi_stereo_metrics_fwd(const void* disp_est, const void* disp_gt, int B, int H, int W, float maxdisp, void* acc,
                                        void* out, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(disp_est && disp_gt && acc && out, RAGMI_EINVAL, "stereo_metrics: null pointer");
  RAGMI_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535, RAGMI_EINVAL, "stereo_metrics: bad size");
  const int64_t hw = (int64_t)H * W;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // zeroed by a kernel, not hipMemsetAsync: memset / memcpy nodes of a captured hipGraph are corrupted by memcpys issued on the null
  // stream between replays, so nothing on the training path may become one
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)ceil_div((int64_t)MET_N * B, 256)), dim3(256), 0, st, (float*)acc, MET_N * B);
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(hw, 1024), 64));
  hipLaunchKernelGGL(stereo_metrics_kernel, dim3(gx, B), dim3(256), 0, st, (const float*)disp_est, (const float*)disp_gt, hw, maxdisp,
                     (float*)acc);
  hipLaunchKernelGGL(stereo_metrics_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)acc, B, (float*)out);
  return check_launch("stereo_metrics");
}

extern "C" int ragmi_masked_smooth_l1_bwd(const void* disp_est, const void* disp_gt, const void* out, const void* gout, void* ddisp,
                                          int B, int H, int W, float maxdisp, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(disp_est && disp_gt && out && gout && ddisp, RAGMI_EINVAL, "masked_smooth_l1_bwd: null pointer");
  RAGMI_REQUIRE(B > 0 && H > 0 && W > 0, RAGMI_EINVAL, "masked_smooth_l1_bwd: bad size");
  const int64_t n = (int64_t)B * H * W;
  hipLaunchKernelGGL(masked_smooth_l1_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)disp_est, (const float*)disp_gt, (const float*)out, (const float*)gout, (float*)ddisp, n, maxdisp);
  return check_launch("masked_smooth_l1_bwd");
}

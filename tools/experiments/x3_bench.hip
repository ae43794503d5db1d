// Standalone development bench for the bf16x3 convolution kernel (rag_amd/csrc/conv3d_x3.hip): instantiates chosen workgroup
// shapes directly, checks each against a naive fp32 GPU convolution (fp64 accumulate) and times it with HIP events, all in ONE
// process (A/B within a run).  No Python, no torch: starts in milliseconds on the GPU box.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -Iinclude -Irag_amd/csrc tools/x3_bench.hip -o tools/x3_bench
//   tools/x3_bench [case ...]      cases: l3dual stem1 l6dual train3 (default: all)
#define RAGMI_X3_NO_DISPATCH
#define RAGMI_X3_BENCH
#include "../../rag_amd/csrc/runtime.hip"
#include "conv3d_x3_planestationary.hip"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace ragmi;

#define CK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// naive reference: one thread per output voxel and channel, fp64 accumulate; out = sum over sets of act(scale * conv + shift)
__global__ void ref_conv_kernel(const float* x, const float* wA, const float* wB, const float* scA, const float* shA, const float* scB,
                                const float* shB, float* y, int B, int CinA, int CinB, int Cout, int D, int H, int W, int relu) {
  const int64_t DHW = (int64_t)D * H * W;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Cout * DHW) return;
  const int xx = idx % W, yy = (idx / W) % H, zz = (idx / ((int64_t)W * H)) % D, co = (idx / DHW) % Cout, b = idx / (DHW * Cout);
  const int Cx = CinA + CinB;
  double out = 0.0;
  for (int set = 0; set < (CinB > 0 ? 2 : 1); ++set) {
    const float* w = set ? wB : wA;
    const int cin = set ? CinB : CinA, c0 = set ? CinA : 0;
    double acc = 0.0;
    for (int ci = 0; ci < cin; ++ci)
      for (int dz = 0; dz < 3; ++dz)
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx) {
            const int z = zz + dz - 1, yv = yy + dy - 1, xv = xx + dx - 1;
            if ((unsigned)z >= (unsigned)D || (unsigned)yv >= (unsigned)H || (unsigned)xv >= (unsigned)W) continue;
            acc += (double)w[((co * cin + ci) * 3 + dz) * 9 + dy * 3 + dx] * (double)x[((int64_t)b * Cx + c0 + ci) * DHW + ((int64_t)z * H + yv) * W + xv];
          }
    const float* sc = set ? scB : scA;
    const float* sh = set ? shB : shA;
    double u = sc ? acc * sc[co] + sh[co] : acc;
    if (relu) u = u > 0 ? u : 0;
    out += u;
  }
  y[idx] = (float)out;
}

static float frand(uint64_t& s) {   // xorshift, uniform in [-1, 1)
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f;
}

struct Case { const char* name; int B, CinA, CinB, Cout, D, H, W; };

template <int NCG, int NSET, int NW, int ROWV, bool VEC>
static float run_variant(const char* label, K3Args a, int nseg, hipStream_t st, int iters, const float* yref, float* y, size_t ny,
                         std::vector<float>& hy, const std::vector<float>& href, int diag = 0) {
  X3Extra e{};
  e.diag = diag;
  dim3 grid;
  bool vec = false;
  if (x3_prepare(a, e, NSET, RAGMI_F32X3, nseg, grid, vec) != RAGMI_OK) { printf("  %s: prepare failed\n", label); return -1; }
  if (VEC && !vec) { printf("  %s: shape not quad-aligned\n", label); return -1; }
  const size_t lds = x3_lds_bytes_c(NCG, NSET, ROWV, false);
  CK_HIP(hipMemsetAsync(y, 0xff, ny * sizeof(float), st));
  if (x3_launch_final<float, NCG, NSET, false, NW, ROWV, VEC>(a, e, grid, lds, st) != RAGMI_OK) { printf("  %s: launch failed: %s\n", label, ragmi_last_error()); return -1; }
  CK_HIP(hipStreamSynchronize(st));
  CK_HIP(hipMemcpy(hy.data(), y, ny * sizeof(float), hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  for (size_t i = 0; i < ny; ++i) {
    const double d = std::fabs((double)hy[i] - (double)href[i]);
    if (!(d <= 1e30)) { maxerr = 1e30; break; }
    if (d > maxerr) maxerr = d;
    if (std::fabs(href[i]) > maxref) maxref = std::fabs(href[i]);
  }
  hipEvent_t e0, e1;
  CK_HIP(hipEventCreate(&e0)); CK_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) x3_launch_final<float, NCG, NSET, false, NW, ROWV, VEC>(a, e, grid, lds, st);
  CK_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) x3_launch_final<float, NCG, NSET, false, NW, ROWV, VEC>(a, e, grid, lds, st);
  CK_HIP(hipEventRecord(e1, st));
  CK_HIP(hipEventSynchronize(e1));
  float ms = 0;
  CK_HIP(hipEventElapsedTime(&ms, e0, e1));
  const float us = ms * 1e3f / iters;
  const double vox = (double)a.B * a.D * a.H * a.W;
  const double bytes = (a.Cin + a.Cout) * vox * 4.0, flops = 2.0 * 27 * (double)a.Cin / NSET * a.Cout * vox * NSET;
  if (diag & 16) {      // in-kernel stamps: one more launch with the stamp buffer, averaged over all waves
    const size_t nst = (size_t)grid.x * grid.y * NW * 16;
    unsigned long long* dst;
    CK_HIP(hipMalloc(&dst, nst * 8));
    CK_HIP(hipMemset(dst, 0, nst * 8));
    e.stamps = dst;
    x3_launch_final<float, NCG, NSET, false, NW, ROWV, VEC>(a, e, grid, lds, st);
    CK_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(nst);
    CK_HIP(hipMemcpy(h.data(), dst, nst * 8, hipMemcpyDeviceToHost));
    double sum[16] = {}, tot = 0;
    for (size_t i = 0; i < nst; ++i) { sum[i % 16] += (double)h[i]; tot += (double)h[i]; }
    const double nwaves = (double)grid.x * grid.y * NW;
    // steps per wave: columns per workgroup x (seg_len + 2)
    const double steps = (double)e.nwork / grid.x * (e.seg_len + 2);
    printf("  %-34s stamps (cycles per step per wave; %.1f steps/wave, total %.0f cycles/wave = %.1f us at 2.4 GHz):\n", label, steps, tot / nwaves, tot / nwaves / 2400.0);
    const char* names[10] = {"loop/epilogue tail", "barrier", "vmcnt wait", "commit+prefetch", "deferred epilogue", "operand read issue", "operand read wait", "full MFMA blocks", "leftover MFMAs", "last epilogue"};
    for (int k = 0; k < 10; ++k) printf("      %-20s %8.0f\n", names[k], sum[k] / nwaves / steps);
    fflush(stdout);
    hipFree(dst);
    return us;
  }
  if (diag) { printf("  %-34s diag %d: %8.1f us\n", label, diag, us); fflush(stdout); return us; }
  printf("  %-34s nseg %2d (len %2d) grid %4u x %u lds %6zu: %8.1f us  %6.2f TB/s alg  %6.1f TF/s  max|err| %.3e (ref max %.3e) %s\n", label, e.nseg,
         e.seg_len, grid.x, grid.y, lds, us, bytes / us * 1e-6, flops / us * 1e-6, maxerr, maxref, maxerr <= 3e-5 * maxref + 1e-6 ? "OK" : "MISMATCH");
  fflush(stdout);
  CK_HIP(hipEventDestroy(e0)); CK_HIP(hipEventDestroy(e1));
  return us;
}

template <int NCG, int NSET>
static void run_case(const Case& c, hipStream_t st, int nseg_arg) {
  const int64_t DHW = (int64_t)c.D * c.H * c.W;
  const int Cx = c.CinA + c.CinB;
  const size_t nx = (size_t)c.B * Cx * DHW, ny = (size_t)c.B * c.Cout * DHW;
  printf("%s: B=%d Cin=%d+%d Cout=%d %dx%dx%d (%.2f M voxels)\n", c.name, c.B, c.CinA, c.CinB, c.Cout, c.D, c.H, c.W, c.B * DHW * 1e-6);
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  std::vector<float> hx(nx), hwA((size_t)c.Cout * c.CinA * 27), hwB((size_t)c.Cout * std::max(c.CinB, 1) * 27), hs(4 * c.Cout), hy(ny), href(ny);
  for (auto& v : hx) v = frand(seed) * 2.f;
  for (auto& v : hwA) v = frand(seed) * 0.2f;
  for (auto& v : hwB) v = frand(seed) * 0.2f;
  for (int i = 0; i < c.Cout; ++i) { hs[i] = 1.f + 0.5f * frand(seed); hs[c.Cout + i] = 0.1f * frand(seed); hs[2 * c.Cout + i] = 1.f + 0.5f * frand(seed); hs[3 * c.Cout + i] = 0.1f * frand(seed); }
  float *x, *y, *yref, *wA, *wB, *sc, *pA, *pB;
  CK_HIP(hipMalloc(&x, nx * 4)); CK_HIP(hipMalloc(&y, ny * 4)); CK_HIP(hipMalloc(&yref, ny * 4));
  CK_HIP(hipMalloc(&wA, hwA.size() * 4)); CK_HIP(hipMalloc(&wB, hwB.size() * 4)); CK_HIP(hipMalloc(&sc, hs.size() * 4));
  CK_HIP(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(wA, hwA.data(), hwA.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(wB, hwB.data(), hwB.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  const int relu = 1;
  hipLaunchKernelGGL(ref_conv_kernel, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, st, x, wA, wB, sc, sc + c.Cout, sc + 2 * c.Cout,
                     sc + 3 * c.Cout, yref, c.B, c.CinA, c.CinB, c.Cout, c.D, c.H, c.W, relu);
  CK_HIP(hipStreamSynchronize(st));
  CK_HIP(hipMemcpy(href.data(), yref, ny * 4, hipMemcpyDeviceToHost));
  // packed weights: [fp32-MFMA section][bf16x3 fragments] per set, as ragmi_conv3d_k3_pack_ex lays them out
  auto pack = [&](const float* w, int cin, float** out) {
    const int64_t k3 = (int64_t)((c.Cout + 3) / 4) * ((cin + CK - 1) / CK) * PACK_PER_GC;
    CK_HIP(hipMalloc(out, (k3 + x3_packed_words(c.Cout, cin)) * 4));
    pack_both(w, *out, k3, c.Cout, cin, 0, 0, st);
  };
  pack(wA, c.CinA, &pA);
  if (c.CinB > 0) pack(wB, c.CinB, &pB); else pB = nullptr;
  K3Args a{};
  a.x = x; a.x_bstride = (int64_t)Cx * DHW; a.y = y; a.y_bstride = (int64_t)c.Cout * DHW; a.res = nullptr;
  a.B = c.B; a.Cin = Cx; a.Cout = c.Cout; a.D = c.D; a.H = c.H; a.W = c.W; a.relu = relu;
  a.wp[0] = pA; a.wp[1] = pB; a.scale[0] = sc; a.shift[0] = sc + c.Cout; a.scale[1] = sc + 2 * c.Cout; a.shift[1] = sc + 3 * c.Cout;
  a.nchunks[0] = (c.CinA + 3) / 4; a.nchunks[1] = (c.CinB + 3) / 4;
  a.store_main = 1; a.ntail = 0;
  for (int g = 0; g < (c.Cout + 3) / 4; ++g) a.y_ch[g] = 4 * g;
  const int iters = 20;
  for (int rep = 0; rep < 2; ++rep) {
    for (int nseg : {nseg_arg > 0 ? nseg_arg : 0, 4, 7}) {
      if (rep == 1 && nseg != 0 && nseg_arg <= 0) continue;
      run_variant<NCG, NSET, 4, 48, true>("4 waves, rows 48, quads", a, nseg, st, iters, yref, y, ny, hy, href);
      run_variant<NCG, NSET, 8, 48, true>("8 waves, rows 48, quads", a, nseg, st, iters, yref, y, ny, hy, href);
      if (nseg_arg > 0) break;
    }
    if (rep == 0) {
      run_variant<NCG, NSET, 4, 48, false>("4 waves, rows 48, single voxels", a, 0, st, iters, yref, y, ny, hy, href);
#ifdef RAGMI_X3_STAMPS
      for (int diag : {16}) {
#else
      for (int diag : {1, 2, 4, 3, 5, 6, 7}) {
#endif
        run_variant<NCG, NSET, 4, 48, true>("4 waves, rows 48, quads", a, 0, st, iters, yref, y, ny, hy, href, diag);
      }
    }
  }
  hipFree(x); hipFree(y); hipFree(yref); hipFree(wA); hipFree(wB); hipFree(sc); hipFree(pA); if (pB) hipFree(pB);
}

int main(int argc, char** argv) {
  hipStream_t st;
  CK_HIP(hipStreamCreate(&st));
  std::vector<std::string> want;
  int nseg = 0;
  for (int i = 1; i < argc; ++i) {
    if (!strncmp(argv[i], "nseg=", 5)) nseg = atoi(argv[i] + 5); else want.push_back(argv[i]);
  }
  auto on = [&](const char* n) { if (want.empty()) return true; for (auto& w : want) if (w == n) return true; return false; };
  if (on("small")) run_case<2, 2>(Case{"small (ragged: H, W not multiples of the tile)", 2, 4, 4, 12, 19, 45, 68}, st, nseg);
  if (on("l3dual")) run_case<2, 2>(Case{"level-3 dual cell", 1, 4, 4, 12, 64, 128, 416}, st, nseg);
  if (on("stem1")) run_case<3, 1>(Case{"stem3d1", 1, 12, 0, 12, 64, 128, 416}, st, nseg);
  if (on("l6dual")) run_case<4, 2>(Case{"level-6 dual cell", 1, 8, 8, 24, 32, 64, 208}, st, nseg);
  if (on("train3")) run_case<3, 1>(Case{"training level-3 (B=4)", 4, 12, 0, 12, 64, 64, 128}, st, nseg);
  return 0;
}

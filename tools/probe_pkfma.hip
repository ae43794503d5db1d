// Hardware probe (not product code): the instruction sequence hipcc 7.2 emitted for the 48-term tail chain of the fused stem kernel
// (conv3d_x3_kernel<float,3,1,1,2>, round 5, before the chain was made scalar) — v_pk_fma_f32 with op_sel broadcasts between
// broadcast ds_read_b128 and v_mov_b32 from SGPRs — replayed verbatim (tools/probes/pkfma_seq.h) against a scalar fmaf chain, with
// and without MFMA waves on the same SIMDs.  In the product kernel ~0.1 % of the results were wrong in the LOW halves, lanes 48..63.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_pkfma.hip -o tools/probe_pkfma && tools/probe_pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include "probes/pkfma_seq.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

// mode bit 0: waves 4..7 of a workgroup run MFMAs instead of the sequence; bit 1: waves 4..7 hammer the LDS with reads
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ in, unsigned* __restrict__ bad, unsigned* __restrict__ lanes, int iters, int mode,
                                             float4* __restrict__ out, float* __restrict__ scratch) {
  __shared__ __attribute__((aligned(16))) float4 tab[20 + 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 20) tab[tid] = make_float4(in[4 * tid] * 0.3f, in[4 * tid + 1] * 0.3f, in[4 * tid + 2] * 0.3f, in[4 * tid + 3] * 0.3f);
  if (tid >= 64 && tid < 128) tab[20 + tid - 64] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)tab;
  float acc = 0.f;
  if (wave >= 4 && (mode & 3)) {
    if (mode & 1) {
      f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      const float a = in[tid & 63], b = in[64 + (tid & 63)];
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
      }
      acc = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
      for (int it = 0; it < iters * 40; ++it) { const float4 q = tab[20 + ((lane + it) & 63)]; acc += q.x + q.w; }
    }
    if (acc == 12345.678f) bad[1] = 1;
    return;
  }
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float x[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) x[c] = fmaxf(in[256 + ((tid * 12 + c + it * 7) & 4095)], 0.f);
    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
    // (mode bit 3: the product kernel's execution mask — the interior voxels of a 10 x 34 halo plane)
    const bool active = !(mode & 8) || (tid < 340 && tid / 34 >= 1 && tid / 34 <= 8 && tid % 34 >= 1 && tid % 34 <= 32);
    if (active)
    // (mode bit 2: a burst of stores in front, as the product kernel's epilogue leaves behind)
    if (mode & 4) {
#pragma unroll
      for (int q = 0; q < 24; ++q) scratch[((size_t)q * gridDim.x + blockIdx.x) * 512 + tid] = x[q % 12];
    }
    float4* const slot = out + (size_t)(blockIdx.x * 512 + tid);
    asm volatile(PK_PRO PK_BODY
                 "global_store_dwordx4 %[slot], v[38:41], off\n"
                 "s_mov_b64 s[12:13], exec\n"
                 "s_or_b64 exec, exec, s[12:13]\n"
                 "s_or_b64 exec, exec, s[12:13]\n"
                 "s_cmp_lt_u32 s3, s10\n"
                 "s_cselect_b64 vcc, -1, 0\n"
                 "v_and_b32_e32 v39, 0x7fffffff, v27\n"
                 "v_cndmask_b32_e32 v0, 0, v26, vcc\n"
                 "v_cmp_gt_u32_e32 vcc, s3, v39\n"
                 "v_and_b32_e32 v38, 0x7fffffff, v26\n"
                 "v_cmp_gt_u32_e64 s[12:13], 4, v0\n"
                 "v_cndmask_b32_e32 v39, 0, v39, vcc\n"
                 "v_max_u32_e32 v40, v38, v39\n"
                 "v_cmp_gt_u32_e32 vcc, s3, v38\n"
                 "v_cndmask_b32_e32 v38, v39, v40, vcc\n"
                 "v_and_b32_e32 v40, 0x7fffffff, v29\n"
                 "v_max_u32_e32 v41, v39, v40\n"
                 "s_waitcnt vmcnt(0)\n"
                 "global_load_dwordx4 v[38:41], %[slot], off sc0 sc1\n"
                 "s_waitcnt vmcnt(0)\n"
                 PK_EPI
                 : [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3)
                 : [slot] "v"(slot), [base] "s"(base), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),
                   [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10]), [x11] "v"(x[11])
                 : "memory", "vcc", "s2", "s3", "s10", "s11", "s12", "s13", "s14", "s15", "s38", "s39", "s40", "s41", "s76", "s80", "s87", "s92", "s93", "s98", "s99",
                   "v0", "v1", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",
                   "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
                   "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v104", "v105",
                   "v120", "v121", "v122", "v123", "v126", "v127");
    // the same chain, one scalar fma per term
    float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const float4 w = tab[6 + c];
      r[0] = fmaf(w.x, x[c], r[0]); asm volatile("" : "+v"(r[0]));
      r[1] = fmaf(w.y, x[c], r[1]); asm volatile("" : "+v"(r[1]));
      r[2] = fmaf(w.z, x[c], r[2]); asm volatile("" : "+v"(r[2]));
      r[3] = fmaf(w.w, x[c], r[3]); asm volatile("" : "+v"(r[3]));
    }
    const float4 ts = tab[18], th = tab[19];
    r[0] = fmaxf(fmaf(r[0], ts.x, th.x), 0.f); r[1] = fmaxf(fmaf(r[1], ts.y, th.y), 0.f);
    r[2] = fmaxf(fmaf(r[2], ts.z, th.z), 0.f); r[3] = fmaxf(fmaf(r[3], ts.w, th.w), 0.f);
    if (!active) continue;
    const unsigned m = (u0 != r[0] ? 1u : 0u) | (u1 != r[1] ? 2u : 0u) | (u2 != r[2] ? 4u : 0u) | (u3 != r[3] ? 8u : 0u);
    if (m) { ++nbad; atomicAdd(&lanes[lane], 1u); atomicAdd(&lanes[64 + m], 1u); }
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

int main() {
  float* in; unsigned *bad, *lanes; float4* out; float* scratch;
  const int blocks = 512 * 4;
  CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&bad, 16)); CK(hipMalloc(&lanes, 128 * 4));
  CK(hipMalloc(&out, (size_t)blocks * 512 * 16)); CK(hipMalloc(&scratch, (size_t)24 * blocks * 512 * 4));
  static float h[8192];
  unsigned s = 12345u;
  for (int i = 0; i < 8192; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  for (int mode = 0; mode < 16; ++mode) {
    if ((mode & 3) == 3) continue;
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(lanes, 0, 128 * 4));
    const int iters = 200;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, in, bad, lanes, iters, mode, out, scratch);
    CK(hipDeviceSynchronize());
    unsigned hb[4], hl[128];
    CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, lanes, sizeof(hl), hipMemcpyDeviceToHost));
    const double total = (double)blocks * ((mode & 3) ? 256 : 512) * iters;
    printf("mode %2d (%s%s%s): %u wrong of %.0f evaluations (%.2e)\n", mode, (mode & 3) == 0 ? "all waves run the sequence" : (mode & 3) == 1 ? "waves 4..7 run MFMAs" : "waves 4..7 read the LDS", (mode & 4) ? ", 24 stores in front" : "", (mode & 8) ? ", interior-voxel exec mask" : "", hb[0], total, hb[0] / total);
    if (hb[0]) {
      printf("  wrong per lane quarter: ");
      for (int q = 0; q < 4; ++q) { unsigned t = 0; for (int l = 0; l < 16; ++l) t += hl[16 * q + l]; printf("%u ", t); }
      printf("\n  wrong-output masks (bit k = output k):");
      for (int m = 1; m < 16; ++m) if (hl[64 + m]) printf(" %d:%u", m, hl[64 + m]);
      printf("\n");
    }
  }
  return 0;
}

// Error plumbing + version for the C ABI (include/rag_amd.h).
#include "common.h"

namespace ragmi {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RAGMI_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RAGMI_OK;
}

}  // namespace ragmi

extern "C" int ragmi_version(void) { return 500; }   // round 5: + ragmi_costvol_stem_conv3d_fwd / _supported, ragmi_conv3d_k3_g4_caps, the G4 / RAGMI_TAIL_F32 / RAGMI_TAIL_ROWS / RAGMI_OUT_F32 flags (round 4: 400)

#include <vector>
extern "C" int ragmi_graph_node_census(void* graph, int32_t* n_kernel, int32_t* n_memcpy, int32_t* n_memset, int32_t* n_other) {
  using namespace ragmi;
  RAGMI_REQUIRE(graph && n_kernel && n_memcpy && n_memset && n_other, RAGMI_EINVAL, "graph_node_census: null pointer");
  hipGraph_t g = static_cast<hipGraph_t>(graph);
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) { (void)hipGetLastError(); return fail(RAGMI_ELAUNCH, "graph_node_census: hipGraphGetNodes failed"); }
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) { (void)hipGetLastError(); return fail(RAGMI_ELAUNCH, "graph_node_census: hipGraphGetNodes failed"); }
  *n_kernel = *n_memcpy = *n_memset = *n_other = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { (void)hipGetLastError(); return fail(RAGMI_ELAUNCH, "graph_node_census: hipGraphNodeGetType failed"); }
    switch (t) {
      case hipGraphNodeTypeKernel: ++*n_kernel; break;
      case hipGraphNodeTypeMemcpy: case hipGraphNodeTypeMemcpyFromSymbol: case hipGraphNodeTypeMemcpyToSymbol: ++*n_memcpy; break;
      case hipGraphNodeTypeMemset: ++*n_memset; break;
      default: ++*n_other; break;
    }
  }
  return RAGMI_OK;
}
extern "C" const char* ragmi_last_error(void) { return ragmi::error_buffer(); }

// conv3d_k3 instantiation: bf16 storage, x-tile 2^5, 4 rows/lane, 1 accumulator set(s)
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_s1_cfg0_bf16(const K3Args& a, int ngroups, hipStream_t s) {
  // depth-1 volumes (the 2-D Feature-Net convolutions): the tile form whose four waves split y instead of z
  return a.D == 1 ? launch_cfg<bf16_t, 5, 4, 1, 2, true>(a, ngroups, s) : launch_cfg<bf16_t, 5, 4, 1, 2>(a, ngroups, s);
}
}  // namespace ragmi

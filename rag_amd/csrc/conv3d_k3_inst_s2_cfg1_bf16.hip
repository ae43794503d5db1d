// conv3d_k3 instantiation: bf16 storage, x-tile 2^4, 2 rows/lane, 2 accumulator set(s)
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_s2_cfg1_bf16(const K3Args& a, int ngroups, hipStream_t s) { return launch_cfg<bf16_t, 4, 2, 2, 2>(a, ngroups, s); }
}  // namespace ragmi

// conv3d_k3 instantiation: f32 storage, x-tile 2^3, 1 rows/lane, 1 accumulator set(s)
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_s1_cfg2_f32(const K3Args& a, int ngroups, hipStream_t s) { return launch_cfg<float, 3, 1, 1, 2>(a, ngroups, s); }
}  // namespace ragmi

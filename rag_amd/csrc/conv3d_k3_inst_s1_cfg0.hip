// conv3d_k3 instantiation: x-tile 2^5, 4 rows/lane, 1 accumulator set(s), 2 wave(s)/SIMD register budget
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_s1_cfg0(const K3Args& a, int ngroups, hipStream_t s) { return launch_cfg<5, 4, 1, 2>(a, ngroups, s); }
}  // namespace ragmi

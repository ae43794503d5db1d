// conv3d_k3 instantiation: x-tile 2^5, 4 rows/lane, 2 accumulator set(s), 1 wave(s)/SIMD register budget
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_s2_cfg0b(const K3Args& a, int ngroups, hipStream_t s) { return launch_cfg<5, 4, 2, 1>(a, ngroups, s); }
}  // namespace ragmi

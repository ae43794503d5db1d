// conv3d_k3 instantiation: x-tile 2^4, 2 rows/lane, 2 accumulator set(s), 2 wave(s)/SIMD register budget
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_s2_cfg1(const K3Args& a, int ngroups, hipStream_t s) { return launch_cfg<4, 2, 2, 2>(a, ngroups, s); }
}  // namespace ragmi

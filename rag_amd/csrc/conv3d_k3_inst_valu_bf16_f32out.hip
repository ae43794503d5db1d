// conv3d_k3 VALU form (Cout <= 2, e.g. last_3_3d) reading bf16 activations and writing an fp32 result (`mat`)
#include "conv3d_k3.h"

namespace ragmi {
int launch_k3_valu_bf16_f32out(const K3Args& a, int cfg, hipStream_t s) {
  if (a.Cout == 1) {
    switch (cfg) {
      case 0: return launch_cfg_valu<bf16_t, 5, 4, 1, float>(a, s);
      case 1: return launch_cfg_valu<bf16_t, 4, 2, 1, float>(a, s);
      default: return launch_cfg_valu<bf16_t, 3, 1, 1, float>(a, s);
    }
  }
  switch (cfg) {
    case 0: return launch_cfg_valu<bf16_t, 5, 4, 2, float>(a, s);
    case 1: return launch_cfg_valu<bf16_t, 4, 2, 2, float>(a, s);
    default: return launch_cfg_valu<bf16_t, 3, 1, 2, float>(a, s);
  }
}
}  // namespace ragmi

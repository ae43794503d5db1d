"""rag_amd — MI355X-native (gfx950) implementation of the stereo Matching-Net forward path
of chzhang18/RAG: cost-volume build -> 3-D conv aggregation -> soft-argmin disparity
regression (and its training step), behind the reference's own nn.Module interface (src/models/rag_model.py).

Host code is PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on
the path runs in hand-written HIP kernels reached through the C ABI of
``rag_amd/lib/librag_amd.so`` (include/rag_amd.h).  There is no CPU or eager fallback: the
ops raise if the library is missing.
"""
from ._lib import lib_path, load_library  # noqa: F401
from . import ops  # noqa: F401
from . import autograd  # noqa: F401
from . import metrics  # noqa: F401
from . import train  # noqa: F401
from .modules import (  # noqa: F401
    ALL_CONV_GENOTYPE, ALL_SKIP_GENOTYPE, Cell_3d, ConvBR_3d, Disp, DisparityRegression, Genotype,
    Identity_3d, MatchingNet, OPS_3d, PRIMITIVES_3D,
)

from .modules import Cell_2d, ConvBR_2d, OPS_2d, PRIMITIVES  # noqa: E402,F401
from .network import Network  # noqa: E402,F401
from . import checkpoint  # noqa: E402,F401
from . import supernet  # noqa: E402,F401
from .supernet import AutoFeature, AutoMatching, BasicNetwork  # noqa: E402,F401

__all__ = [
    "Network", "Cell_2d", "ConvBR_2d",
    "ops", "load_library", "lib_path", "MatchingNet", "Cell_3d", "ConvBR_3d", "Identity_3d", "Disp",
    "DisparityRegression", "OPS_3d", "PRIMITIVES_3D", "Genotype", "ALL_CONV_GENOTYPE", "ALL_SKIP_GENOTYPE",
]

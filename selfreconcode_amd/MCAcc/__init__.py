from .grid_sampler_mine import GridSamplerMine3dFunction, GridSamplerMine3dBackwardFunction  # noqa: F401
from .seg3d_lossless import Seg3dLossless, create_grid3D  # noqa: F401

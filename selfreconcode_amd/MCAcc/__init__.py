from .grid_sampler_mine import GridSamplerMine3dFunction, GridSamplerMine3dBackwardFunction  # noqa: F401

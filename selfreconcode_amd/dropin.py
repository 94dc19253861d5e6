"""Registers this package's HIP-backed modules under the import names the reference uses for its
pybind/CUDA extensions, so that `import FastMinv`, `import GridSamplerMine`, `import MCGpu` inside the
reference's own files resolve here."""
import sys


def install():
    from .ext import FastMinv, GridSamplerMine, MCGpu, interp2x_boundary3d
    sys.modules['interp2x_boundary3d'] = interp2x_boundary3d
    sys.modules['FastMinv'] = FastMinv
    sys.modules['GridSamplerMine'] = GridSamplerMine
    sys.modules['MCGpu'] = MCGpu
    return FastMinv, GridSamplerMine, MCGpu

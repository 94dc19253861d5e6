"""Python faces of the reference's four pybind extension modules (FastMinv, GridSamplerMine,
MCGpu, interp2x_boundary3d), re-implemented over the C ABI of libselfrecon_hip.so.
`selfreconcode_amd.dropin.install()` registers them under the reference's import names."""

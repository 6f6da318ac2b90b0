"""MI355X-native supergrid generator: the coordinate-transform and metrics hot path of
nikizadehgfdl/ocean_model_grid_generator as hand-written gfx950 HIP kernels behind a ctypes C ABI.

    ocean_grid_generator   host mirror of the reference's functions, CLI and NetCDF layout
    supergrid              device-resident, latitude-band-sharded pass (bench / multi-GPU)
    _lib                   ctypes binding of csrc/libogg_hip.so (include/ogg_hip.h)
"""
__all__ = ["ocean_grid_generator", "supergrid", "_lib"]

"""mi3d - MI355X-native implementation of Make-It-3D's coarse-stage SDS hot path (host-side Python).

    raymarching/   drop-in for the reference's `raymarching` operator package
    tinycudann/    drop-in for the `tcnn.Encoding` HashGrid the reference uses
    mi3d/          fused field, renderer (NeRFRenderer surface), ray provider, data-parallel wrapper,
                   SD guidance stand-in
All device work goes through libmi3d.so (csrc/, C ABI in include/mi3d.h).
"""
__version__ = "0.1.0"

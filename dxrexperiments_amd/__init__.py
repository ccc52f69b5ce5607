"""MI355X-native progressive ray tracer: a drop-in for the hot path of
philcn/DXRExperiments' ProgressiveRaytracingPipeline.  The compute path is the
HIP library in dxrexperiments_amd/lib (C ABI: include/dxr_amd.h); this package
holds only its ctypes binding, the record dtypes and synthetic-scene builders.
"""

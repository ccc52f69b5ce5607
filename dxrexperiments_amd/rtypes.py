"""numpy views of the C-ABI records in include/dxr_amd_types.h.

Field names and byte layout restate the reference's shared host/device header
(assets/shaders/RaytracingHlslCompat.h:35-96).
"""
import numpy as np

F4 = ("<f4", 4)

VERTEX = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3)])

CAMERA_PARAMS = np.dtype([
    ("worldEyePos", *F4), ("U", *F4), ("V", *F4), ("W", *F4),
    ("jitters", "<f4", 2), ("frameCount", "<u4"), ("accumCount", "<u4")])

DIRECTIONAL_LIGHT = np.dtype([("forwardDir", *F4), ("color", *F4)])
POINT_LIGHT = np.dtype([("worldPos", *F4), ("color", *F4)])

DEBUG_OPTIONS = np.dtype([
    ("maxIterations", "<u4"), ("cosineHemisphereSampling", "<u4"), ("showIndirectDiffuseOnly", "<u4"),
    ("showIndirectSpecularOnly", "<u4"), ("showAmbientOcclusionOnly", "<u4"), ("showGBufferAlbedoOnly", "<u4"),
    ("showDirectLightingOnly", "<u4"), ("showFresnelTerm", "<u4"), ("noIndirectDiffuse", "<u4"),
    ("environmentStrength", "<f4"), ("debug", "<u4")])

PER_FRAME_CONSTANTS = np.dtype([
    ("cameraParams", CAMERA_PARAMS), ("directionalLight", DIRECTIONAL_LIGHT),
    ("pointLight", POINT_LIGHT), ("options", DEBUG_OPTIONS)])

MATERIAL_PARAMS = np.dtype([
    ("albedo", *F4), ("specular", *F4), ("emissive", *F4),
    ("reflectivity", "<f4"), ("roughness", "<f4"), ("IoR", "<f4"), ("type", "<u4")])

BVH_NODE = np.dtype([("bmin", "<f4", 3), ("left", "<u4"), ("bmax", "<f4", 3), ("right", "<u4")])

assert VERTEX.itemsize == 24 and CAMERA_PARAMS.itemsize == 80 and DEBUG_OPTIONS.itemsize == 44
assert PER_FRAME_CONSTANTS.itemsize == 188 and MATERIAL_PARAMS.itemsize == 64 and BVH_NODE.itemsize == 32

RT_LEAF = 0xFFFFFFFF
RT_NO_HIT = 0xFFFFFFFF
RAY_FLAG_NONE = 0x00
RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH = 0x04
RAY_FLAG_SKIP_CLOSEST_HIT_SHADER = 0x08
RAY_FLAG_CULL_BACK_FACING_TRIANGLES = 0x10
FORMAT_R32G32B32A32_FLOAT = 2
FORMAT_R16G16B16A16_FLOAT = 10
ROUND_NEAREST_EVEN, ROUND_TOWARD_ZERO = 0, 1          # rt_pipeline_set_accumulation_storage
ACCUM_RUNNING_MEAN = 0
ACCUM_SUM = 1

IDENTITY_3X4 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


def default_material():
    """The one material the reference app creates (src/DXRExperimentsApp.cpp:95-104)."""
    m = np.zeros((), MATERIAL_PARAMS)
    m["albedo"] = (0.95, 0.05, 0.0, 1.0)
    m["specular"] = (0.58, 0.58, 0.58, 1.0)
    m["roughness"] = 0.5
    m["reflectivity"] = 0.7
    m["type"] = 1
    return m


def default_options():
    """ProgressiveRaytracingPipeline ctor defaults (src/ProgressiveRaytracingPipeline.cpp:74-84)."""
    o = np.zeros((), DEBUG_OPTIONS)
    o["maxIterations"] = 1024
    o["cosineHemisphereSampling"] = 1
    o["environmentStrength"] = 1.0
    return o

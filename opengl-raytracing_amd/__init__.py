"""Python host mirror of librt_mi355.so (ctypes over the C ABI of include/rt_mi355.h).

The product is the shared library (HIP kernels + C++17 host code).  This module is plumbing for
tests and bench.py: struct mirrors, thin call wrappers, and a `Renderer` that walks the same steps
as the reference's `Application::mainLoop` -> `renderRay` (src/app/application.cpp:381-459,
src/render/render.cpp:55-243).  It never renders on the CPU: every frame goes through
`rt_render_frame` on a gfx950 device, and loading fails loudly when the library is not built.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import meshgen  # noqa: F401  (procedural scene inputs)

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "librt_mi355.so"

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_STATE, RT_ERR_UNSUPPORTED, RT_ERR_IO = -1, -2, -3, -4, -5, -6
RT_TARGET_COLOR, RT_TARGET_MOTION, RT_TARGET_GPOS, RT_TARGET_GNRM = 0, 1, 2, 3
RT_FORMAT_F16, RT_FORMAT_F32 = 0, 1
RT_PIPELINE_AUTO, RT_PIPELINE_MEGAKERNEL, RT_PIPELINE_WAVEFRONT = 0, 1, 2
TARGET_CHANNELS = {0: 4, 1: 2, 2: 4, 3: 4}
RT_MAX_STAGES = 14
RT_COMM_ID_BYTES = 128

f32, i32 = C.c_float, C.c_int32


def _fields(spec):
    out = []
    for name, kind in spec:
        if isinstance(kind, tuple):
            out.append((name, kind[0] * kind[1]))
        else:
            out.append((name, kind))
    return out


class _Struct(C.Structure):
    def to_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d

    def copy(self):
        other = type(self)()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(self))
        return other


class RtUniforms(_Struct):  # include/rt_mi355.h RtUniforms == shaders/rt/rt_uniforms.glsl:25-177
    _fields_ = _fields([
        ("eps", f32), ("pi", f32), ("inf", f32),
        ("camPos", (f32, 3)), ("camRight", (f32, 3)), ("camUp", (f32, 3)), ("camFwd", (f32, 3)),
        ("tanHalfFov", f32), ("aspect", f32),
        ("frameIndex", i32), ("spp", i32),
        ("resolution", (f32, 2)), ("jitter", (f32, 2)), ("enableJitter", i32),
        ("useBVH", i32), ("nodeCount", i32), ("triCount", i32), ("showMotion", i32),
        ("prevViewProj", (f32, 16)), ("currViewProj", (f32, 16)), ("cameraMoved", i32),
        ("taaStillThresh", f32), ("taaHardMovingThresh", f32),
        ("taaHistoryMinWeight", f32), ("taaHistoryAvgWeight", f32), ("taaHistoryMaxWeight", f32), ("taaHistoryBoxSize", f32),
        ("enableTAA", i32),
        ("giScaleAnalytic", f32), ("giScaleBVH", f32), ("enableGI", i32), ("enableAO", i32), ("aoSamples", i32),
        ("aoRadius", f32), ("aoBias", f32), ("aoMin", f32),
        ("useEnvMap", i32), ("envIntensity", f32),
        ("sunEnabled", i32), ("sunColor", (f32, 3)), ("sunIntensity", f32), ("sunDir", (f32, 3)),
        ("skyEnabled", i32), ("skyColor", (f32, 3)), ("skyIntensity", f32), ("skyUpDir", (f32, 3)),
        ("pointLightEnabled", i32), ("pointLightPos", (f32, 3)), ("pointLightColor", (f32, 3)), ("pointLightIntensity", f32),
        ("matAlbedoColor", (f32, 3)), ("matAlbedoSpecStrength", f32), ("matAlbedoGloss", f32),
        ("matGlassAlbedo", (f32, 3)), ("matGlassIOR", f32), ("matGlassDistortion", f32), ("matGlassEnabled", i32),
        ("matMirrorAlbedo", (f32, 3)), ("matMirrorGloss", f32), ("matMirrorEnabled", i32),
    ])


class RtRenderParams(_Struct):  # include/render/RenderParams.h:14-239
    _fields_ = _fields([
        ("sppPerFrame", i32), ("exposure", f32),
        ("matAlbedoColor", (f32, 3)), ("matAlbedoSpecStrength", f32), ("matAlbedoGloss", f32),
        ("matGlassEnabled", i32), ("matGlassColor", (f32, 3)), ("matGlassIOR", f32), ("matGlassDistortion", f32),
        ("matMirrorEnabled", i32), ("matMirrorColor", (f32, 3)), ("matMirrorGloss", f32),
        ("enableJitter", i32), ("jitterStillScale", f32), ("jitterMovingScale", f32),
        ("enableGI", i32), ("giScaleAnalytic", f32), ("giScaleBVH", f32),
        ("enableEnvMap", i32), ("envMapIntensity", f32),
        ("sunEnabled", i32), ("sunColor", (f32, 3)), ("sunIntensity", f32), ("sunYaw", f32), ("sunPitch", f32),
        ("skyEnabled", i32), ("skyColor", (f32, 3)), ("skyIntensity", f32), ("skyYaw", f32), ("skyPitch", f32),
        ("pointLightEnabled", i32), ("pointLightColor", (f32, 3)), ("pointLightIntensity", f32), ("pointLightPos", (f32, 3)),
        ("pointLightOrbitEnabled", i32), ("pointLightOrbitRadius", f32), ("pointLightOrbitSpeed", f32),
        ("pointLightYaw", f32), ("pointLightPitch", f32),
        ("enableAO", i32), ("aoSamples", i32), ("aoRadius", f32), ("aoBias", f32), ("aoMin", f32),
        ("enableTAA", i32), ("taaStillThresh", f32), ("taaHardMovingThresh", f32), ("taaHistoryMinWeight", f32),
        ("taaHistoryAvgWeight", f32), ("taaHistoryMaxWeight", f32), ("taaHistoryBoxSize", f32),
        ("enableSVGF", i32), ("svgfVarMax", f32), ("svgfKVar", f32), ("svgfKColor", f32), ("svgfKVarMotion", f32),
        ("svgfKColorMotion", f32), ("svgfStrength", f32),
        ("motionScale", f32),
    ])


class RtCamera(_Struct):
    _fields_ = _fields([("pos", (f32, 3)), ("yaw", f32), ("pitch", f32), ("fov", f32), ("aspect", f32)])


class RtCounters(_Struct):
    _fields_ = [(n, C.c_uint64) for n in ("raysClosest", "raysShadow", "raysAnalytic", "nodeFetch", "triFetch", "envLookup", "hitPixels",
                                             "fetchPrimary", "fetchShadow", "fetchAO")]

    @property
    def rays(self):
        return self.raysClosest + self.raysShadow + self.raysAnalytic


class RtDeviceConfig(_Struct):
    _fields_ = _fields([("device", i32), ("rank", i32), ("worldSize", i32), ("pipeline", i32), ("countWork", i32), ("reserved", (i32, 3))])


class RtStageTimes(_Struct):
    _fields_ = [("nStages", i32), ("frames", i32), ("ms", C.c_double * RT_MAX_STAGES), ("launches", C.c_uint64 * RT_MAX_STAGES)]


class RtCommInfo(_Struct):
    _fields_ = [(n, i32) for n in ("commWorld", "commRank", "rank", "worldSize")] + [(n, C.c_uint64) for n in ("gathers", "gatherBytes", "historyExchanges")]


class RtTracedRays(_Struct):
    _fields_ = [(n, C.c_uint64) for n in ("candidatePixels", "hitPixels", "primary", "shadow", "bounce", "bounceShadow", "frames",
                                             "gatherLoadsPrimary", "gatherLoadsShadow", "gatherLoadsBounce",
                                             "mergedLoadsPrimary", "mergedLoadsShadow", "mergedLoadsBounce", "ao", "gatherLoadsAO")]

    @property
    def rays(self):
        return self.primary + self.shadow + self.bounce + self.bounceShadow + self.ao


class RtSceneInfo(_Struct):
    _fields_ = [(n, i32) for n in ("nNodes", "nTris", "nInner", "treeDepth", "nWide4", "nPairs")] + \
               [(n, C.c_uint64) for n in ("bytesNodes2", "bytesNodes4", "bytesPairs", "bytesTris")] + \
               [("nFused", i32), ("flags", i32), ("implicitDepth", i32), ("reserved", i32)]


class RtMemoryInfo(_Struct):
    _fields_ = [(n, C.c_uint64) for n in ("queueArenaBytes", "frameArrayBytes", "hybridArenaBytes", "deviceFreeBytes", "deviceTotalBytes")] + \
               [(n, i32) for n in ("queueArenas", "lanes")]


class RtExtension(_Struct):
    _fields_ = _fields([("giBounces", i32), ("envFilter", i32), ("reserved", (i32, 2))])


RT_SCENE_HYBRID = 2   # RtUniforms.useBVH: the analytic scene + the BVH mesh (extension, not in the reference)
RT_SCENE_QNODES_REJECTED, RT_SCENE_NOT_FUSED, RT_SCENE_IMPLICIT = 1, 2, 4   # RtSceneInfo.flags


class RtPresentParams(_Struct):  # uniforms of shaders/rt/rt_present.frag:38-50
    _fields_ = _fields([("exposure", f32), ("showMotion", i32), ("motionScale", f32), ("resolution", (f32, 2)), ("varMax", f32), ("kVar", f32),
                        ("kColor", f32), ("kVarMotion", f32), ("kColorMotion", f32), ("svgfStrength", f32), ("enableSVGF", i32)])


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt_mi355 error {code}: {msg}")
        self.code = code


_lib = None
_FP = C.POINTER(C.c_float)
_U8P = C.POINTER(C.c_uint8)
_U32P = C.POINTER(C.c_uint32)

# name -> (restype, argtypes); this is also the list of symbols include/rt_mi355.h declares.
SIGNATURES = {
    "rt_create": (C.c_int, [C.POINTER(RtDeviceConfig), C.POINTER(C.c_void_p)]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_upload_bvh": (C.c_int, [C.c_void_p, _FP, C.c_int, _FP, C.c_int]),
    "rt_build_bvh_gpu": (C.c_int, [C.c_void_p, _FP, C.c_int, _FP, _FP]),
    "rt_upload_env": (C.c_int, [C.c_void_p, _U8P, C.c_int, C.c_int]),
    "rt_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_reset_accum": (C.c_int, [C.c_void_p]),
    "rt_frame_index": (C.c_int, [C.c_void_p]),
    "rt_render_frame": (C.c_int, [C.c_void_p, C.POINTER(RtUniforms)]),
    "rt_render_frames": (C.c_int, [C.c_void_p, C.POINTER(RtUniforms), C.c_int]),
    "rt_render_ray": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.POINTER(RtCamera), C.c_int, C.c_int, _FP, _FP]),
    "rt_set_extension": (C.c_int, [C.c_void_p, C.POINTER(RtExtension)]),
    "rt_render_ray_frames": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.POINTER(RtCamera), C.c_int, C.c_int, C.c_int]),
    "rt_synchronize": (C.c_int, [C.c_void_p]),
    "rt_read_target": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rt_write_target": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "rt_make_present_params": (None, [C.POINTER(RtRenderParams), C.c_int, C.c_int, C.c_int, C.POINTER(RtPresentParams)]),
    "rt_present": (C.c_int, [C.c_void_p, C.POINTER(RtPresentParams), _U8P]),
    "rt_history_exchange_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_history_exchanged": (C.c_int, [C.c_void_p]),
    "rt_present_gathered": (C.c_int, [C.c_void_p, C.POINTER(RtPresentParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _U8P]),
    "rt_local_target": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_gather_block_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "rt_assemble_gathered": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rt_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rt_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rt_comm_destroy": (C.c_int, [C.c_void_p]),
    "rt_gather_frame": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_gathered_frame": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_read_gathered": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "rt_present_last_gathered": (C.c_int, [C.c_void_p, C.POINTER(RtPresentParams), _U8P]),
    "rt_exchange_history": (C.c_int, [C.c_void_p]),
    "rt_comm_info": (C.c_int, [C.c_void_p, C.POINTER(RtCommInfo)]),
    "rt_get_counters": (C.c_int, [C.c_void_p, C.POINTER(RtCounters)]),
    "rt_reset_counters": (C.c_int, [C.c_void_p]),
    "rt_get_scene_info": (C.c_int, [C.c_void_p, C.POINTER(RtSceneInfo)]),
    "rt_get_memory_info": (C.c_int, [C.c_void_p, C.POINTER(RtMemoryInfo)]),
    "rt_get_traced_rays": (C.c_int, [C.c_void_p, C.POINTER(RtTracedRays), C.c_int]),
    "rt_enable_stage_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_get_stage_times": (C.c_int, [C.c_void_p, C.POINTER(RtStageTimes)]),
    "rt_stage_name": (C.c_char_p, [C.c_int]),
    "rt_debug_eval": (C.c_int, [C.c_void_p, C.c_int, _FP, _FP, _FP, _U32P, C.c_int]),
    "rt_debug_trace": (C.c_int, [C.c_void_p, C.c_int, _FP, _FP, _FP, C.c_float, C.c_float, _FP, C.c_int]),
    "rt_default_render_params": (None, [C.POINTER(RtRenderParams)]),
    "rt_default_camera": (None, [C.POINTER(RtCamera)]),
    "rt_default_bvh_transform": (None, [_FP]),
    "rt_camera_view": (None, [C.POINTER(RtCamera), _FP]),
    "rt_camera_proj": (None, [C.POINTER(RtCamera), _FP]),
    "rt_mat4_mul": (None, [_FP, _FP, _FP]),
    "rt_generate_jitter": (None, [C.c_int, _FP]),
    "rt_camera_moved": (C.c_int, [_FP, _FP]),
    "rt_make_uniforms": (None, [C.POINTER(RtRenderParams), C.POINTER(RtCamera), _FP, _FP, _FP] + [C.c_int] * 9 + [C.POINTER(RtUniforms)]),
    "rt_gather_triangles": (C.c_int, [_FP, _U32P, C.c_int, _FP, _FP]),
    "rt_gather_triangles_checked": (C.c_int, [_FP, C.c_int, _U32P, C.c_int, _FP, _FP]),
    "rt_build_bvh": (C.c_int, [_FP, C.c_int, _FP, _FP]),
    "rt_load_obj": (C.c_int, [C.c_char_p, C.POINTER(_FP), C.POINTER(C.c_int), C.POINTER(_U32P), C.POINTER(C.c_int)]),
    "rt_load_png": (C.c_int, [C.c_char_p, C.POINTER(_U8P), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_save_png": (C.c_int, [C.c_char_p, _U8P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rt_free": (None, [C.c_void_p]),
    "rt_cubemap_from_cross": (C.c_int, [_U8P, C.c_int, C.c_int, C.c_int, _U8P]),
    "rt_sizeof_uniforms": (C.c_int, []),
    "rt_sizeof_render_params": (C.c_int, []),
    "rt_version": (C.c_char_p, []),
}


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same
    SONAMEs as /opt/rocm's, found through libtorch's RPATH under a different file name).  If librt_mi355.so pulled in
    /opt/rocm's copy first, a later `import torch` (FrameGatherer, a user's own code) would load the bundled copy as a
    second runtime and fail with "No HIP GPUs are available".  When a torch wheel with bundled libraries is installed,
    load those first (no `import torch`); librt_mi355.so's NEEDED entries then bind to them by SONAME."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    d = Path(list(spec.submodule_search_locations)[0]) / "lib"
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        f = d / name
        if f.exists():
            try:
                C.CDLL(str(f), mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load librt_mi355.so (built by `make -C opengl-raytracing_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RtError(RT_ERR_NO_DEVICE, f"{LIB_PATH} is not built; run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                                        "There is no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    assert L.rt_sizeof_uniforms() == C.sizeof(RtUniforms), "RtUniforms layout drifted from include/rt_mi355.h"
    assert L.rt_sizeof_render_params() == C.sizeof(RtRenderParams), "RtRenderParams layout drifted"
    _lib = L
    return L


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: 128 bytes for rank 0 to hand to every rank's Renderer.comm_init."""
    buf = C.create_string_buffer(RT_COMM_ID_BYTES)
    rc = lib().rt_comm_unique_id(buf, RT_COMM_ID_BYTES)
    if rc != RT_OK:
        raise RtError(rc, (lib().rt_last_error(None) or b"").decode())
    return buf.raw


def _fp(a):
    return a.ctypes.data_as(_FP)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------------------------ host side
def default_render_params() -> RtRenderParams:
    p = RtRenderParams()
    lib().rt_default_render_params(C.byref(p))
    return p


def default_camera() -> RtCamera:
    c = RtCamera()
    lib().rt_default_camera(C.byref(c))
    return c


def closeup_camera() -> RtCamera:
    """Second camera of SURVEY.md 8d config 2: the mesh fills ~45 % of a 16:9 frame."""
    c = default_camera()
    c.pos[0], c.pos[1], c.pos[2] = -2.0, 1.5, 1.0
    c.yaw, c.pitch = -90.0, 0.0
    return c


def default_bvh_transform() -> np.ndarray:
    m = np.zeros(16, np.float32)
    lib().rt_default_bvh_transform(_fp(m))
    return m


def camera_view(cam) -> np.ndarray:
    m = np.zeros(16, np.float32)
    lib().rt_camera_view(C.byref(cam), _fp(m))
    return m


def camera_proj(cam) -> np.ndarray:
    m = np.zeros(16, np.float32)
    lib().rt_camera_proj(C.byref(cam), _fp(m))
    return m


def mat4_mul(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    m = np.zeros(16, np.float32)
    lib().rt_mat4_mul(_fp(a), _fp(b), _fp(m))
    return m


def generate_jitter(frame_index: int) -> np.ndarray:
    j = np.zeros(2, np.float32)
    lib().rt_generate_jitter(frame_index, _fp(j))
    return j


def camera_moved(curr_vp, prev_vp) -> bool:
    a, b = _f32(curr_vp), _f32(prev_vp)
    return bool(lib().rt_camera_moved(_fp(a), _fp(b)))


def make_uniforms(params, cam, view, curr_vp, prev_vp, w, h, frame_index=0, camera_moved=False, use_bvh=False,
                  show_motion=False, node_count=0, tri_count=0, env_loaded=True) -> RtUniforms:
    u = RtUniforms()
    v, c, p = _f32(view), _f32(curr_vp), _f32(prev_vp)
    lib().rt_make_uniforms(C.byref(params), C.byref(cam), _fp(v), _fp(c), _fp(p), int(w), int(h), int(frame_index),
                           int(camera_moved), int(use_bvh), int(show_motion), int(node_count), int(tri_count), int(env_loaded),
                           C.byref(u))
    return u


def make_present_params(params, show_motion, w, h) -> RtPresentParams:
    pp = RtPresentParams()
    lib().rt_make_present_params(C.byref(params), int(show_motion), int(w), int(h), C.byref(pp))
    return pp


def gather_triangles(positions, indices, model=None) -> np.ndarray:
    pos = _f32(positions).reshape(-1)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    m = default_bvh_transform() if model is None else _f32(model)
    out = np.zeros((idx.size // 3, 9), np.float32)
    n = lib().rt_gather_triangles_checked(_fp(pos), pos.size // 3, idx.ctypes.data_as(_U32P), idx.size, _fp(m), _fp(out))
    if n < 0:
        raise RtError(n, "rt_gather_triangles: index out of range" if n == RT_ERR_INVALID else "rt_gather_triangles")
    return out[:n]


def build_bvh(tris9):
    """-> (nodes12 [nNodes,12], tris12 [nTris,12]) in the reference's texture-buffer layout."""
    t = _f32(tris9).reshape(-1, 9)
    n = t.shape[0]
    nodes = np.zeros((max(2 * n, 1), 12), np.float32)
    tris = np.zeros((max(n, 1), 12), np.float32)
    k = lib().rt_build_bvh(_fp(t), n, _fp(nodes), _fp(tris))
    if k < 0:
        raise RtError(k, "rt_build_bvh")
    return nodes[:k].copy(), tris[:n].copy()


def load_obj(path):
    pos, idx = _FP(), _U32P()
    nv, ni = C.c_int(), C.c_int()
    rc = lib().rt_load_obj(str(path).encode(), C.byref(pos), C.byref(nv), C.byref(idx), C.byref(ni))
    if rc != RT_OK:
        raise RtError(rc, f"rt_load_obj({path})")
    p = np.ctypeslib.as_array(pos, shape=(max(nv.value * 3, 1),))[: nv.value * 3].copy().reshape(-1, 3)
    i = np.ctypeslib.as_array(idx, shape=(max(ni.value, 1),))[: ni.value].copy()
    lib().rt_free(pos)
    lib().rt_free(idx)
    return p, i


def load_png(path) -> np.ndarray:
    pix = _U8P()
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    rc = lib().rt_load_png(str(path).encode(), C.byref(pix), C.byref(w), C.byref(h), C.byref(ch))
    if rc != RT_OK:
        raise RtError(rc, f"rt_load_png({path})")
    a = np.ctypeslib.as_array(pix, shape=(h.value * w.value * ch.value,)).copy().reshape(h.value, w.value, ch.value)
    lib().rt_free(pix)
    return a


def save_png(path, img, flip_y=False):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    rc = lib().rt_save_png(str(path).encode(), a.ctypes.data_as(_U8P), w, h, ch, int(flip_y))
    if rc != RT_OK:
        raise RtError(rc, f"rt_save_png({path})")


def cubemap_from_cross(img) -> np.ndarray:
    """HxWxC uint8 4x3 cross -> [6, N, N, C] faces in GL order (src/render/cubemap.cpp:86-91)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, ch = img.shape
    n = h // 3 if h % 3 == 0 else 0
    faces = np.zeros((6, max(n, 1), max(n, 1), ch), np.uint8)
    got = lib().rt_cubemap_from_cross(img.ctypes.data_as(_U8P), w, h, ch, faces.ctypes.data_as(_U8P))
    if got == 0:
        raise RtError(RT_ERR_INVALID, f"not a 4x3 cube-map cross: {w}x{h}")
    return faces


def load_cubemap_cross(path) -> np.ndarray:
    return cubemap_from_cross(load_png(path))


ASSET_DIR = _PKG_DIR.parent / "assets"


# ------------------------------------------------------------------------------------ device side
class Renderer:
    """One RtContext.  Mirrors the reference's per-frame call sequence."""

    def __init__(self, device=0, rank=0, world_size=1, pipeline=RT_PIPELINE_AUTO, count_work=False):
        self._h = C.c_void_p()
        cfg = RtDeviceConfig(device=device, rank=rank, worldSize=world_size, pipeline=pipeline, countWork=int(count_work))
        rc = lib().rt_create(C.byref(cfg), C.byref(self._h))
        if rc != RT_OK:
            raise RtError(rc, (lib().rt_last_error(None) or b"").decode())
        self.rank, self.world_size = rank, world_size
        self.width = self.height = 0
        self.n_nodes = self.n_tris = 0
        self.env_loaded = True  # the dummy cube map counts (application.cpp:281)

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, (lib().rt_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_bvh(self, nodes12, tris12):
        n, t = _f32(nodes12).reshape(-1, 12), _f32(tris12).reshape(-1, 12)
        self._check(lib().rt_upload_bvh(self._h, _fp(n), n.shape[0], _fp(t), t.shape[0]))
        self.n_nodes, self.n_tris = n.shape[0], t.shape[0]

    def build_bvh_gpu(self, tris9):
        """Median-split BVH built on this context's GPU -> (nodes12, tris12); same tree as build_bvh, leaf-internal order may differ."""
        t = _f32(tris9).reshape(-1, 9)
        n = t.shape[0]
        nodes = np.zeros((max(2 * n, 1), 12), np.float32)
        tris = np.zeros((max(n, 1), 12), np.float32)
        k = lib().rt_build_bvh_gpu(self._h, _fp(t), n, _fp(nodes), _fp(tris))
        if k < 0:
            self._check(k)
        return nodes[:k].copy(), tris[:n].copy()

    def upload_env(self, faces):
        if faces is None:
            self._check(lib().rt_upload_env(self._h, None, 0, 0))
            return
        f = np.ascontiguousarray(faces, dtype=np.uint8)
        self._check(lib().rt_upload_env(self._h, f.ctypes.data_as(_U8P), f.shape[1], f.shape[3]))

    def resize(self, w, h):
        self._check(lib().rt_resize(self._h, w, h))
        self.width, self.height = w, h

    def reset_accum(self):
        self._check(lib().rt_reset_accum(self._h))

    @property
    def frame_index(self):
        return lib().rt_frame_index(self._h)

    def render_frame(self, u: RtUniforms):
        self._check(lib().rt_render_frame(self._h, C.byref(u)))

    def render_frames(self, us):
        """Consecutive frames, batched into as few launches as possible (rt_render_frames); us: sequence of RtUniforms."""
        arr = (RtUniforms * len(us))(*us)
        self._check(lib().rt_render_frames(self._h, arr, len(us)))

    def render_ray(self, params, cam, use_bvh=False, show_motion=False, view=None, proj=None):
        v = None if view is None else _f32(view)
        p = None if proj is None else _f32(proj)
        self._check(lib().rt_render_ray(self._h, C.byref(params), C.byref(cam), int(use_bvh), int(show_motion),
                                        None if v is None else _fp(v), None if p is None else _fp(p)))

    def set_extension(self, gi_bounces=1, env_filter=0):
        """gi_bounces: EXTENSION (not in the reference), bounces of the analytic / hybrid GI path.  env_filter: cube-map filter model
        (0 = bilinear weights in exact fp32, the default; 1 = texel coordinates rounded to 1/256 texel first)."""
        e = RtExtension(giBounces=int(gi_bounces), envFilter=int(env_filter))
        self._check(lib().rt_set_extension(self._h, C.byref(e)))

    def render_ray_frames(self, params, cam, count, use_bvh=False, show_motion=False):
        self._check(lib().rt_render_ray_frames(self._h, C.byref(params), C.byref(cam), int(use_bvh), int(show_motion), int(count)))

    def synchronize(self):
        self._check(lib().rt_synchronize(self._h))

    def read_target(self, which, fmt=RT_FORMAT_F16) -> np.ndarray:
        ch = TARGET_CHANNELS[which]
        out = np.zeros((self.height, self.width, ch), np.uint16 if fmt == RT_FORMAT_F16 else np.float32)
        self._check(lib().rt_read_target(self._h, which, out.ctypes.data_as(C.c_void_p), fmt))
        return out

    def write_target(self, which, image) -> None:
        """Overwrite a target of the last frame from an [H, W, channels] uint16 (half bits) image; COLOR = the history."""
        a = np.ascontiguousarray(image, np.uint16)
        if a.shape != (self.height, self.width, TARGET_CHANNELS[which]):
            raise ValueError(f"write_target: expected {(self.height, self.width, TARGET_CHANNELS[which])}, got {a.shape}")
        self._check(lib().rt_write_target(self._h, which, a.ctypes.data_as(C.c_void_p), RT_FORMAT_F16))

    def present(self, params, show_motion=False) -> np.ndarray:
        """Present pass over the last frame -> [H, W, 4] uint8 (row 0 = bottom)."""
        pp = make_present_params(params, show_motion, self.width, self.height)
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._check(lib().rt_present(self._h, C.byref(pp), out.ctypes.data_as(_U8P)))
        return out

    def present_with(self, pp: RtPresentParams) -> np.ndarray:
        """Present pass with an explicit rt_present.frag uniform block."""
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._check(lib().rt_present(self._h, C.byref(pp), out.ctypes.data_as(_U8P)))
        return out

    def history_exchange_buffer(self):
        """(device pointer, bytes) of the buffer the ranks' COLOR0 blocks of the last frame are all-gathered into (moving camera)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(lib().rt_history_exchange_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def history_exchanged(self):
        self._check(lib().rt_history_exchanged(self._h))

    def present_gathered(self, pp: RtPresentParams, color_ptr, motion_ptr, gpos_ptr, gnrm_ptr) -> np.ndarray:
        """Present pass on the gathering rank over four device arrays of gathered blocks -> [H, W, 4] uint8."""
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._check(lib().rt_present_gathered(self._h, C.byref(pp), C.c_void_p(color_ptr), C.c_void_p(motion_ptr), C.c_void_p(gpos_ptr),
                                              C.c_void_p(gnrm_ptr), out.ctypes.data_as(_U8P)))
        return out

    def read_all(self):
        return [self.read_target(i) for i in range(4)]

    # ---- tile-parallel exchange owned by the library (RCCL behind the C ABI)
    def comm_init(self, comm_id: bytes):
        """Collective over the ranks of the frame.  comm_id: the 128 bytes rank 0 got from comm_unique_id()."""
        buf = C.create_string_buffer(bytes(comm_id), RT_COMM_ID_BYTES)
        self._check(lib().rt_comm_init(self._h, buf, RT_COMM_ID_BYTES))

    def comm_destroy(self):
        self._check(lib().rt_comm_destroy(self._h))

    def gather_frame(self, which=RT_TARGET_COLOR):
        """Enqueue the gather of the last frame's target to rank 0 (+ un-tiling there) on that frame's stream."""
        self._check(lib().rt_gather_frame(self._h, which))

    def gathered_frame_ptr(self, which=RT_TARGET_COLOR):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(lib().rt_gathered_frame(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_gathered(self, which=RT_TARGET_COLOR) -> np.ndarray:
        """Rank 0: the frame of the last gather_frame(which) as half bit patterns [H, W, C] (synchronises)."""
        out = np.zeros((self.height, self.width, TARGET_CHANNELS[which]), np.uint16)
        self._check(lib().rt_read_gathered(self._h, which, out.ctypes.data_as(C.c_void_p)))
        return out

    def present_last_gathered(self, pp: RtPresentParams) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._check(lib().rt_present_last_gathered(self._h, C.byref(pp), out.ctypes.data_as(_U8P)))
        return out

    def exchange_history(self):
        """All-gather of the last frame's COLOR0 blocks so that the next frame may reproject across tiles (moving camera)."""
        self._check(lib().rt_exchange_history(self._h))

    def comm_info(self) -> RtCommInfo:
        """What the RCCL communicator reports about itself (commWorld / commRank, -1 without one) + gather counts and bytes of this context."""
        i = RtCommInfo()
        self._check(lib().rt_comm_info(self._h, C.byref(i)))
        return i

    def local_target(self, which):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(lib().rt_local_target(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def gather_block_bytes(self, which):
        n = C.c_size_t()
        self._check(lib().rt_gather_block_bytes(self._h, which, C.byref(n)))
        return n.value

    def assemble_gathered(self, which, gathered_ptr, dst_ptr):
        self._check(lib().rt_assemble_gathered(self._h, which, C.c_void_p(gathered_ptr), C.c_void_p(dst_ptr)))

    def stream(self):
        s = C.c_void_p()
        self._check(lib().rt_stream(self._h, C.byref(s)))
        return s.value

    def counters(self) -> RtCounters:
        c = RtCounters()
        self._check(lib().rt_get_counters(self._h, C.byref(c)))
        return c

    def reset_counters(self):
        self._check(lib().rt_reset_counters(self._h))

    def scene_info(self) -> RtSceneInfo:
        i = RtSceneInfo()
        self._check(lib().rt_get_scene_info(self._h, C.byref(i)))
        return i

    def memory_info(self) -> RtMemoryInfo:
        i = RtMemoryInfo()
        self._check(lib().rt_get_memory_info(self._h, C.byref(i)))
        return i

    def traced_rays(self, reset=False) -> RtTracedRays:
        t = RtTracedRays()
        self._check(lib().rt_get_traced_rays(self._h, C.byref(t), int(reset)))
        return t

    def enable_stage_timing(self, on=True):
        self._check(lib().rt_enable_stage_timing(self._h, int(on)))

    def stage_times(self):
        t = RtStageTimes()
        self._check(lib().rt_get_stage_times(self._h, C.byref(t)))
        return {"frames": t.frames,
                "stages": {lib().rt_stage_name(i).decode(): {"ms": t.ms[i], "launches": t.launches[i]}
                           for i in range(t.nStages) if t.launches[i]}}

    def debug_eval(self, op, a, b=None, c=None) -> np.ndarray:
        a = _f32(a).reshape(-1)
        b = None if b is None else _f32(b).reshape(-1)
        c = None if c is None else _f32(c).reshape(-1)
        out = np.zeros(a.size, np.uint32)
        self._check(lib().rt_debug_eval(self._h, op, _fp(a), None if b is None else _fp(b), None if c is None else _fp(c),
                                        out.ctypes.data_as(_U32P), a.size))
        return out

    def debug_trace(self, kind, origins, dirs, tmax=None, eps=1e-4, inf=1e30) -> np.ndarray:
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        t = np.full(o.shape[0], inf, np.float32) if tmax is None else _f32(tmax).reshape(-1)
        out = np.zeros((o.shape[0], 7), np.float32)
        self._check(lib().rt_debug_trace(self._h, kind, _fp(o), _fp(d), _fp(t), eps, inf, _fp(out), o.shape[0]))
        return out


def frame_uniforms(params, cam, w, h, frame_index, use_bvh, node_count=0, tri_count=0, prev_vp=None, env_loaded=True,
                   show_motion=False) -> RtUniforms:
    """Static-camera convenience: the uniform block mainLoop would hand renderRay for this frame."""
    view, proj = camera_view(cam), camera_proj(cam)
    vp = mat4_mul(proj, view)
    prev = vp if prev_vp is None else prev_vp
    moved = camera_moved(vp, prev)
    return make_uniforms(params, cam, view, vp, prev, w, h, frame_index, moved, use_bvh, show_motion, node_count, tri_count, env_loaded)

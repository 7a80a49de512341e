"""ctypes binding of the parity oracle (oracle/liborc.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

import opengl_raytracing_amd as rt

ROOT = Path(__file__).resolve().parent.parent
ORC_DIR = ROOT / "oracle"
ORC_LIB = ORC_DIR / "liborc.so"

_FP = C.POINTER(C.c_float)
_U8P = C.POINTER(C.c_uint8)
_U16P = C.POINTER(C.c_uint16)
_U32P = C.POINTER(C.c_uint32)
_lib = None


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("raysClosest", "raysShadow", "raysAnalytic", "nodeFetch", "triFetch", "envLookup", "hitPixels",
                                             "fetchPrimary", "fetchShadow", "fetchAO")]

    @property
    def rays(self):
        return self.raysClosest + self.raysShadow + self.raysAnalytic

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


def build():
    subprocess.run(["make", "-C", str(ORC_DIR)], check=True, capture_output=True)


def build_native(out_dir=None):
    """The same two sources with `-O3 -march=native` (SURVEY.md 8d's CPU-baseline flags), built ON THE MACHINE THAT RUNS IT
    (an -march=native object from another host may not run here) -> path of liborc_native.so."""
    out = Path(out_dir) if out_dir else ORC_DIR
    target = out / "liborc_native.so"
    subprocess.run(["make", "-C", str(ORC_DIR), "native", f"NATIVE_OUT={target}"], check=True, capture_output=True)
    return target


def load(path):
    """A second, independently loaded oracle library with the signatures of render() set (the CPU-baseline build)."""
    L = C.CDLL(str(path))
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [C.POINTER(rt.RtUniforms), _FP, _FP, _U8P, C.c_int, C.c_int, _U16P, _U16P, _U16P, _U16P, _U16P,
                             C.c_int, C.c_int, C.c_int, C.c_int, _U8P, C.c_int, C.POINTER(OrcCounters)]
    L.orc_set_gi_bounces.argtypes = [C.c_int]
    L.orc_set_env_filter.argtypes = [C.c_int]
    return L


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not ORC_LIB.exists():
        build()
    L = C.CDLL(str(ORC_LIB))
    L.orc_hash2.restype = C.c_uint32; L.orc_hash2.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_rand_bits.restype = C.c_uint32; L.orc_rand_bits.argtypes = [C.c_float, C.c_float, C.c_int]
    L.orc_rand.restype = C.c_float; L.orc_rand.argtypes = [C.c_float, C.c_float, C.c_int]
    L.orc_ld2.argtypes = [C.c_int, _FP]
    for n in ("sin", "cos", "exp2", "log2"):
        f = getattr(L, "orc_" + n); f.restype = C.c_float; f.argtypes = [C.c_float]
    L.orc_pow.restype = C.c_float; L.orc_pow.argtypes = [C.c_float, C.c_float]
    L.orc_f32_to_f16.restype = C.c_uint16; L.orc_f32_to_f16.argtypes = [C.c_float]
    L.orc_f16_to_f32.restype = C.c_float; L.orc_f16_to_f32.argtypes = [C.c_uint16]
    L.orc_concentric.argtypes = [C.c_float, C.c_float, C.c_float, _FP]
    L.orc_sample_hemisphere.argtypes = [C.c_float, _FP, C.c_float, C.c_float, _FP]
    L.orc_texture_cube.argtypes = [_U8P, C.c_int, C.c_int, _FP, _FP]
    L.orc_trace_bvh.restype = C.c_int
    L.orc_trace_bvh.argtypes = [C.POINTER(rt.RtUniforms), _FP, _FP, _FP, _FP, _FP, _FP, _FP, C.POINTER(OrcCounters)]
    L.orc_trace_bvh_shadow.restype = C.c_int
    L.orc_trace_bvh_shadow.argtypes = [C.POINTER(rt.RtUniforms), _FP, _FP, _FP, _FP, C.c_float]
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [C.POINTER(rt.RtUniforms), _FP, _FP, _U8P, C.c_int, C.c_int, _U16P, _U16P, _U16P, _U16P, _U16P,
                             C.c_int, C.c_int, C.c_int, C.c_int, _U8P, C.c_int, C.POINTER(OrcCounters)]
    L.orc_set_gi_bounces.argtypes = [C.c_int]
    L.orc_set_env_filter.argtypes = [C.c_int]
    L.orc_default_render_params.argtypes = [C.POINTER(rt.RtRenderParams)]
    L.orc_default_camera.argtypes = [C.POINTER(rt.RtCamera)]
    L.orc_default_bvh_transform.argtypes = [_FP]
    L.orc_camera_view.argtypes = [C.POINTER(rt.RtCamera), _FP]
    L.orc_camera_proj.argtypes = [C.POINTER(rt.RtCamera), _FP]
    L.orc_mat4_mul.argtypes = [_FP, _FP, _FP]
    L.orc_generate_jitter.argtypes = [C.c_int, _FP]
    L.orc_camera_moved.restype = C.c_int; L.orc_camera_moved.argtypes = [_FP, _FP]
    L.orc_make_uniforms.argtypes = [C.POINTER(rt.RtRenderParams), C.POINTER(rt.RtCamera), _FP, _FP, _FP] + [C.c_int] * 9 + [C.POINTER(rt.RtUniforms)]
    L.orc_gather_triangles.restype = C.c_int; L.orc_gather_triangles.argtypes = [_FP, _U32P, C.c_int, _FP, _FP]
    L.orc_build_bvh.restype = C.c_int; L.orc_build_bvh.argtypes = [_FP, C.c_int, _FP, _FP]
    L.orc_cubemap_from_cross.restype = C.c_int; L.orc_cubemap_from_cross.argtypes = [_U8P, C.c_int, C.c_int, C.c_int, _U8P]
    L.orc_present.restype = C.c_int
    L.orc_present.argtypes = [C.POINTER(rt.RtPresentParams), _U16P, _U16P, _U16P, _U16P, _U8P, C.c_int]
    L.orc_exp.restype = C.c_float; L.orc_exp.argtypes = [C.c_float]
    L.orc_atan2.restype = C.c_float; L.orc_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_sizeof_uniforms.restype = C.c_int
    L.orc_sizeof_render_params.restype = C.c_int
    assert L.orc_sizeof_uniforms() == C.sizeof(rt.RtUniforms)
    assert L.orc_sizeof_render_params() == C.sizeof(rt.RtRenderParams)
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(_FP)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def default_render_params():
    p = rt.RtRenderParams(); lib().orc_default_render_params(C.byref(p)); return p


def default_camera():
    c = rt.RtCamera(); lib().orc_default_camera(C.byref(c)); return c


def default_bvh_transform():
    m = np.zeros(16, np.float32); lib().orc_default_bvh_transform(_fp(m)); return m


def camera_view(cam):
    m = np.zeros(16, np.float32); lib().orc_camera_view(C.byref(cam), _fp(m)); return m


def camera_proj(cam):
    m = np.zeros(16, np.float32); lib().orc_camera_proj(C.byref(cam), _fp(m)); return m


def mat4_mul(a, b):
    a, b = _f32(a), _f32(b)
    m = np.zeros(16, np.float32); lib().orc_mat4_mul(_fp(a), _fp(b), _fp(m)); return m


def generate_jitter(i):
    j = np.zeros(2, np.float32); lib().orc_generate_jitter(i, _fp(j)); return j


def camera_moved(a, b):
    a, b = _f32(a), _f32(b)
    return bool(lib().orc_camera_moved(_fp(a), _fp(b)))


def make_uniforms(params, cam, view, curr_vp, prev_vp, w, h, frame_index=0, camera_moved=False, use_bvh=False, show_motion=False,
                  node_count=0, tri_count=0, env_loaded=True):
    u = rt.RtUniforms()
    v, c, p = _f32(view), _f32(curr_vp), _f32(prev_vp)
    lib().orc_make_uniforms(C.byref(params), C.byref(cam), _fp(v), _fp(c), _fp(p), int(w), int(h), int(frame_index), int(camera_moved),
                            int(use_bvh), int(show_motion), int(node_count), int(tri_count), int(env_loaded), C.byref(u))
    return u


def frame_uniforms(params, cam, w, h, frame_index, use_bvh, node_count=0, tri_count=0, prev_vp=None, env_loaded=True, show_motion=False):
    view, proj = camera_view(cam), camera_proj(cam)
    vp = mat4_mul(proj, view)
    prev = vp if prev_vp is None else prev_vp
    return make_uniforms(params, cam, view, vp, prev, w, h, frame_index, camera_moved(vp, prev), use_bvh, show_motion, node_count,
                         tri_count, env_loaded)


def gather_triangles(positions, indices, model=None):
    pos = _f32(positions).reshape(-1)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    m = default_bvh_transform() if model is None else _f32(model)
    out = np.zeros((idx.size // 3, 9), np.float32)
    n = lib().orc_gather_triangles(_fp(pos), idx.ctypes.data_as(_U32P), idx.size, _fp(m), _fp(out))
    return out[:n]


def build_bvh(tris9):
    t = _f32(tris9).reshape(-1, 9)
    n = t.shape[0]
    nodes = np.zeros((max(2 * n, 1), 12), np.float32)
    tris = np.zeros((max(n, 1), 12), np.float32)
    k = lib().orc_build_bvh(_fp(t), n, _fp(nodes), _fp(tris))
    return nodes[:k].copy(), tris[:n].copy()


def cubemap_from_cross(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, ch = img.shape
    n = h // 3
    faces = np.zeros((6, n, n, ch), np.uint8)
    got = lib().orc_cubemap_from_cross(img.ctypes.data_as(_U8P), w, h, ch, faces.ctypes.data_as(_U8P))
    if got == 0:
        raise ValueError("bad cross")
    return faces


def render(u, nodes12=None, tris12=None, env_faces=None, prev=None, region=None, mask=None, nthreads=8, L=None, gi_bounces=1, env_filter=0):
    """One frame by the oracle -> ([color, motion, gpos, gnrm] uint16 arrays HxWxC, OrcCounters).  L: a library from load().
    gi_bounces: EXTENSION knob of the analytic / hybrid GI path (1 = the reference).  env_filter: cube-map filter model (0 = exact
    fp32 weights, 1 = texel coordinates rounded to 1/256 texel)."""
    (L or lib()).orc_set_gi_bounces(int(gi_bounces))
    (L or lib()).orc_set_env_filter(int(env_filter))
    W, H = int(u.resolution[0]), int(u.resolution[1])
    outs = [np.zeros((H, W, c), np.uint16) for c in (4, 2, 4, 4)]
    n = None if nodes12 is None else _f32(nodes12)
    t = None if tris12 is None else _f32(tris12)
    e = None if env_faces is None else np.ascontiguousarray(env_faces, np.uint8)
    p = None if prev is None else np.ascontiguousarray(prev, np.uint16)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    x0, y0, x1, y1 = region if region else (0, 0, W, H)
    cnt = OrcCounters()
    rc = (L or lib()).orc_render(C.byref(u), None if n is None else _fp(n), None if t is None else _fp(t),
                          None if e is None else e.ctypes.data_as(_U8P), 0 if e is None else e.shape[1], 0 if e is None else e.shape[3],
                          None if p is None else p.ctypes.data_as(_U16P), *[o.ctypes.data_as(_U16P) for o in outs],
                          x0, y0, x1, y1, None if m is None else m.ctypes.data_as(_U8P), nthreads, C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return outs, cnt


def present(pp, targets, nthreads=8):
    """rt_present.frag by the oracle over [color, motion, gpos, gnrm] half images -> [H, W, 4] uint8."""
    t = [np.ascontiguousarray(a, np.uint16) for a in targets]
    H, W = t[0].shape[:2]
    out = np.zeros((H, W, 4), np.uint8)
    rc = lib().orc_present(C.byref(pp), *[a.ctypes.data_as(_U16P) for a in t], out.ctypes.data_as(_U8P), nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_present failed: {rc}")
    return out


def aabb_hit(ro, rd, bmin, bmax):
    out = np.zeros(4, np.float32)
    f = lib().orc_aabb_hit
    f.restype = None
    f.argtypes = [_FP] * 5
    f(_fp(_f32(ro)), _fp(_f32(rd)), _fp(_f32(bmin)), _fp(_f32(bmax)), _fp(out))
    return out


def tri_hit(u, ro, rd, tri12, tmax):
    out = np.zeros(5, np.float32)
    f = lib().orc_tri_hit
    f.restype = None
    f.argtypes = [C.POINTER(rt.RtUniforms), _FP, _FP, _FP, C.c_float, _FP]
    f(C.byref(u), _fp(_f32(ro)), _fp(_f32(rd)), _fp(_f32(tri12)), float(tmax), _fp(out))
    return out


def shade_bvh_hits(u, env_faces, hits12):
    """rt.frag's BVH shading branch for given hits with an empty BVH (see orc_shade_bvh_hits) -> n x 3 float32."""
    h = _f32(hits12).reshape(-1, 12)
    out = np.zeros((h.shape[0], 3), np.float32)
    e = None if env_faces is None else np.ascontiguousarray(env_faces, np.uint8)
    f = lib().orc_shade_bvh_hits
    f.restype = None
    f.argtypes = [C.POINTER(rt.RtUniforms), _U8P, C.c_int, C.c_int, _FP, C.c_int, _FP]
    f(C.byref(u), None if e is None else e.ctypes.data_as(_U8P), 0 if e is None else e.shape[1], 0 if e is None else e.shape[3], _fp(h), h.shape[0], _fp(out))
    return out


def trace_bvh(u, nodes12, tris12, ro, rd):
    n, t = _f32(nodes12), _f32(tris12)
    ro, rd = _f32(ro), _f32(rd)
    tt = C.c_float(); p = np.zeros(3, np.float32); nn = np.zeros(3, np.float32); cnt = OrcCounters()
    hit = lib().orc_trace_bvh(C.byref(u), _fp(n), _fp(t), _fp(ro), _fp(rd), C.byref(tt), _fp(p), _fp(nn), C.byref(cnt))
    return bool(hit), tt.value, p, nn, cnt


def trace_bvh_shadow(u, nodes12, tris12, ro, rd, tmax):
    n, t = _f32(nodes12), _f32(tris12)
    ro, rd = _f32(ro), _f32(rd)
    return bool(lib().orc_trace_bvh_shadow(C.byref(u), _fp(n), _fp(t), _fp(ro), _fp(rd), float(tmax)))


def half_to_float(a):
    return np.ascontiguousarray(a, np.uint16).view(np.float16).astype(np.float32)


def compare(a, b):
    """RMSE / max-abs / #|d|>1e-2 / #bit-different between two half images (same shape)."""
    fa, fb = half_to_float(a), half_to_float(b)
    d = (fa - fb).astype(np.float64)
    d = np.nan_to_num(d, nan=1e9)
    return {"rmse": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()), "outliers": int((np.abs(d) > 1e-2).sum()),
            "bit_diff": int((np.asarray(a) != np.asarray(b)).sum())}

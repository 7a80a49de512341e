"""oracle/glsl_ref.py -- TEST INFRASTRUCTURE ONLY: run the REFERENCE's own GLSL on the CPU.

The reference's hot path is GLSL (shaders/rt/rt.frag + includes, rt_present.frag).  This image has no GPU
GL driver, X server or OSMesa, but it does hold a software OpenGL ES 3.0 implementation (SwiftShader
4.1, shipped inside the `kaleido` wheel as libEGL.so / libGLESv2.so).  This module drives it through
ctypes with a pbuffer context and executes the reference shaders, read from /root/reference at run
time, never copied, to produce golden input/output vectors for tests/golden/ (tests/golden/make_glsl_golden.py).

What is changed in the shader text at load time, and nothing else:
  * `#include "x.glsl"` lines are expanded (what src/render/Shader.cpp:20-60 does in the reference);
  * line 1 `#version 410 core` becomes `#version 300 es` + default precision statements (highp = IEEE
    fp32 in SwiftShader), because the only GL on this image is GLES 3.0;
  * GLES 3.0 has no samplerBuffer: the two texture buffers (uBvhNodes / uBvhTris, rt_uniforms.glsl:73-74,
    RGBA32F texels, src/scene/bvh.cpp:181,217) are bound as 2048-texel-wide RGBA32F 2D textures and
    `texelFetch(buf, i)` is macro-mapped to `texelFetch(tex, ivec2(i & 2047, i >> 11), 0)` (index clamped to
    the texture) -- the same texel values reach the shader.
  * the vertex stage: SwiftShader 4.1 evaluates gl_VertexID as 0 for every vertex, so rt_fullscreen.vert
    (which derives the three clip-space corners (-1,-1) (3,-1) (-1,3) from gl_VertexID, :30-45) collapses to
    a degenerate triangle.  The same three corners are fed through a vertex attribute instead and
    vUV = 0.5 * (p + 1.0) is computed as at :44.  The fragment stage -- the hot path -- is untouched.
  * SwiftShader 4.1 mis-executes `continue` inside a `while` loop (minimal reproducer without any reference text:
    tests/golden/swiftshader_continue_defect.py, log beside it), which is what broke traceBVH / traceBVHShadow
    (rt_bvh.glsl:208,272) in round 1.  With rewrite_continue=True the two `if (C) continue;` statements become
    `if (!(C)) { rest of the loop body }` (structured_continue below: same condition, same evaluation order, same statements
    executed) and the traversal loops, BVH frames included, run.  Analytic frames and the present pass need no rewrite.
All fragment arithmetic is the reference's.  SwiftShader's sin/cos/pow/exp2/log2/inversesqrt/normalize are its own
approximations, so outputs agree with the oracle's float model to a tolerance, not to the bit, and the
sin-based hash (rt_common.glsl) decorrelates the *noise* of stochastic terms: tests compare deterministic
outputs (G-buffer, motion, direct light, sky, analytic materials) per pixel and stochastic ones statistically.

This module exists only in this container's workflow (it needs /root/reference and kaleido's SwiftShader);
nothing in tests/, bench.py or smoke() imports it -- they read the committed fixtures.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import re
import tempfile

import numpy as np

REF_SHADERS = "/root/reference/shaders/rt"
TBO_W_LOG2 = 11
TBO_H = 64       # every stand-in texture is 2048 x 64 texels, so the index clamp is a constant

GL_FRAGMENT_SHADER, GL_VERTEX_SHADER = 0x8B30, 0x8B31
GL_TEXTURE_2D, GL_TEXTURE_CUBE_MAP, GL_TEXTURE_CUBE_MAP_POSITIVE_X = 0x0DE1, 0x8513, 0x8515
GL_RGBA, GL_RG, GL_RGB = 0x1908, 0x8227, 0x1907
GL_RGBA16F, GL_RG16F, GL_RGBA32F, GL_RGB8, GL_RGBA8 = 0x881A, 0x822F, 0x8814, 0x8051, 0x8058
GL_FLOAT, GL_HALF_FLOAT, GL_UNSIGNED_BYTE = 0x1406, 0x140B, 0x1401
GL_NEAREST, GL_LINEAR, GL_CLAMP_TO_EDGE = 0x2600, 0x2601, 0x812F
GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_FRAMEBUFFER_COMPLETE = 0x8D40, 0x8CE0, 0x8CD5

_FULLSCREEN_VS = """#version 300 es
in vec2 aCorner;
out vec2 vUV;
void main() { vUV = 0.5 * (aCorner + 1.0); gl_Position = vec4(aCorner, 0.0, 1.0); }
"""

_PREAMBLE = """#version 300 es
precision highp float;
precision highp int;
precision highp sampler2D;
precision highp samplerCube;
#define samplerBuffer sampler2D
#define RT_TBO_CLAMP(i) clamp((i), 0, %d)
#define texelFetch(s, i) texelFetch(s, ivec2(RT_TBO_CLAMP(i) & %d, RT_TBO_CLAMP(i) >> %d), 0)
""" % ((1 << TBO_W_LOG2) * TBO_H - 1, (1 << TBO_W_LOG2) - 1, TBO_W_LOG2)

# RtUniforms field -> GLSL uniform name (rt_uniforms.glsl:25-177, same order as the struct)
UNIFORM_NAMES = {
    "eps": "uEPS", "pi": "uPI", "inf": "uINF", "camPos": "uCamPos", "camRight": "uCamRight", "camUp": "uCamUp",
    "camFwd": "uCamFwd", "tanHalfFov": "uTanHalfFov", "aspect": "uAspect", "frameIndex": "uFrameIndex", "spp": "uSpp",
    "resolution": "uResolution", "jitter": "uJitter", "enableJitter": "uEnableJitter", "useBVH": "uUseBVH",
    "nodeCount": "uNodeCount", "triCount": "uTriCount", "showMotion": "uShowMotion", "prevViewProj": "uPrevViewProj",
    "currViewProj": "uCurrViewProj", "cameraMoved": "uCameraMoved", "taaStillThresh": "uTaaStillThresh",
    "taaHardMovingThresh": "uTaaHardMovingThresh", "taaHistoryMinWeight": "uTaaHistoryMinWeight",
    "taaHistoryAvgWeight": "uTaaHistoryAvgWeight", "taaHistoryMaxWeight": "uTaaHistoryMaxWeight",
    "taaHistoryBoxSize": "uTaaHistoryBoxSize", "enableTAA": "uEnableTAA", "giScaleAnalytic": "uGiScaleAnalytic",
    "giScaleBVH": "uGiScaleBVH", "enableGI": "uEnableGI", "enableAO": "uEnableAO", "aoSamples": "uAO_SAMPLES",
    "aoRadius": "uAO_RADIUS", "aoBias": "uAO_BIAS", "aoMin": "uAO_MIN", "useEnvMap": "uUseEnvMap",
    "envIntensity": "uEnvIntensity", "sunEnabled": "uSunEnabled", "sunColor": "uSunColor", "sunIntensity": "uSunIntensity",
    "sunDir": "uSunDir", "skyEnabled": "uSkyEnabled", "skyColor": "uSkyColor", "skyIntensity": "uSkyIntensity",
    "skyUpDir": "uSkyUpDir", "pointLightEnabled": "uPointLightEnabled", "pointLightPos": "uPointLightPos",
    "pointLightColor": "uPointLightColor", "pointLightIntensity": "uPointLightIntensity",
    "matAlbedoColor": "uMatAlbedo_AlbedoColor", "matAlbedoSpecStrength": "uMatAlbedo_SpecStrength",
    "matAlbedoGloss": "uMatAlbedo_Gloss", "matGlassAlbedo": "uMatGlass_Albedo", "matGlassIOR": "uMatGlass_IOR",
    "matGlassDistortion": "uMatGlass_Distortion", "matGlassEnabled": "uMatGlass_Enabled",
    "matMirrorAlbedo": "uMatMirror_Albedo", "matMirrorGloss": "uMatMirror_Gloss", "matMirrorEnabled": "uMatMirror_Enabled",
}


def _find_swiftshader():
    for pat in ("/usr/local/lib/python3*/dist-packages/kaleido/executable/bin/swiftshader",
                "/usr/lib/python3*/site-packages/kaleido/executable/bin/swiftshader"):
        for d in glob.glob(pat):
            if os.path.exists(os.path.join(d, "libEGL.so")):
                return d
    raise RuntimeError("no SwiftShader libEGL.so / libGLESv2.so on this image")


def expand_includes(path):
    out = []
    for line in open(path).read().split("\n"):
        m = re.match(r'\s*#include\s+"([^"]+)"', line)
        out.append(expand_includes(os.path.join(os.path.dirname(path), m.group(1))) if m else line)
    return "\n".join(out)


def structured_continue(src):
    """`if (C) continue;` + rest of the loop body  ->  `if (!(C)) {` rest of the loop body `}`.

    SwiftShader 4.1 mis-executes a `continue` in the `while (sp > 0)` traversal loops of rt_bvh.glsl:208,272: after the first
    lane of a fragment quad takes it, loads in later iterations return garbage (minimal reproducer, independent of the
    reference: tests/golden/swiftshader_continue_defect.py + .log).  The two statements are rewritten into the structured form
    above -- C is evaluated once, in the same order, with the same short-circuiting, and exactly the same statements run when it
    is false; no arithmetic changes.  Returns (text, number of statements rewritten)."""
    lines = src.split("\n")
    n = 0
    i = 0
    while i < len(lines):
        m = re.match(r"^(\s*)if \((.*)\) continue;\s*$", lines[i])
        if not m:
            i += 1
            continue
        depth, j, closed = 0, i + 1, False
        while j < len(lines) and not closed:
            code = lines[j].split("//")[0]
            for ch in code:
                if ch == "{":
                    depth += 1
                elif ch == "}":
                    if depth == 0:
                        closed = True
                        break
                    depth -= 1
            if not closed:
                j += 1
        if not closed:
            raise RuntimeError("structured_continue: no enclosing loop end for line %d" % (i + 1))
        lines[i] = "%sif (!(%s)) {" % (m.group(1), m.group(2))
        lines.insert(j, m.group(1) + "}")
        n += 1
        i += 1
    return "\n".join(lines), n


def adapt(src, rewrite_continue=False):
    lines = src.split("\n")
    if not lines[0].startswith("#version 410"):
        raise RuntimeError("unexpected first line: " + lines[0])
    body = "\n".join(lines[1:])
    if rewrite_continue:
        body, n = structured_continue(body)
        if n != 2:
            raise RuntimeError("expected the two `continue` statements of rt_bvh.glsl:208,272, rewrote %d" % n)
    return _PREAMBLE + body


class GlslReference:
    """One GLES 3.0 pbuffer context with the reference's rt pass (and present pass) linked."""

    def __init__(self, shader_dir=REF_SHADERS):
        d = _find_swiftshader()
        # SwiftShader 4.1 crashes intermittently in its worker threads on the BVH traversal loops; it reads
        # ./SwiftShader.ini when it starts, so start it from a scratch directory that asks for one thread.
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "SwiftShader.ini"), "w") as f:
                f.write("[Processor]\nThreadCount=1\n")
            os.chdir(tmp)
            try:
                self._start(d, shader_dir)
            finally:
                os.chdir(cwd)

    def _start(self, d, shader_dir):
        self.gl = gl = C.CDLL(os.path.join(d, "libGLESv2.so"), mode=C.RTLD_GLOBAL)
        self.egl = egl = C.CDLL(os.path.join(d, "libEGL.so"), mode=C.RTLD_GLOBAL)
        egl.eglGetDisplay.restype = C.c_void_p
        egl.eglGetDisplay.argtypes = [C.c_void_p]
        dpy = egl.eglGetDisplay(None)
        egl.eglInitialize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if not egl.eglInitialize(dpy, None, None):
            raise RuntimeError("eglInitialize failed")
        attrs = (C.c_int * 5)(0x3033, 1, 0x3040, 0x40, 0x3038)   # SURFACE_TYPE=PBUFFER, RENDERABLE_TYPE=ES3
        cfg, n = C.c_void_p(), C.c_int()
        egl.eglChooseConfig.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        if not egl.eglChooseConfig(dpy, attrs, C.byref(cfg), 1, C.byref(n)) or n.value < 1:
            raise RuntimeError("eglChooseConfig failed")
        pb = (C.c_int * 5)(0x3057, 16, 0x3056, 16, 0x3038)
        egl.eglCreatePbufferSurface.restype = C.c_void_p
        egl.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        surf = egl.eglCreatePbufferSurface(dpy, cfg, pb)
        egl.eglBindAPI(0x30A0)
        ca = (C.c_int * 3)(0x3098, 3, 0x3038)
        egl.eglCreateContext.restype = C.c_void_p
        egl.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        ctx = egl.eglCreateContext(dpy, cfg, None, ca)
        egl.eglMakeCurrent.argtypes = [C.c_void_p] * 4
        if not egl.eglMakeCurrent(dpy, surf, surf, ctx):
            raise RuntimeError("eglMakeCurrent failed")
        gl.glGetString.restype = C.c_char_p
        self.version = gl.glGetString(0x1F02).decode()
        gl.glCreateShader.restype = C.c_uint
        gl.glCreateProgram.restype = C.c_uint
        gl.glGetUniformLocation.restype = C.c_int
        gl.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        gl.glUniform1f.argtypes = [C.c_int, C.c_float]
        gl.glUniform2f.argtypes = [C.c_int, C.c_float, C.c_float]
        gl.glUniform3f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        gl.glUniform1i.argtypes = [C.c_int, C.c_int]
        gl.glUniformMatrix4fv.argtypes = [C.c_int, C.c_int, C.c_ubyte, C.c_void_p]
        gl.glTexImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        gl.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        self.dir = shader_dir
        self._compile(GL_VERTEX_SHADER, "rt_fullscreen.vert")   # must still compile; not linked (see module docstring)
        vs = self._compile_src(GL_VERTEX_SHADER, _FULLSCREEN_VS.encode(), "fullscreen corners")
        self.prog_rt = self._link(vs, self._compile(GL_FRAGMENT_SHADER, "rt.frag"))
        self.prog_present = self._link(vs, self._compile(GL_FRAGMENT_SHADER, "rt_present.frag"))
        vao, vb = C.c_uint(), C.c_uint()
        gl.glGenVertexArrays(1, C.byref(vao))
        gl.glBindVertexArray(vao)
        gl.glGenBuffers(1, C.byref(vb))
        gl.glBindBuffer(0x8892, vb)
        corners = np.array([-1, -1, 3, -1, -1, 3], np.float32)
        gl.glBufferData(0x8892, C.c_long(corners.nbytes), corners.ctypes.data_as(C.c_void_p), 0x88E4)
        gl.glEnableVertexAttribArray(0)
        gl.glVertexAttribPointer(0, 2, GL_FLOAT, 0, 0, None)

    # ---- GL helpers
    def _compile(self, kind, name):
        # rt.frag carries the two traversal loops; everything else compiles from the unmodified text
        return self._compile_src(kind, adapt(expand_includes(os.path.join(self.dir, name)), rewrite_continue=(name == "rt.frag")).encode(), name)

    def _compile_src(self, kind, src, name):
        gl = self.gl
        s = gl.glCreateShader(kind)
        p, ln = C.c_char_p(src), C.c_int(len(src))
        gl.glShaderSource(s, 1, C.byref(p), C.byref(ln))
        gl.glCompileShader(s)
        ok = C.c_int()
        gl.glGetShaderiv(s, 0x8B81, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(65536)
            gl.glGetShaderInfoLog(s, 65536, None, log)
            raise RuntimeError(f"{name}: {log.value.decode(errors='replace')}")
        return s

    def _link(self, vs, fs):
        gl = self.gl
        p = gl.glCreateProgram()
        gl.glAttachShader(p, vs)
        gl.glAttachShader(p, fs)
        gl.glBindAttribLocation(p, 0, b"aCorner")
        gl.glLinkProgram(p)
        ok = C.c_int()
        gl.glGetProgramiv(p, 0x8B82, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(65536)
            gl.glGetProgramInfoLog(p, 65536, None, log)
            raise RuntimeError("link: " + log.value.decode(errors="replace"))
        return p

    def _tex2d(self, internal, w, h, fmt, typ, data, filt=GL_NEAREST):
        gl = self.gl
        t = C.c_uint()
        gl.glGenTextures(1, C.byref(t))
        gl.glBindTexture(GL_TEXTURE_2D, t)
        gl.glPixelStorei(0x0CF5, 1)   # UNPACK_ALIGNMENT
        gl.glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, typ, None if data is None else data.ctypes.data_as(C.c_void_p))
        for pn, v in ((0x2801, filt), (0x2800, filt), (0x2802, GL_CLAMP_TO_EDGE), (0x2803, GL_CLAMP_TO_EDGE)):
            gl.glTexParameteri(GL_TEXTURE_2D, pn, v)
        self._check("tex2d")
        return t

    def _tbo(self, arr12):
        """N x 12 floats -> 3N RGBA32F texels in a 2048-wide 2D texture (stand-in binding for the reference's TBO)."""
        texels = np.ascontiguousarray(arr12, np.float32).reshape(-1, 4)
        w = 1 << TBO_W_LOG2
        h = TBO_H
        assert texels.shape[0] <= w * h
        buf = np.zeros((h * w, 4), np.float32)
        buf[:texels.shape[0]] = texels
        return self._tex2d(GL_RGBA32F, w, h, GL_RGBA, GL_FLOAT, buf)

    def _cube(self, faces):
        """6 x S x S x 3 uint8 -> RGB8 cube map, LINEAR, CLAMP_TO_EDGE (src/render/cubemap.cpp:77-102)."""
        gl = self.gl
        t = C.c_uint()
        gl.glGenTextures(1, C.byref(t))
        gl.glBindTexture(GL_TEXTURE_CUBE_MAP, t)
        gl.glPixelStorei(0x0CF5, 1)
        f = np.ascontiguousarray(faces, np.uint8)
        ch = f.shape[3]
        for i in range(6):
            gl.glTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + i, 0, GL_RGB8 if ch == 3 else GL_RGBA8, f.shape[2], f.shape[1], 0,
                            GL_RGB if ch == 3 else GL_RGBA, GL_UNSIGNED_BYTE, f[i].ctypes.data_as(C.c_void_p))
        for pn, v in ((0x2801, GL_LINEAR), (0x2800, GL_LINEAR), (0x2802, GL_CLAMP_TO_EDGE), (0x2803, GL_CLAMP_TO_EDGE), (0x8072, GL_CLAMP_TO_EDGE)):
            gl.glTexParameteri(GL_TEXTURE_CUBE_MAP, pn, v)
        self._check("cube")
        return t

    def _check(self, what):
        e = self.gl.glGetError()
        if e:
            raise RuntimeError(f"GL error 0x{e:x} after {what}")

    def _set_uniforms(self, prog, u):
        gl = self.gl
        for field, name in UNIFORM_NAMES.items():
            loc = gl.glGetUniformLocation(prog, name.encode())
            if loc < 0:
                continue
            v = getattr(u, field)
            if isinstance(v, int):
                gl.glUniform1i(loc, v)
            elif isinstance(v, float):
                gl.glUniform1f(loc, v)
            else:
                a = [float(x) for x in v]
                if len(a) == 2:
                    gl.glUniform2f(loc, *a)
                elif len(a) == 3:
                    gl.glUniform3f(loc, *a)
                elif len(a) == 16:
                    m = (C.c_float * 16)(*a)
                    gl.glUniformMatrix4fv(loc, 1, 0, m)   # column-major as uploaded by Shader.cpp:190-192
                else:
                    raise RuntimeError(field)

    def _bind_sampler(self, prog, name, unit, target, tex):
        gl = self.gl
        loc = gl.glGetUniformLocation(prog, name.encode())
        gl.glActiveTexture(0x84C0 + unit)
        gl.glBindTexture(target, tex)
        if loc >= 0:
            gl.glUniform1i(loc, unit)

    def _delete(self, texs):
        for t in texs:
            self.gl.glDeleteTextures(1, C.byref(t))

    # ---- the ray pass: src/render/render.cpp:120-195 (bind MRT, uniforms, textures, draw 3 vertices)
    def render(self, u, nodes12=None, tris12=None, env_faces=None, prev=None):
        """-> [color HxWx4, motion HxWx2, gpos HxWx4, gnrm HxWx4] as float16 bit patterns (uint16), row 0 = bottom."""
        gl = self.gl
        W, H = int(u.resolution[0]), int(u.resolution[1])
        one = np.zeros((1, 12), np.float32)
        tn = self._tbo(one if nodes12 is None else nodes12)
        tt = self._tbo(one if tris12 is None else tris12)
        te = self._cube(np.zeros((6, 1, 1, 3), np.uint8) if env_faces is None else env_faces)
        pv = np.zeros((H, W, 4), np.uint16) if prev is None else np.ascontiguousarray(prev, np.uint16)
        tp = self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, pv)
        outs = [self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, None), self._tex2d(GL_RG16F, W, H, GL_RG, GL_HALF_FLOAT, None),
                self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, None), self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, None)]
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        for i, t in enumerate(outs):
            gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, t, 0)
        bufs = (C.c_uint * 4)(*[GL_COLOR_ATTACHMENT0 + i for i in range(4)])
        gl.glDrawBuffers(4, bufs)
        st = gl.glCheckFramebufferStatus(GL_FRAMEBUFFER)
        if st != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError(f"FBO incomplete 0x{st:x}")
        gl.glViewport(0, 0, W, H)
        gl.glDisable(0x0C11)   # scissor
        gl.glDisable(0x0B71)   # depth
        gl.glDisable(0x0BE2)   # blend
        gl.glUseProgram(self.prog_rt)
        self._set_uniforms(self.prog_rt, u)
        self._bind_sampler(self.prog_rt, "uPrevAccum", 0, GL_TEXTURE_2D, tp)
        self._bind_sampler(self.prog_rt, "uBvhNodes", 1, GL_TEXTURE_2D, tn)
        self._bind_sampler(self.prog_rt, "uBvhTris", 2, GL_TEXTURE_2D, tt)
        self._bind_sampler(self.prog_rt, "uEnvMap", 3, GL_TEXTURE_CUBE_MAP, te)
        gl.glDrawArrays(0x0004, 0, 3)
        gl.glFinish()
        self._check("draw")
        res = []
        for i, c in enumerate((4, 2, 4, 4)):
            gl.glReadBuffer(GL_COLOR_ATTACHMENT0 + i)
            buf = np.zeros((H, W, 4), np.float32)
            gl.glPixelStorei(0x0D05, 1)   # PACK_ALIGNMENT
            gl.glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, buf.ctypes.data_as(C.c_void_p))
            self._check("read")
            res.append(np.ascontiguousarray(buf[:, :, :c]).astype(np.float16).view(np.uint16))   # exact: values are halves
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        self._delete([tn, tt, te, tp] + outs)
        return res

    # ---- the present pass: src/render/render.cpp:200-239 (default framebuffer = RGBA8 here)
    PRESENT_NAMES = {"exposure": "uExposure", "showMotion": "uShowMotion", "motionScale": "uMotionScale", "resolution": "uResolution",
                     "varMax": "uVarMax", "kVar": "uKVar", "kColor": "uKColor", "kVarMotion": "uKVarMotion",
                     "kColorMotion": "uKColorMotion", "svgfStrength": "uSvgfStrength", "enableSVGF": "uEnableSVGF"}

    def present(self, pp, targets):
        """rt_present.frag over [color, motion, gpos, gnrm] half images -> HxWx4 uint8."""
        gl = self.gl
        color, motion, gpos, gnrm = [np.ascontiguousarray(a, np.uint16) for a in targets]
        H, W = color.shape[:2]
        tin = [self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, color), self._tex2d(GL_RG16F, W, H, GL_RG, GL_HALF_FLOAT, motion),
               self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, gpos), self._tex2d(GL_RGBA16F, W, H, GL_RGBA, GL_HALF_FLOAT, gnrm)]
        out = self._tex2d(GL_RGBA8, W, H, GL_RGBA, GL_UNSIGNED_BYTE, None)
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, out, 0)
        bufs = (C.c_uint * 1)(GL_COLOR_ATTACHMENT0)
        gl.glDrawBuffers(1, bufs)
        st = gl.glCheckFramebufferStatus(GL_FRAMEBUFFER)
        if st != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError(f"FBO incomplete 0x{st:x}")
        gl.glViewport(0, 0, W, H)
        gl.glUseProgram(self.prog_present)
        for field, name in self.PRESENT_NAMES.items():
            loc = gl.glGetUniformLocation(self.prog_present, name.encode())
            if loc < 0:
                continue
            v = getattr(pp, field)
            if isinstance(v, int):
                gl.glUniform1i(loc, v)
            elif isinstance(v, float):
                gl.glUniform1f(loc, v)
            else:
                gl.glUniform2f(loc, float(v[0]), float(v[1]))
        for i, n in enumerate(("uTex", "uMotionTex", "uGPos", "uGNrm")):
            self._bind_sampler(self.prog_present, n, i, GL_TEXTURE_2D, tin[i])
        gl.glDrawArrays(0x0004, 0, 3)
        gl.glFinish()
        self._check("present draw")
        gl.glReadBuffer(GL_COLOR_ATTACHMENT0)
        buf = np.zeros((H, W, 4), np.uint8)
        gl.glPixelStorei(0x0D05, 1)
        gl.glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_BYTE, buf.ctypes.data_as(C.c_void_p))
        self._check("present read")
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        self._delete(tin + [out])
        return buf

    # ---- per-function vectors for the BVH primitives: nodeFetch, triFetch, aabbHit and triHit are called straight from a
    # small main() appended to the reference's rt_uniforms.glsl + rt_common.glsl + rt_bvh.glsl text (unmodified: no loop runs).
    _KAT_MAIN = """
uniform samplerBuffer uKatRays;   // 2 texels per case: (ro, tMax), (rd, 0)
uniform int uKatWidth;
layout(location = 0) out vec4 o0;
layout(location = 1) out vec4 o1;
layout(location = 2) out vec4 o2;
void main() {
    int i = int(gl_FragCoord.x) + int(gl_FragCoord.y) * uKatWidth;
    vec4 a = texelFetch(uKatRays, 2 * i), b = texelFetch(uKatRays, 2 * i + 1);
    NodeSOA N = nodeFetch(i);
    float t0 = 0.0, t1 = 0.0;
    bool hb = aabbHit(a.xyz, 1.0 / b.xyz, N.bmin, N.bmax, t0, t1);
    TriSOA T = triFetch(i);
    float t = 0.0;
    vec3 n = vec3(0.0);
    bool ht = triHit(a.xyz, b.xyz, T, a.w, t, n);
    o0 = vec4(hb ? 1.0 : 0.0, t0, t1, float(N.left));
    o1 = vec4(ht ? 1.0 : 0.0, t, float(N.right), float(N.first * 8 + N.count));
    o2 = vec4(n, 0.0);
}
"""

    def bvh_kat(self, nodes12, tris12, rays8, eps):
        """Case i: nodeFetch(i), triFetch(i), aabbHit(ray i, node i), triHit(ray i, tri i, tMax_i).
        rays8: n x 8 floats (ro, tMax, rd, 0).  -> (o0, o1, o2) n x 4 float32 each."""
        gl = self.gl
        n = rays8.shape[0]
        W = 64
        H = (n + W - 1) // W
        if not hasattr(self, "prog_kat"):
            body = "\n".join(open(os.path.join(self.dir, f)).read() for f in ("rt_uniforms.glsl", "rt_common.glsl", "rt_bvh.glsl"))
            vs = self._compile_src(GL_VERTEX_SHADER, _FULLSCREEN_VS.encode(), "fullscreen corners")
            fs = self._compile_src(GL_FRAGMENT_SHADER, (_PREAMBLE + body + self._KAT_MAIN).encode(), "bvh kat")
            self.prog_kat = self._link(vs, fs)
        pad = np.zeros((W * H, 12), np.float32)
        tn, tt = pad.copy(), pad.copy()
        tn[:n], tt[:n] = nodes12[:n], tris12[:n]
        rr = np.zeros((W * H, 8), np.float32)
        rr[:n] = rays8
        rr[n:, 4:7] = 1.0
        texs = [self._tbo(tn), self._tbo(tt), self._tbo(rr)]
        outs = [self._tex2d(GL_RGBA32F, W, H, GL_RGBA, GL_FLOAT, None) for _ in range(3)]
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        for i, t in enumerate(outs):
            gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, t, 0)
        bufs = (C.c_uint * 3)(*[GL_COLOR_ATTACHMENT0 + i for i in range(3)])
        gl.glDrawBuffers(3, bufs)
        if gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError("KAT FBO incomplete")
        gl.glViewport(0, 0, W, H)
        p = self.prog_kat
        gl.glUseProgram(p)
        self._bind_sampler(p, "uBvhNodes", 1, GL_TEXTURE_2D, texs[0])
        self._bind_sampler(p, "uBvhTris", 2, GL_TEXTURE_2D, texs[1])
        self._bind_sampler(p, "uKatRays", 3, GL_TEXTURE_2D, texs[2])
        gl.glUniform1i(gl.glGetUniformLocation(p, b"uKatWidth"), W)
        gl.glUniform1f(gl.glGetUniformLocation(p, b"uEPS"), float(eps))
        gl.glDrawArrays(0x0004, 0, 3)
        gl.glFinish()
        self._check("kat draw")
        res = []
        for i in range(3):
            gl.glReadBuffer(GL_COLOR_ATTACHMENT0 + i)
            buf = np.zeros((H, W, 4), np.float32)
            gl.glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, buf.ctypes.data_as(C.c_void_p))
            res.append(buf.reshape(-1, 4)[:n].copy())
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        self._delete(texs + outs)
        return res

    # ---- the traversal loops themselves: traceBVH (rt_bvh.glsl:193-243) and traceBVHShadow (:260-304), one ray per fragment,
    # called from a main() appended to the reference's rt_uniforms.glsl + rt_common.glsl + rt_bvh.glsl text with the two
    # `continue` statements in structured form (structured_continue).  Outputs are fp32: nothing is hidden by fp16 rounding.
    _TRACE_MAIN = """
uniform samplerBuffer uKatRays;   // 2 texels per ray: (ro, tMax of the shadow query), (rd, 0)
uniform int uKatWidth;
layout(location = 0) out vec4 o0;
layout(location = 1) out vec4 o1;
layout(location = 2) out vec4 o2;
void main() {
    int i = int(gl_FragCoord.x) + int(gl_FragCoord.y) * uKatWidth;
    vec4 a = texelFetch(uKatRays, 2 * i), b = texelFetch(uKatRays, 2 * i + 1);
    Hit h;
    h.t = -1.0; h.p = vec3(-2.0); h.n = vec3(-3.0); h.mat = -7;
    bool hit = traceBVH(a.xyz, b.xyz, h);
    bool occ = traceBVHShadow(a.xyz, b.xyz, a.w);
    o0 = vec4(hit ? 1.0 : 0.0, h.t, occ ? 1.0 : 0.0, float(h.mat));
    o1 = vec4(h.p, 0.0);
    o2 = vec4(h.n, 0.0);
}
"""

    def bvh_trace_kat(self, nodes12, tris12, rays8, eps, inf):
        """Ray i: traceBVH(ro, rd) and traceBVHShadow(ro, rd, tMax_i) over the given BVH.  rays8: n x 8 floats (ro, tMax, rd, 0).
        -> (o0, o1, o2) n x 4 float32: (hit, t, occluded, mat), (p, 0), (n, 0)."""
        gl = self.gl
        n = rays8.shape[0]
        W = 64
        H = (n + W - 1) // W
        if not hasattr(self, "prog_trace"):
            body = "\n".join(open(os.path.join(self.dir, f)).read() for f in ("rt_uniforms.glsl", "rt_common.glsl", "rt_bvh.glsl"))
            body, nrew = structured_continue(body)
            if nrew != 2:
                raise RuntimeError("expected the two `continue` statements of rt_bvh.glsl:208,272, rewrote %d" % nrew)
            vs = self._compile_src(GL_VERTEX_SHADER, _FULLSCREEN_VS.encode(), "fullscreen corners")
            fs = self._compile_src(GL_FRAGMENT_SHADER, (_PREAMBLE + body + self._TRACE_MAIN).encode(), "bvh trace kat")
            self.prog_trace = self._link(vs, fs)
        rr = np.zeros((W * H, 8), np.float32)
        rr[:n] = rays8
        rr[n:, 4:7] = 1.0
        texs = [self._tbo(nodes12), self._tbo(tris12), self._tbo(rr)]
        outs = [self._tex2d(GL_RGBA32F, W, H, GL_RGBA, GL_FLOAT, None) for _ in range(3)]
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        for i, t in enumerate(outs):
            gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, t, 0)
        bufs = (C.c_uint * 3)(*[GL_COLOR_ATTACHMENT0 + i for i in range(3)])
        gl.glDrawBuffers(3, bufs)
        if gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError("trace KAT FBO incomplete")
        gl.glViewport(0, 0, W, H)
        p = self.prog_trace
        gl.glUseProgram(p)
        self._bind_sampler(p, "uBvhNodes", 1, GL_TEXTURE_2D, texs[0])
        self._bind_sampler(p, "uBvhTris", 2, GL_TEXTURE_2D, texs[1])
        self._bind_sampler(p, "uKatRays", 3, GL_TEXTURE_2D, texs[2])
        for nm, v in (("uKatWidth", W), ("uNodeCount", nodes12.shape[0]), ("uTriCount", tris12.shape[0])):
            gl.glUniform1i(gl.glGetUniformLocation(p, nm.encode()), int(v))
        gl.glUniform1f(gl.glGetUniformLocation(p, b"uEPS"), float(eps))
        gl.glUniform1f(gl.glGetUniformLocation(p, b"uINF"), float(inf))
        gl.glDrawArrays(0x0004, 0, 3)
        gl.glFinish()
        self._check("trace kat draw")
        res = []
        for i in range(3):
            gl.glReadBuffer(GL_COLOR_ATTACHMENT0 + i)
            buf = np.zeros((H, W, 4), np.float32)
            gl.glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, buf.ctypes.data_as(C.c_void_p))
            res.append(buf.reshape(-1, 4)[:n].copy())
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        self._delete(texs + outs)
        return res

    # ---- the BVH shading branch of rt.frag:92-106 for given hits, with an empty BVH (uNodeCount = 0: traceBVH /
    # traceBVHShadow return before their loops, so SwiftShader's loop defect is not reached and every shadow / bounce /
    # AO ray leaves unoccluded).  main() of rt.frag is replaced by one that takes the hit from a texture.
    _SHADE_MAIN = """
uniform samplerBuffer uKatRays;   // 3 texels per case: (p, fragX), (n, fragY), (V, seed)
uniform int uKatWidth;
void main() {
    int i = int(gl_FragCoord.x) + int(gl_FragCoord.y) * uKatWidth;
    vec4 a = texelFetch(uKatRays, 3 * i), b = texelFetch(uKatRays, 3 * i + 1), c = texelFetch(uKatRays, 3 * i + 2);
    Hit h;
    h.p = a.xyz; h.n = b.xyz; h.mat = 1; h.t = 1.0;
    vec3 V = c.xyz;
    int seed = int(c.w);
    vec3 radiance = directLightBVH(h, seed, V);
    if (uEnableGI == 1) radiance += uGiScaleBVH * oneBounceGIBVH(h, uFrameIndex, seed);
    if (uEnableAO == 1) radiance *= computeAO(h, uFrameIndex);
    fragColor = vec4(radiance, 1.0);
    outMotion = vec2(0.0); outGPos = vec4(0.0); outGNrm = vec4(0.0);
}
"""

    def shade_bvh_hits(self, u, env_faces, hits12, width):
        """hits12: n x 12 floats {p, fragX, n, fragY, V, seed} with (fragX, fragY) = centre of pixel i of a `width`-wide grid.
        -> n x 3 float32 radiance."""
        gl = self.gl
        n = hits12.shape[0]
        W = width
        H = (n + W - 1) // W
        if not hasattr(self, "prog_shade"):
            src = adapt(expand_includes(os.path.join(self.dir, "rt.frag")))
            cut = src.index("void main()")
            vs = self._compile_src(GL_VERTEX_SHADER, _FULLSCREEN_VS.encode(), "fullscreen corners")
            fs = self._compile_src(GL_FRAGMENT_SHADER, (src[:cut] + self._SHADE_MAIN).encode(), "bvh shade kat")
            self.prog_shade = self._link(vs, fs)
        hh = np.zeros((W * H, 12), np.float32)
        hh[:n] = hits12
        hh[n:, 4:7] = hh[n:, 8:11] = (0.0, 1.0, 0.0)
        one = np.zeros((1, 12), np.float32)
        texs = [self._tbo(one), self._tbo(one), self._tbo(hh),
                self._cube(np.zeros((6, 1, 1, 3), np.uint8) if env_faces is None else env_faces),
                self._tex2d(GL_RGBA16F, 1, 1, GL_RGBA, GL_HALF_FLOAT, np.zeros((1, 1, 4), np.uint16))]
        out = self._tex2d(GL_RGBA32F, W, H, GL_RGBA, GL_FLOAT, None)
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, out, 0)
        bufs = (C.c_uint * 1)(GL_COLOR_ATTACHMENT0)
        gl.glDrawBuffers(1, bufs)
        if gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError("shade KAT FBO incomplete")
        gl.glViewport(0, 0, W, H)
        p = self.prog_shade
        gl.glUseProgram(p)
        self._set_uniforms(p, u)
        self._bind_sampler(p, "uPrevAccum", 0, GL_TEXTURE_2D, texs[4])
        self._bind_sampler(p, "uBvhNodes", 1, GL_TEXTURE_2D, texs[0])
        self._bind_sampler(p, "uBvhTris", 2, GL_TEXTURE_2D, texs[1])
        self._bind_sampler(p, "uEnvMap", 3, GL_TEXTURE_CUBE_MAP, texs[3])
        self._bind_sampler(p, "uKatRays", 4, GL_TEXTURE_2D, texs[2])
        gl.glUniform1i(gl.glGetUniformLocation(p, b"uKatWidth"), W)
        gl.glDrawArrays(0x0004, 0, 3)
        gl.glFinish()
        self._check("shade kat draw")
        gl.glReadBuffer(GL_COLOR_ATTACHMENT0)
        buf = np.zeros((H, W, 4), np.float32)
        gl.glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, buf.ctypes.data_as(C.c_void_p))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
        gl.glDeleteFramebuffers(1, C.byref(fbo))
        gl.glDeleteTextures(1, C.byref(texs[3]))
        self._delete(texs[:3] + [texs[4], out])
        return buf.reshape(-1, 4)[:n, :3].copy()

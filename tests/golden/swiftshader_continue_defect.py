"""Minimal reproducer of the SwiftShader 4.1 defect that kept the reference's traversal loops from running in round 1
(no reference text involved).  A fragment shader walks a small binary tree held in a uniform array with an explicit stack --
the shape of traceBVH (shaders/rt/rt_bvh.glsl:205-241) -- once with the culling step written as `if (c) continue;` and once
in the structured form `if (!(c)) { ... }`.  Both are the same program; SwiftShader returns the right sum only for the second.
oracle/glsl_ref.py therefore rewrites the two `continue` statements of rt_bvh.glsl:208,272 into the structured form at load
time (structured_continue) -- the only change to the traversal functions' text.

    python tests/golden/swiftshader_continue_defect.py > tests/golden/swiftshader_continue_defect.log
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import glsl_ref as G  # noqa: E402

FS = """#version 300 es
precision highp float;
precision highp int;
uniform vec4 uTree[15];     // x: value, y: 1 = cull this subtree, z / w: children (0 = none)
out vec4 o;
bool keep(vec4 nd, int ni, int start, out float lo, out float hi) {   // like aabbHit: a bool plus two out parameters
    lo = nd.x;
    hi = nd.x + 1.0;
    return !(nd.y > 0.5 && (ni %% 3) == start);
}
void main() {
    int stack[16];
    int sp = 0;
    stack[sp++] = 0;
    float acc = 0.0, lo, hi;
    int iters = 0;
    int start = int(gl_FragCoord.x) %% 3;      // neighbouring fragments cull different subtrees
    while (sp > 0) {
        int ni = stack[--sp];
        vec4 nd = uTree[ni];
        iters++;
        %s
        if (nd.z == 0.0) {
            int cnt = 1 + (ni %% 4);
            for (int k = 0; k < cnt; ++k) acc += nd.x;        // leaf: a dynamic inner loop, like the triangle loop
        } else {
            acc += nd.x;
            stack[sp++] = int(nd.z);
            stack[sp++] = int(nd.w);
        }
        %s
    }
    o = vec4(acc, float(iters), 0.0, 1.0);
}
"""
VARIANTS = {"continue": ("if (!keep(nd, ni, start, lo, hi) || lo > 1.0e9) continue;", ""),
            "structured": ("if (!(!keep(nd, ni, start, lo, hi) || lo > 1.0e9)) {", "}")}


def expected(tree, start):
    acc, iters, stack = 0.0, 0, [0]
    while stack:
        ni = stack.pop()
        x, y, z, w = tree[ni]
        iters += 1
        if y > 0.5 and ni % 3 == start:
            continue
        if z == 0:
            acc += x * (1 + ni % 4)
        else:
            acc += x
            stack.append(int(z))
            stack.append(int(w))
    return acc, iters


def main():
    g = G.GlslReference()
    gl = g.gl
    print("GL:", g.version)
    tree = np.zeros((15, 4), np.float32)
    for i in range(15):
        tree[i] = (float(1 << (i % 10)), 1.0 if i in (1, 5, 6, 9) else 0.0, 2 * i + 1 if i < 7 else 0, 2 * i + 2 if i < 7 else 0)
    W = 6
    want = np.array([expected(tree, x % 3) for x in range(W)], np.float32)
    vs = g._compile_src(G.GL_VERTEX_SHADER, G._FULLSCREEN_VS.encode(), "vs")
    ok_all = {}
    for name, (a, b) in VARIANTS.items():
        prog = g._link(vs, g._compile_src(G.GL_FRAGMENT_SHADER, (FS % (a, b)).encode(), name))
        out = g._tex2d(G.GL_RGBA32F, W, 1, G.GL_RGBA, G.GL_FLOAT, None)
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(G.GL_FRAMEBUFFER, fbo)
        gl.glFramebufferTexture2D(G.GL_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0, G.GL_TEXTURE_2D, out, 0)
        bufs = (C.c_uint * 1)(G.GL_COLOR_ATTACHMENT0)
        gl.glDrawBuffers(1, bufs)
        gl.glViewport(0, 0, W, 1)
        gl.glUseProgram(prog)
        gl.glUniform4fv.argtypes = [C.c_int, C.c_int, C.c_void_p]
        gl.glUniform4fv(gl.glGetUniformLocation(prog, b"uTree"), 15, tree.ctypes.data_as(C.c_void_p))
        gl.glDrawArrays(4, 0, 3)
        gl.glFinish()
        buf = np.zeros((1, W, 4), np.float32)
        gl.glReadPixels(0, 0, W, 1, G.GL_RGBA, G.GL_FLOAT, buf.ctypes.data_as(C.c_void_p))
        got = buf[0, :, :2]
        ok_all[name] = bool(np.array_equal(got, want))
        print(f"variant {name:10s}: (sum, iterations) per fragment = {got.tolist()}")
        print(f"{'':19s}  expected                       = {want.tolist()}  -> {'OK' if ok_all[name] else 'WRONG'}")
        gl.glBindFramebuffer(G.GL_FRAMEBUFFER, 0)
        gl.glDeleteFramebuffers(1, C.byref(fbo))
    print("defect reproduced" if (ok_all["structured"] and not ok_all["continue"]) else "defect NOT reproduced with this shader")


if __name__ == "__main__":
    main()

"""Shared scene builders for tests and bench (inputs only; no rendering here)."""
from __future__ import annotations

import functools
from pathlib import Path

import numpy as np

import opengl_raytracing_amd as rt

ROOT = Path(__file__).resolve().parent.parent
ASSETS = ROOT / "assets" / "cubemaps"


@functools.lru_cache(maxsize=8)
def env_faces(name="Sky_01"):
    return rt.load_cubemap_cross(ASSETS / f"{name}.png")


@functools.lru_cache(maxsize=8)
def bunny_bvh(subdiv=6):
    """(nodes12, tris12) of the procedural bunny stand-in under defaultBvhTransform, built by the PRODUCT host code."""
    v, f = rt.meshgen.bunny_standin(subdiv)
    tris9 = rt.gather_triangles(v, f)
    return rt.build_bvh(tris9)


def tiny_env(n=8, seed=3):
    """Small random RGB cube map (exercises face seams and bilinear weights)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(6, n, n, 3), dtype=np.uint8)


def camera(kind="default", aspect=None):
    c = rt.default_camera() if kind == "default" else rt.closeup_camera()
    if aspect is not None:
        c.aspect = aspect
    return c

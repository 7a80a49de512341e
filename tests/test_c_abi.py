"""The drop-in boundary without a GPU: librt_mi355.so loads, exports every symbol include/rt_mi355.h
declares, keeps its PODs in sync with the ctypes mirrors, and refuses to work without a device
(no CPU fallback).  No compute calls here."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

import opengl_raytracing_amd as rt

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "rt_mi355.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 40
    L = rt.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rt_mi355.h but not exported"
    assert sorted(rt.SIGNATURES) == names, "ctypes signature table and header drifted apart"
    out = subprocess.run(["nm", "-D", "--defined-only", str(rt.LIB_PATH)], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (rt_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_pod_layouts():
    L = rt.lib()
    assert L.rt_sizeof_uniforms() == C.sizeof(rt.RtUniforms) == 472
    assert L.rt_sizeof_render_params() == C.sizeof(rt.RtRenderParams)
    assert C.sizeof(rt.RtCounters) == 80 and C.sizeof(rt.RtDeviceConfig) == 32
    assert b"gfx950" in L.rt_version()
    assert [L.rt_stage_name(i) for i in range(11)][:3] == [b"mega", b"primary", b"trace_primary"]


def test_product_never_touches_the_oracle():
    for p in list((ROOT / "opengl-raytracing_amd").rglob("*")) + [ROOT / "opengl_raytracing_amd.py"]:
        if p.is_file() and p.suffix in (".py", ".hip", ".hpp", ".cpp", ".h", "") and p.name != "librt_mi355.so":
            try:
                txt = p.read_text()
            except UnicodeDecodeError:
                continue
            assert "liborc" not in txt and "oracle/" not in txt and "import oracle" not in txt, p


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Renderer()
    assert e.value.code == rt.RT_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value) or "no HIP device" in str(e.value)


def test_argument_validation_without_device():
    L = rt.lib()
    h = C.c_void_p()
    bad = rt.RtDeviceConfig(device=0, rank=2, worldSize=2)
    assert L.rt_create(C.byref(bad), C.byref(h)) == rt.RT_ERR_INVALID and not h
    assert L.rt_create(None, C.byref(h)) == rt.RT_ERR_INVALID
    assert L.rt_frame_index(None) == rt.RT_ERR_INVALID
    assert L.rt_render_frame(None, None) == rt.RT_ERR_INVALID
    assert L.rt_build_bvh(None, 3, None, None) == rt.RT_ERR_INVALID
    L.rt_destroy(None)

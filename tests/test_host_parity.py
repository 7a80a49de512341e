"""Product host code (C++ in librt_mi355.so, reached through the C ABI) against the oracle's host
restatement, on the CPU: scene build, camera / frame state, uniform marshalling, asset readers.
Everything here is integer / exactly-rounded fp32 work, so the bar is bit equality."""
import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes

GOLDEN = scenes.ROOT / "tests" / "golden"


def test_defaults_match(orc):
    assert bytes(rt.default_render_params()) == bytes(orc.default_render_params())
    assert bytes(rt.default_camera()) == bytes(orc.default_camera())
    assert np.array_equal(rt.default_bvh_transform(), orc.default_bvh_transform())
    p = rt.default_render_params()
    assert (p.sppPerFrame, p.aoSamples, p.enableTAA, p.enableSVGF) == (1, 4, 1, 1)
    assert abs(p.taaHistoryMinWeight - 0.85) < 1e-7 and abs(p.giScaleBVH - 0.20) < 1e-7 and abs(p.motionScale - 4.0) < 1e-7


@pytest.mark.parametrize("yaw,pitch,fov,aspect", [(-90, -10, 60, 16 / 9), (-90, 0, 60, 1.0), (37.5, 22, 45, 4 / 3), (180, -89, 90, 2.0)])
def test_camera_matrices_and_uniforms(orc, yaw, pitch, fov, aspect):
    cam = rt.default_camera()
    cam.yaw, cam.pitch, cam.fov, cam.aspect = yaw, pitch, fov, aspect
    cam.pos[0] = 1.25
    assert np.array_equal(rt.camera_view(cam), orc.camera_view(cam))
    assert np.array_equal(rt.camera_proj(cam), orc.camera_proj(cam))
    vp = rt.mat4_mul(rt.camera_proj(cam), rt.camera_view(cam))
    assert np.array_equal(vp, orc.mat4_mul(orc.camera_proj(cam), orc.camera_view(cam)))
    p = rt.default_render_params()
    p.pointLightOrbitEnabled, p.pointLightYaw, p.pointLightPitch = 1, 33.0, 12.0
    p.sunYaw, p.skyPitch = 200.0, 70.0
    for frame, moved in ((0, False), (5, True), (1030, False)):
        prev = vp.copy()
        if moved:
            prev[12] += 0.01
        a = rt.make_uniforms(p, cam, rt.camera_view(cam), vp, prev, 640, 360, frame, moved, True, False, 10, 20, True)
        b = orc.make_uniforms(p, cam, orc.camera_view(cam), vp, prev, 640, 360, frame, moved, True, False, 10, 20, True)
        assert bytes(a) == bytes(b)
        assert rt.camera_moved(vp, prev) == orc.camera_moved(vp, prev) == moved
    u = rt.make_uniforms(p, cam, rt.camera_view(cam), vp, vp, 64, 64, 0, False, False, True, 0, 0, False)
    assert u.spp == 1 and u.showMotion == 1 and u.useEnvMap == 0     # render.cpp:81,102


def test_jitter_sequence(orc):
    for i in list(range(40)) + [1023, 1024, 5000]:
        assert np.array_equal(rt.generate_jitter(i), orc.generate_jitter(i))


@pytest.mark.parametrize("kind", ["bunny2", "bunny4", "random300", "equal_centroids", "n9", "n1", "degenerate"])
def test_bvh_build_matches_oracle(orc, kind):
    rng = np.random.default_rng(9)
    if kind.startswith("bunny"):
        v, f = rt.meshgen.bunny_standin(int(kind[-1]))
        a, b = rt.gather_triangles(v, f), orc.gather_triangles(v, f)
        assert np.array_equal(a, b)
        t = a
    elif kind == "random300":
        t = rng.normal(size=(300, 9)).astype(np.float32)
    elif kind == "equal_centroids":   # nth_element ties: both sides run the same libstdc++ algorithm on the same keys
        t = np.tile(rng.normal(size=(1, 9)).astype(np.float32), (64, 1))
        t[::2, 0] += 1.0
    elif kind == "n9":
        t = rng.normal(size=(9, 9)).astype(np.float32)
    elif kind == "n1":
        t = rng.normal(size=(1, 9)).astype(np.float32)
    else:
        t = np.zeros((20, 9), np.float32)   # zero-area triangles at the origin
    n1, t1 = rt.build_bvh(t)
    n2, t2 = orc.build_bvh(t)
    assert n1.shape == n2.shape and np.array_equal(n1.view(np.uint32), n2.view(np.uint32))
    assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))


def test_bvh_build_matches_committed_fixture():
    d = np.load(GOLDEN / "bvh_build_320.npz")
    n, t = rt.build_bvh(d["tris9"])
    assert np.array_equal(n, d["nodes12"]) and np.array_equal(t, d["tris12"])
    assert rt.build_bvh(np.zeros((0, 9), np.float32))[0].shape[0] == 0


def test_gather_with_custom_matrix(orc):
    rng = np.random.default_rng(2)
    v = rng.normal(size=(50, 3)).astype(np.float32)
    f = rng.integers(0, 50, size=90).astype(np.uint32)
    m = rng.normal(size=16).astype(np.float32)
    assert np.array_equal(rt.gather_triangles(v, f, m), orc.gather_triangles(v, f, m))
    assert rt.gather_triangles(v, f[:4], m).shape[0] == 1      # trailing partial triple ignored (bvh.cpp:234)


def test_obj_reader_round_trip(tmp_path):
    v, f = rt.meshgen.bunny_standin(2)
    path = tmp_path / "m.obj"
    rt.meshgen.write_obj(path, v, f)
    v2, f2 = rt.load_obj(path)
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    # quads (fan), negative indices, v/vt/vn forms, comments
    (tmp_path / "q.obj").write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3//1 4\nf -4 -3 -2\n")
    v3, f3 = rt.load_obj(tmp_path / "q.obj")
    assert v3.shape == (4, 3) and list(f3) == [0, 1, 2, 0, 2, 3, 0, 1, 2]
    with pytest.raises(rt.RtError):
        rt.load_obj(tmp_path / "missing.obj")
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(rt.RtError):
        rt.load_obj(tmp_path / "bad.obj")


def test_png_reader_and_cross_slicing(orc, tmp_path):
    from PIL import Image
    img = rt.load_png(scenes.ASSETS / "Sky_16.png")
    ref = np.asarray(Image.open(scenes.ASSETS / "Sky_16.png"))
    assert img.shape == (1536, 2048, 3) and np.array_equal(img, ref)
    assert np.array_equal(rt.cubemap_from_cross(img), orc.cubemap_from_cross(img))
    rng = np.random.default_rng(4)
    for mode, shape in (("RGBA", (9, 12, 4)), ("L", (6, 8)), ("RGB", (3, 4, 3))):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        Image.fromarray(a, mode).save(tmp_path / f"{mode}.png")
        got = rt.load_png(tmp_path / f"{mode}.png")
        assert np.array_equal(got.reshape(a.shape), a)
    with pytest.raises(rt.RtError):
        rt.cubemap_from_cross(np.zeros((9, 10, 3), np.uint8))        # not 4x3 (cubemap.cpp:47)
    with pytest.raises(rt.RtError):
        rt.load_png(tmp_path / "nope.png")
    (tmp_path / "junk.png").write_bytes(b"not a png at all, really not" * 4)
    with pytest.raises(rt.RtError):
        rt.load_png(tmp_path / "junk.png")


def test_png_reader_matches_the_references_own_decoder(tmp_path):
    """oracle/_ref/libstb_ref.so = the reference's vendored include/stb_image.h, compiled from where it lies under /root/reference by
    `make -C oracle ref` (oracle/stb_ref.c).  stbi_load(path, &w, &h, &n, 0) is the call loadCubeMapFromCross makes
    (src/render/cubemap.cpp:40): both cube-map crosses of the reference and PNGs of every colour type / filter the product's reader
    accepts must decode to the same bytes."""
    import ctypes as C
    from pathlib import Path
    from PIL import Image
    lib = Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "libstb_ref.so"
    if not lib.exists():
        if not Path("/root/reference/include/stb_image.h").exists():
            pytest.skip("oracle/_ref is built from /root/reference, which is not present here")
        import subprocess
        subprocess.run(["make", "-C", str(lib.parent.parent), "ref"], check=True, capture_output=True)
    L = C.CDLL(str(lib))
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]

    def stb(path):
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        p = L.stbi_load(str(path).encode(), C.byref(w), C.byref(h), C.byref(n), 0)
        assert p, path
        a = np.ctypeslib.as_array(p, shape=(h.value, w.value, n.value)).copy()
        L.stbi_image_free(p)
        return a

    for name in ("Sky_01", "Sky_16"):
        want = stb(scenes.ASSETS / f"{name}.png")
        got = rt.load_png(scenes.ASSETS / f"{name}.png")
        assert want.shape == (1536, 2048, 3) and np.array_equal(got, want), name
    rng = np.random.default_rng(9)
    for i, (mode, shape) in enumerate((("RGBA", (31, 17, 4)), ("RGB", (64, 64, 3)), ("L", (5, 9)), ("RGB", (1, 1, 3)))):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        if i == 1:
            a = np.sort(a, axis=1)                      # smooth rows: the encoder picks Sub / Up / Average / Paeth filters
        Image.fromarray(a, mode).save(tmp_path / f"t{i}.png", optimize=bool(i % 2))
        want = stb(tmp_path / f"t{i}.png")
        got = rt.load_png(tmp_path / f"t{i}.png")
        assert np.array_equal(got.reshape(want.shape), want), (mode, shape)


def test_tile_layout_mirror():
    from opengl_raytracing_amd import tiles
    rng = np.random.default_rng(1)
    for (w, h, world) in ((200, 120, 1), (200, 120, 2), (1920, 1080, 8), (33, 17, 3)):
        img = rng.integers(0, 65535, size=(h, w, 4)).astype(np.uint16)
        blocks = [tiles.pack_local(img, r, world) for r in range(world)]
        assert all(b.shape == blocks[0].shape for b in blocks)
        assert np.array_equal(tiles.assemble(blocks, w, h), img)
        masks = [tiles.owner_mask(w, h, r, world) for r in range(world)]
        assert np.array_equal(sum(m.astype(int) for m in masks), np.ones((h, w), int))


def test_host_entry_points_reject_bad_input(tmp_path):
    """nothing thrown across the C ABI, sizes from files are not trusted, indices are range-checked."""
    import struct
    import zlib
    # a PNG whose IHDR claims 1 000 000 x 1 000 000 pixels over a few bytes of IDAT: refused, not std::bad_alloc
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1000000, 1000000, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    (tmp_path / "bomb.png").write_bytes(bomb)
    with pytest.raises(rt.RtError) as e:
        rt.load_png(tmp_path / "bomb.png")
    assert e.value.code == rt.RT_ERR_IO
    # out-of-range vertex index
    with pytest.raises(rt.RtError):
        rt.gather_triangles(np.zeros((3, 3), np.float32), np.array([0, 1, 7], np.uint32))
    # a face record far longer than any fixed line buffer: 3000 vertices in one polygon -> 2998 fan triangles
    n = 3000
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    with open(tmp_path / "ngon.obj", "w") as f:
        for a in ang:
            f.write(f"v {np.cos(a):.6f} {np.sin(a):.6f} 0\n")
        f.write("f " + " ".join(f"{i + 1}/{i + 1}/{i + 1}" for i in range(n)) + "\n")
    v, idx = rt.load_obj(tmp_path / "ngon.obj")
    assert v.shape == (n, 3) and idx.size == (n - 2) * 3 and idx.max() == n - 1
    assert np.array_equal(idx.reshape(-1, 3)[:, 0], np.zeros(n - 2, np.uint32))


def test_tile_deal_scatters_a_ranks_tiles():
    """at 1080p (tilesX = 120) and 4K (240) the number of tile columns is a multiple of nearly every world size, so `tile % world` alone gave
    every rank fixed 16-pixel COLUMNS of the frame.  With the row shift of csrc/rt_frame.hpp (11 tile columns per tile row) every rank owns tiles in every tile
    column and in every tile row, its share differs from the others' by at most one tile, and the numpy mirror round-trips (pack_local -> assemble)."""
    from opengl_raytracing_amd import tiles
    for w, h in ((1920, 1080), (3840, 2160), (640, 360)):
        for world in (2, 3, 4, 5, 6, 8):
            owner, slot = tiles.slot_map(w, h, world)
            tile_owner = owner[::16, ::16]                       # one entry per tile
            counts = np.bincount(tile_owner.ravel(), minlength=world)
            assert counts.max() - counts.min() <= 1, (w, h, world, counts)
            for r in range(world):
                mine = tile_owner == r
                assert mine.any(axis=0).all(), (w, h, world, r, "a tile column without this rank")
                assert mine.any(axis=1).all(), (w, h, world, r, "a tile row without this rank")
                assert mine.all(axis=0).sum() == 0                # and no column it owns alone
            # slots are dense per rank: every slot index below the rank's tile count x 256 is used exactly once
            for r in range(world):
                s = np.sort(slot[owner == r])
                full_tiles = (owner == r).sum()
                assert np.array_equal(np.unique(s), s) and s.size == full_tiles
    # world of one: the plain row-major tile order, as in rounds 1-4
    owner, slot = tiles.slot_map(64, 48, 1)
    assert owner.max() == 0 and slot[0, 16] == 256 and slot[16, 0] == 4 * 256

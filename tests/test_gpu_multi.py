"""GPU: one scene on several GPUs of one process (rt355_scene_create_multi) through the C ABI.  The box has one GPU, so
the device list names it three times: three scenes, three host threads, tiles and frames dealt between them exactly as
between three GPUs.  Pixels are independent given the replicated scene (main.cpp:355-370), so every frame must equal
the single-device frame byte for byte, ray counts included."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import binding as B
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu
KEYS = ("primary_rays", "shadow_rays", "primary_hits", "secondary_rays", "gather_rays", "exact_tests", "overflow_rays")


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    from raytracer_amd import api as A
    A.lib()
    return A


@pytest.mark.parametrize("name", ["default_2lights", "spheres300", "mesh", "mirrors_mesh", "glass", "textured", "environment_spheres"])
def test_multi_frame_equals_single_device_frame(api, name):
    sc, sh, cam, w, h = S.small_case(name)
    one = api.Renderer(sc)
    want, wst = one.render(cam, sh, w, h)
    one.close()
    multi = api.Renderer(sc, devices=[0, 0, 0])
    for gather in (0, 1):
        multi.set_option("gather", gather)
        for pinned in (False, True):
            hp = api.HostPixmap(w, h) if pinned else None
            out = hp.array if pinned else np.full((h, w, 3), 7, dtype=np.uint8)
            out[:] = 7
            got, st = multi.render(cam, sh, w, h, out=out)
            assert np.array_equal(got, want), (name, gather, pinned)
            for k in KEYS:
                assert st[k] == wst[k], (name, gather, pinned, k)
            if hp is not None:
                hp.close()
    multi.close()


def test_multi_slabs_supersampling_and_gather_shaders(api):
    sc, sh, cam, w, h = S.small_case("default_2lights")
    dirs = B.geodesic(B.icosphere(1))
    one = api.Renderer(sc)
    one.set_directions(dirs)
    multi = api.Renderer(sc, devices=[0, 0])
    multi.set_directions(dirs)
    for kw in ({"alias": 2, "seed": 5}, {"row_begin": 17, "row_end": 101}, {"shading": B.SHADING_BLEED, "seed": 3},
               {"shading": B.SHADING_OCCLUSION, "alias": 2, "seed": 9}):
        want, wst = one.render(cam, sh, w, h, out=np.full((h, w, 3), 9, dtype=np.uint8), **kw)
        got, st = multi.render(cam, sh, w, h, out=np.full((h, w, 3), 9, dtype=np.uint8), **kw)
        assert np.array_equal(got, want), kw  # (rows outside the slab keep what the caller had there)
        for k in KEYS:
            assert st[k] == wst[k], (kw, k)
    one.close()
    multi.close()


def test_multi_more_devices_than_tiles(api):
    """A frame of 48 rows has two 32-row tiles: of five scenes, three have nothing to render."""
    sc, sh, cam, w, h = S.small_case("spheres300")
    one = api.Renderer(sc)
    want, wst = one.render(S.default_camera(64, 48), sh, 64, 48)
    one.close()
    multi = api.Renderer(sc, devices=[0] * 5)
    for gather in (0, 1):
        multi.set_option("gather", gather)
        got, st = multi.render(S.default_camera(64, 48), sh, 64, 48)
        assert np.array_equal(got, want) and st["shadow_rays"] == wst["shadow_rays"]
    multi.close()


@pytest.mark.parametrize("name", ["spheres300", "mirrors_mesh", "default_2lights"])
def test_multi_balanced_slabs_equal_single_device_frame(api, name):
    """Option "balance": contiguous slabs whose cuts move from call to call by what the shares took -- the frame stays
    the single-device frame whatever the cuts are, in both gathers, for a slab of the frame, and for frames of fewer
    rows than devices (which go in tiles)."""
    sc, sh, cam, w, h = S.small_case(name)
    one = api.Renderer(sc)
    want, wst = one.render(cam, sh, w, h)
    multi = api.Renderer(sc, devices=[0, 0, 0])
    multi.set_option("balance", 1)
    for gather in (0, 1):
        multi.set_option("gather", gather)
        for call in range(5):  # (the cuts move after every call)
            got, st = multi.render(cam, sh, w, h, out=np.full((h, w, 3), 7, dtype=np.uint8))
            assert np.array_equal(got, want), (name, gather, call)
            for k in KEYS:
                assert st[k] == wst[k], (name, gather, call, k)
    kw = {"row_begin": h // 5, "row_end": h - h // 4}
    want_slab, _ = one.render(cam, sh, w, h, out=np.full((h, w, 3), 9, dtype=np.uint8), **kw)
    for call in range(3):
        got, _ = multi.render(cam, sh, w, h, out=np.full((h, w, 3), 9, dtype=np.uint8), **kw)
        assert np.array_equal(got, want_slab), call
    want_ss, _ = one.render(cam, sh, w, h, alias=2, seed=5)
    for call in range(3):  # (supersampled: the slabs are rows of pixels, each of alias rows of sub-samples)
        got, _ = multi.render(cam, sh, w, h, alias=2, seed=5)
        assert np.array_equal(got, want_ss), call
    tiny, _ = one.render(S.default_camera(64, 2), sh, 64, 2)
    got, _ = multi.render(S.default_camera(64, 2), sh, 64, 2)
    assert np.array_equal(got, tiny)
    with pytest.raises(api.Rt355Error, match="balance"):
        multi.set_option("balance", 2)
    one.close()
    multi.close()


def test_multi_sweep_deals_frames_round_robin(api):
    w, h = 160, 96
    sc = S.mesh_scene(4000, seed=356, tex_size=64)
    sh = S.mesh_shader()
    cams = [S.orbit_camera(9 * f, w, h, n_frames=60) for f in range(7)]
    o = B.opts(w, h)
    one = api.Renderer(sc)
    want = [np.zeros((h, w, 3), dtype=np.uint8) for _ in cams]
    wst = one.render_sweep(cams, sh, o, want)
    one.close()
    multi = api.Renderer(sc, devices=[0, 0, 0])
    pix = [api.HostPixmap(w, h) for _ in cams]
    st = multi.render_sweep(cams, sh, o, [p.array for p in pix])
    for f in range(len(cams)):
        assert np.array_equal(pix[f].array, want[f]), f
        assert st[f]["shadow_rays"] == wst[f]["shadow_rays"] and st[f]["primary_rays"] == w * h
    assert not np.array_equal(want[0], want[3])  # (the camera did move)
    for p in pix:
        p.close()
    multi.close()


def test_multi_arguments(api):
    sc, sh, cam, w, h = S.small_case("default_2lights")
    L = api.lib()
    handle = C.c_void_p()
    packed = sc.pack()
    for devs in ([], [-1], [0, 99]):
        arr = (C.c_int32 * max(1, len(devs)))(*devs)
        assert L.rt355_scene_create_multi(C.byref(packed.desc), arr, len(devs), C.byref(handle)) == -1
    assert L.rt355_scene_create_multi(C.byref(packed.desc), None, 2, C.byref(handle)) == -1
    multi = api.Renderer(sc, devices=[0, 0])
    with pytest.raises(api.Rt355Error, match="tile_count"):
        multi.render(cam, sh, w, h, tile_rows=32, tile_count=2, tile_phase=0)
    import torch
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(api.Rt355Error, match="device pixmap"):
        multi.render_device(cam, sh, B.opts(w, h), fb.data_ptr())
    with pytest.raises(api.Rt355Error, match="gather"):
        multi.set_option("gather", 5)
    with pytest.raises(api.Rt355Error, match="unknown option"):
        multi.set_option("no_such_option", 1)
    multi.set_option("tail_parts", 2)  # reaches every device's scene
    got, _ = multi.render(cam, sh, w, h)
    one = api.Renderer(sc)
    with pytest.raises(api.Rt355Error, match="unknown option"):
        one.set_option("gather", 1)  # (a single-device scene has no gather)
    want, _ = one.render(cam, sh, w, h)
    assert np.array_equal(got, want)
    one.close()
    multi.close()

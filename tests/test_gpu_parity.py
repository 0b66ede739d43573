"""GPU: the HIP library (through the C ABI) against the CPU checker and the golden vectors.

Bar: the reference path is integer colour arithmetic on top of double geometry, and the
library evaluates every accepted intersection with the same double formulas, so frames are
compared BIT-EXACT (stricter than the +-1 LSB on 99.9 % of pixels the metric allows).  The only
operations that are not guaranteed bit-identical between x86 libm and the device maths library
are pow(s,25) in CalcSpecular and acos in the sphere texture mapping; a 1-ulp difference there
moves a colour only if a product lands within ~1e-13 of an integer, so those cases are held
to "at most 0.1 % of pixels differ, never by more than 1 LSB".
"""
import os

import numpy as np
import pytest

from raytracer_amd import binding as B
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    from raytracer_amd import api as A
    A.lib()  # fails loudly if librt355.so is missing
    return A


def _compare(got, want, exact=True):
    diff = (got.astype(np.int16) - want.astype(np.int16))
    npx = int((diff != 0).any(axis=2).sum())
    if exact:
        assert npx == 0, f"{npx} pixels differ (max |d| = {int(np.abs(diff).max())})"
    else:
        assert npx <= max(1, got.shape[0] * got.shape[1] // 1000), f"{npx} pixels differ"
        assert int(np.abs(diff).max()) <= 1


@pytest.mark.parametrize("mode", [B.MODE_ALLPAIRS, B.MODE_BRUTE, B.MODE_PACKET])
@pytest.mark.parametrize("name", S.SMALL_CASES)
def test_small_cases_vs_golden(api, golden_dir, name, mode):
    gold = np.load(os.path.join(golden_dir, f"frame_{name}.npy"))
    sc, sh, cam, w, h = S.small_case(name)
    r = api.Renderer(sc)
    frame, st = r.render(cam, sh, w, h, mode=mode)
    r.close()
    _compare(frame, gold, exact=(name != "textured"))
    assert st["primary_rays"] == w * h


@pytest.mark.parametrize("name", ["mesh", "spheres300", "inside", "default_2lights", "mirrors", "mirrors_mesh"])
def test_ray_counts_match_checker(api, oracle, name):
    sc, sh, cam, w, h = S.small_case(name)
    _, want = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=4)
    r = api.Renderer(sc)
    _, st = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
    r.close()
    assert st["primary_rays"] == want["primary_rays"]
    assert st["shadow_rays"] == want["shadow_rays"]
    assert st["secondary_rays"] == want["secondary_rays"]
    assert (st["secondary_rays"] > 0) == name.startswith("mirrors")
    assert st["overflow_rays"] == 0


def _mirror_spheres(n, seed, n_lights):
    """Random spheres over a mirror floor; every third sphere reflects, a few of them almost perfectly."""
    sc = S.spheres_scene(n, seed=seed)
    for i, o in enumerate(sc.objects):
        if o["type"] == B.OBJ_PLANE:
            o["ks"] = 0.4
        elif i % 3 == 0:
            o["ks"] = 0.9 if i % 9 == 0 else 0.5
    return sc, S.Shader(S.ring_lights(n_lights), 0x101010, 0xDDDDDD, 0x202020)


@pytest.mark.parametrize("mode", [B.MODE_PACKET, B.MODE_ALLPAIRS])
def test_mirror_rays_on_searched_scenes(api, oracle, mode):
    """Reflection (ks != 0, main.cpp:93-112) where the closest hits are searched: the mirror rays of each level go
    through the scene table (k_search_general), their hits are shaded by the same light groups, and the levels are
    folded back.  Frames, mirror-ray and shadow-ray counts equal the CPU checker's; the mesh case also against the
    frame the reference's own classes produced (fixture)."""
    for n, seed, nl, (w, h) in ((300, 91, 2, (160, 120)), (1500, 92, 3, (121, 91)), (64, 93, 9, (80, 60))):
        sc, sh = _mirror_spheres(n, seed, nl)
        cam = S.default_camera(w, h)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        r = api.Renderer(sc)
        got, st = r.render(cam, sh, w, h, mode=mode)
        brute, bst = r.render(cam, sh, w, h, mode=B.MODE_BRUTE)
        r.close()
        _compare(got, want)
        _compare(brute, want)
        assert st["secondary_rays"] == wst["secondary_rays"] == bst["secondary_rays"] > w * h // 20
        assert st["shadow_rays"] == wst["shadow_rays"]
    # supersampled, and slab by slab
    sc, sh = _mirror_spheres(200, 94, 2)
    w, h = 67, 45
    cam = S.default_camera(w, h)
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=2, seed=5), threads=0)
    r = api.Renderer(sc)
    got, st = r.render(cam, sh, w, h, alias=2, seed=5, mode=mode)
    _compare(got, want)
    assert st["secondary_rays"] == wst["secondary_rays"]
    parts = np.zeros_like(want)
    for y0, y1 in ((0, 20), (20, h)):
        r.render(cam, sh, w, h, alias=2, seed=5, mode=mode, row_begin=y0, row_end=y1, out=parts)
    r.close()
    assert np.array_equal(parts, want)


def test_mirrors_supersampled(api, oracle):
    """Reflection rays under alias > 0: every sub-sample runs its own chain of up to three reflections."""
    sc, sh, cam, w, h = S.small_case("mirrors")
    w, h = 81, 61
    cam = S.default_camera(w, h)
    o = B.opts(w, h, alias=2, seed=77)
    want, wst = oracle.render(sc.pack(), cam, sh, o, threads=0)
    r = api.Renderer(sc)
    got, st = r.render(cam, sh, w, h, alias=2, seed=77)
    r.close()
    _compare(got, want)
    assert st["secondary_rays"] == wst["secondary_rays"] > 0


def test_probe_intersections_bit_exact(api, golden_dir):
    """Plane / Sphere / Mesh::CalcIntersect on the device vs the reference's own values."""
    g = np.load(os.path.join(golden_dir, "intersect.npz"))
    for case, rays_key, prefix in (("mesh", "rays", "obj"), ("inside", "rays_inside", "inside_obj")):
        sc, _, _, _, _ = S.small_case(case)
        r = api.Renderer(sc)
        for obj in range(len(sc.objects)):
            want = g[f"{prefix}{obj}"]
            got = r.probe_intersect(obj, g[rays_key])
            hit = want[:, 0] >= 0
            assert np.array_equal(got[:, 0] >= 0, hit)
            assert np.array_equal(got[hit], want[hit]), (case, obj)
        r.close()


def test_mid_size_scenes_vs_checker(api, oracle):
    """Search path on scenes big enough to exercise several LDS tiles and many candidates."""
    dir_lights = [S.light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1)),
                  S.light_src(0, (40, 10, 100), (0.4, 0.3, -1), (1, 1, 1)),
                  S.light_src(2, (0, 20, 90), (0.1, 0.5, -1), (1, 0.9, 0.8))]
    cases = [
        (S.spheres_scene(3000, seed=5), S.Shader(S.ring_lights(4), 0x101010, 0xDDDDDD, 0x000000), 256, 160),
        (S.mesh_scene(4000, seed=9, tex_size=64), S.mesh_shader(), 224, 128),
        (S.spheres_scene(2500, seed=6), S.Shader(dir_lights, 0x101010, 0xDDDDDD, 0x000000), 224, 144),
        (S.mesh_scene(3000, seed=10, tex_size=64), S.Shader(dir_lights[:2], 0x101010, 0xDDDDDD, 0x000000), 192, 128),
    ]
    for sc, sh, w, h in cases:
        cam = S.default_camera(w, h)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        r = api.Renderer(sc)
        got, st = r.render(cam, sh, w, h, mode=B.MODE_ALLPAIRS)
        brute, _ = r.render(cam, sh, w, h, mode=B.MODE_BRUTE)
        packet, pst = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
        r.close()
        _compare(got, want)
        _compare(brute, want)
        _compare(packet, want)
        assert st["shadow_rays"] == wst["shadow_rays"]
        assert st["overflow_rays"] == 0
        # the packet search proposes exactly the candidates of the all-pairs search
        assert pst["exact_tests"] == st["exact_tests"]


@pytest.mark.gpu
@pytest.mark.parametrize("n_lights", [9, 19])
def test_more_lights_than_one_group(api, oracle, n_lights):
    """Lights are shaded in groups of at most 8; the running mean is carried from group to group."""
    lights = S.ring_lights(n_lights)
    lights[3] = S.light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 0.8, 0.6))
    lights[8] = S.light_src(2, (0, 20, 90), (0.1, 0.5, -1), (1, 0.9, 0.8))
    sh = S.Shader(lights, 0x101010, 0xDDDDDD, 0x202020)
    for sc, (w, h) in ((S.spheres_scene(600, seed=11), (160, 96)), (S.default_scene(), (96, 64))):
        cam = S.default_camera(w, h)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        r = api.Renderer(sc)
        for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, mode=mode)
            _compare(got, want)
            assert st["shadow_rays"] == wst["shadow_rays"]
        r.close()


@pytest.mark.parametrize("rays_per_lane", ["1", "4"])
def test_packet_sizes_give_the_same_frames(api, oracle, rays_per_lane):
    """The packet search exists with 64-ray and 256-ray waves; a scene handle normally times both and keeps
    the faster one.  Pinned to either, frames and candidate counts must not change."""
    for name in ("spheres300", "spheres300_dir", "mesh", "mesh_dir", "textured"):
        sc, sh, cam, w, h = S.small_case(name)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        r = api.Renderer(sc)
        r.set_option("packet_rays", int(rays_per_lane))
        got, st = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
        ref, rst = r.render(cam, sh, w, h, mode=B.MODE_ALLPAIRS)
        r.close()
        _compare(got, want)
        assert st["exact_tests"] == rst["exact_tests"]


@pytest.mark.parametrize("threads", [64, 128, 256])
def test_workgroup_sizes_give_the_same_frames(api, oracle, threads):
    """The shading kernel and the one-kernel renderers of refracted and mirror rays run in workgroups of one wave by
    default ("shade_block", "direct_block"); two or four waves must give the same bytes and ray counts, with and
    without supersampling (whose frame ends in k_average), and for a frame whose width is no multiple of the size."""
    for name, kw in (("spheres300", {}), ("mesh", {"alias": 2, "seed": 4}), ("glass", {}), ("mirrors", {}), ("mirrors_mesh", {}),
                     ("glass", {"alias": 2, "seed": 7})):
        sc, sh, cam, w, h = S.small_case(name)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, **kw), threads=0)
        r = api.Renderer(sc)
        r.set_option("shade_block", threads)
        r.set_option("direct_block", threads)
        got, st = r.render(cam, sh, w, h, **kw)
        _compare(got, want)
        assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"], name
        w2, h2 = 157, 61
        want2, _ = oracle.render(sc.pack(), S.default_camera(w2, h2), sh, B.opts(w2, h2), threads=0)
        got2, _ = r.render(S.default_camera(w2, h2), sh, w2, h2)
        _compare(got2, want2)
        r.close()
    r = api.Renderer(S.small_case("glass")[0])
    with pytest.raises(api.Rt355Error, match="shade_block"):
        r.set_option("shade_block", 96)
    r.close()


def test_frames_do_not_change_while_the_packet_size_is_tuned(api):
    """Frames 1 and 2 of a handle try the two packet sizes, later ones use the faster: same bytes throughout."""
    sc, sh, cam, w, h = S.small_case("mesh_dir")
    r = api.Renderer(sc)
    frames = [r.render(cam, sh, w, h)[0] for _ in range(4)]
    r.close()
    for f in frames[1:]:
        assert np.array_equal(frames[0], f)


@pytest.mark.parametrize("alias", [1, 2, 3])
def test_supersampling_vs_checker(api, oracle, alias):
    """alias > 0 (CalcPixel's sub-sample loop, main.cpp:289-325): jittered sub-samples from the counter-based
    generator both sides share, averaged per pixel.  Parity is against the CPU checker, which the reference's
    own Color arithmetic pins for this path (test_oracle_golden.py); libc's rand() stream is not reproduced."""
    for name in ("default_2lights", "textured", "mesh", "spheres300", "spheres300_dir"):
        sc, sh, cam, w, h = S.small_case(name)
        w, h = (w // 2) | 1, (h // 2) | 1  # odd sizes: partial tiles of the sub-sample grid
        cam = S.default_camera(w, h)
        o = B.opts(w, h, alias=alias, seed=1234 + alias)
        want, wst = oracle.render(sc.pack(), cam, sh, o, threads=0)
        r = api.Renderer(sc)
        for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, alias=alias, mode=mode, seed=1234 + alias)
            _compare(got, want)
            assert st["primary_rays"] == wst["primary_rays"] == w * h * alias * alias
            assert st["shadow_rays"] == wst["shadow_rays"]
        # row slabs of a supersampled frame
        parts = np.zeros_like(want)
        for y0, y1 in ((0, 7), (7, h)):
            r.render(cam, sh, w, h, alias=alias, row_begin=y0, row_end=y1, out=parts, seed=1234 + alias)
        r.close()
        assert np.array_equal(parts, want)


def test_orbit_cameras_mesh(api, oracle):
    """Other view points (the camera sweep): eye bundle tables are rebuilt per frame."""
    sc = S.mesh_scene(2000, seed=3, tex_size=32)
    sh = S.mesh_shader()
    r = api.Renderer(sc)
    for frame in (0, 7, 23, 41):
        w, h = 128, 96
        cam = S.orbit_camera(frame, w, h)
        want, _ = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        got, _ = r.render(cam, sh, w, h)
        _compare(got, want)
    r.close()


def test_slabs_equal_whole_frame(api):
    """Row slabs (the multi-GPU partition) must reproduce the one-call frame byte for byte."""
    sc, sh, cam, w, h = S.small_case("spheres300")
    r = api.Renderer(sc)
    full, _ = r.render(cam, sh, w, h)
    parts = np.zeros_like(full)
    for y0, y1 in ((0, 17), (17, 64), (64, 65), (65, h)):
        r.render(cam, sh, w, h, row_begin=y0, row_end=y1, out=parts)
    r.close()
    assert np.array_equal(full, parts)


def test_full_size_properties(api, oracle):
    """3840x2160 default scene: known pixels recorded from the reference main.cpp path, hash equal to
    the CPU checker's, and slab/whole-frame idempotence at full size."""
    w, h = 3840, 2160
    sc, sh, cam = S.default_scene(), S.default_shader(1), S.default_camera(w, h)
    r = api.Renderer(sc)
    got, st = r.render(cam, sh, w, h)
    samples = {(0, 0): 181, (1920, 1080): 11, (960, 1080): 157, (2880, 720): 180, (1920, 270): 181,
               (3839, 2159): 161}
    for (x, y), v in samples.items():
        assert list(got[y, x]) == [v, v, v]
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    assert oracle.fnv(got) == oracle.fnv(want)
    assert st["shadow_rays"] == wst["shadow_rays"]
    again = np.zeros_like(got)
    r.render(cam, sh, w, h, row_begin=0, row_end=1000, out=again)
    r.render(cam, sh, w, h, row_begin=1000, row_end=h, out=again)
    assert np.array_equal(again, got)
    r.close()


@pytest.mark.parametrize("name", ["glass", "quadrics", "main_all", "environment", "solid_texture"])
def test_full_size_every_object_class(api, oracle, name):
    """The small cases that hold the other object classes, texture types and refraction, at the metric's 3840x2160:
    whole frame and ray counts equal to the CPU checker's (these scenes are unsearched: one kernel per frame)."""
    w, h = 3840, 2160
    sc, sh, _, _, _ = S.small_case(name)
    cam = S.default_camera(w, h)
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    got, st = r.render(cam, sh, w, h)
    r.close()
    _compare(got, want, exact=(name != "environment"))  # (its sphere mapping goes through acos: <= 0.1 % pixels, 1 unit)
    assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"]


def test_full_size_gather(api, oracle):
    """The renderer HEAD of the reference selects, at the metric's size: the default scene at 3840x2160 under
    shading3 with 42 gather directions (350 M rays), whole-frame hash and ray count equal to the CPU checker's;
    and with 162 directions two row slabs equal the frame rendered at once."""
    w, h = 3840, 2160
    sc, sh, cam = S.default_scene(), S.default_shader(1), S.default_camera(w, h)
    r = api.Renderer(sc)
    dirs = B.geodesic(B.icosphere(1))
    r.set_directions(dirs)
    got, st = r.render(cam, sh, w, h, seed=2024, shading=B.SHADING_BLEED)
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, seed=2024, shading=B.SHADING_BLEED), threads=0, dirs=dirs)
    assert st["gather_rays"] == wst["gather_rays"] > 100_000_000
    assert oracle.fnv(got) == oracle.fnv(want)
    r.set_directions(B.geodesic(B.icosphere(2)))
    whole, _ = r.render(cam, sh, w, h, seed=7, shading=B.SHADING_BLEED)
    parts = np.zeros_like(whole)
    r.render(cam, sh, w, h, row_begin=0, row_end=777, out=parts, seed=7, shading=B.SHADING_BLEED)
    r.render(cam, sh, w, h, row_begin=777, row_end=h, out=parts, seed=7, shading=B.SHADING_BLEED)
    assert np.array_equal(parts, whole)
    r.close()


def test_error_behaviour(api):
    # a glass sphere alone: a ray that leaves it and hits nothing goes on from the origin with its coefficients
    # for ever (main.cpp:94,116) -- the reference overflows its stack (the checker and the harness around the
    # reference's classes both stop at a depth of 2000), the library reports the frame
    sc = S.Scene()
    sc.add_sphere((0, 10, 0), 6, (100, 200, 100))
    sc.objects[0]["fresnel"] = 0.5; sc.objects[0]["ior"] = 1.5
    r = api.Renderer(sc)
    with pytest.raises(api.Rt355Error, match="did not end"):
        r.render(S.default_camera(64, 48), S.default_shader(1), 64, 48)
    got, _ = r.render(S.default_camera(64, 48), S.default_shader(1), 64, 48, shading=B.SHADING_OCCLUSION)  # no recursion there
    r.close()
    sc = S.default_scene()
    sc.set_texture_type(2, 4)
    with pytest.raises(api.Rt355Error, match="texture_type"):
        api.Renderer(sc)
    for kind in (1, 2, 3):   # the reference dereferences textures[0] (and [1]) of such an object
        sc = S.default_scene()
        sc.set_texture_type(2, kind)
        with pytest.raises(api.Rt355Error, match="needs textures"):
            api.Renderer(sc)
    sc = S.default_scene()
    r = api.Renderer(sc)
    with pytest.raises(api.Rt355Error, match="alias"):
        r.render(S.default_camera(64, 48), S.default_shader(1), 64, 48, alias=-1)
    with pytest.raises(api.Rt355Error, match="ARG"):
        r.render(S.default_camera(64, 48), S.default_shader(1), 64, 48, row_begin=40, row_end=20)
    with pytest.raises(api.Rt355Error, match="shading"):
        r.render(S.default_camera(64, 48), S.default_shader(1), 64, 48, shading=7)
    L = api.lib()
    assert L.rt355_scene_set_directions(r.handle, None, 5) == -1 and b"bad argument" in L.rt355_last_error()
    assert L.rt355_scene_set_directions(r.handle, None, -1) == -1
    assert L.rt355_scene_set_directions(None, None, 0) == -1
    r.set_directions(np.zeros((0, 3)))  # clearing the set is allowed
    r.close()


def test_more_candidates_than_slots_spill(api, oracle):
    """More than 32 primitives on one line of sight: a ray's candidate slots overflow and the rest goes through
    the bundle's spill list -- the frame must still be exact, for spheres and for stacked mesh triangles."""
    sc = S.Scene()
    sc.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
    for i in range(48):      # nested on the view axis, all hit by the central rays
        sc.add_sphere((0.0, 5.0 + 0.75 * i, 0.0), 2.0 + 0.01 * i, (40 + 4 * i, 200 - 3 * i, 90))
    for i in range(40):      # filler so that the search path (>= 64 primitives) is taken
        sc.add_sphere((-30.0 + 1.5 * i, 60.0, 20.0), 0.5, (200, 200, 50))
    # 45 camera-facing quads behind one another (90 front-facing triangles on the central lines of sight)
    tris = []
    for i in range(45):
        y, a = 8.0 + 0.5 * i, 3.0 + 0.05 * i
        p00, p10, p11, p01 = (-a - 12, y, -a), (a - 12, y, -a), (a - 12, y, a), (-a - 12, y, a)
        tris.append(list(p00) + list(p10) + list(p11) + [0, 0, 1, 0, 1, 1])
        tris.append(list(p00) + list(p11) + list(p01) + [0, 0, 1, 1, 0, 1])
    sc.add_mesh(np.array(tris, dtype=np.float64), (90, 160, 220))
    sh = S.Shader([S.light_src(1, (0.0, -30.0, 0.5), (0, 0, -1), (1, 1, 1)),
                   S.light_src(0, (-100, -30, 100), (0.2, 1, -0.1), (1, 1, 1))], 0x101010, 0xDDDDDD, 0x000000)
    w, h = 96, 64
    cam = S.default_camera(w, h)
    want, _ = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS):
        got, st = r.render(cam, sh, w, h, mode=mode)
        assert st["overflow_rays"] > 0
        _compare(got, want)
    r.close()


def test_cpp_host_demo_matches_checker(api, oracle, tmp_path):
    """raytracer_amd/host/rt355_demo (C++ mirror of the reference main()) end to end."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracer_amd", "host", "rt355_demo")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    prefix = str(tmp_path / "demo")
    subprocess.run([exe, "320", "240", "2", prefix], check=True, timeout=100)
    got = np.fromfile(prefix + ".rgb", dtype=np.uint8).reshape(240, 320, 3)
    want, _ = oracle.render(S.default_scene().pack(), S.default_camera(320, 240), S.default_shader(2),
                            B.opts(320, 240), threads=0)
    _compare(got, want)
    bmp = open(prefix + ".bmp", "rb").read()
    assert bmp[:2] == b"BM" and len(bmp) == 54 + 320 * 240 * 3
    # BMP rows are bottom-up B,G,R: first pixel of the file = pixel (0,0) of the pixmap
    assert list(bmp[54:57]) == list(want[0, 0][::-1])
    # HEAD's renderer: gather directions read from an OBJ file by the host's Geodesic() (main.cpp:408-450)
    verts = [(1.5 * x + 0.25, 1.5 * y - 0.5, 1.5 * z + 1.0) for x, y, z in B.icosphere(1)]  # off-centre, not unit
    obj = tmp_path / "geodesic.obj"
    obj.write_text("# directions\n" + "".join(f"v {x!r} {y!r} {z!r}\n" for x, y, z in verts) + "f 1 2 3\n")
    subprocess.run([exe, "160", "120", "1", prefix + "_g", str(obj)], check=True, timeout=100)
    got = np.fromfile(prefix + "_g.rgb", dtype=np.uint8).reshape(120, 160, 3)
    dirs = B.read_geodesic(str(obj))
    assert dirs.shape == (42, 3) and np.array_equal(dirs, B.geodesic(verts))
    want, _ = oracle.render(S.default_scene().pack(), S.default_camera(160, 120), S.default_shader(1),
                            B.opts(160, 120, shading=B.SHADING_BLEED), threads=0, dirs=dirs)
    _compare(got, want)


def test_cpp_host_classes_of_every_kind(api, oracle, tmp_path):
    """The C++ host's mirror of the reference's classes beyond main()'s four objects: its `all` scene (main()'s
    whole object list plus a glass sphere, a solid texture with its own projector read from a BMP file, an
    Ellipsoid and a Cylinder) against the same scene built through the Python host and rendered by the checker."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracer_amd", "host", "rt355_demo")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    tex = S._small_texture(32, 16, 77)
    bmp = tmp_path / "picture.bmp"
    B.write_bmp(str(bmp), tex)
    prefix = str(tmp_path / "all")
    subprocess.run([exe, "200", "150", "2", prefix, "all", str(bmp)], check=True, timeout=100)
    got = np.fromfile(prefix + ".rgb", dtype=np.uint8).reshape(150, 200, 3)
    n = B._normalize
    s = S.Scene()
    s.add_plane((0, 13, 0), (0, -1, 0), (-1, 0, 0), (200, 200, 200))
    s.add_plane((0, -100, 0), (0, 1, 0), (-1, 0, 0), (214, 147, 44))
    s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
    s.add_plane((0, 0, 100), (0, 0, -1), (1, 0, 0), (255, 0, 0))
    s.add_plane((100, 0, 0), (-1, 0, 0), (0, 1, 0), (0, 255, 0))
    s.add_plane((-100, 0, 0), (1, 0, 0), (0, 1, 0), (0, 255, 0))
    s1 = s.add_sphere((-10.5, 0, 0), 10, (100, 200, 100))
    frame = (n((1, 0.2, 0)), n((-0.1, 0.1, 1)), n((0.1, -1, 0.2)), (-8, 0, 2))
    s.add_texture(s1, tex, 24.0, 18.0, frame=frame); s.add_texture(s1, tex, 24.0, 18.0, frame=frame)
    s.set_texture_type(s1, 2)
    s2 = s.add_sphere((10.5, 0, 0), 10, (200, 100, 100), ks=0.3)
    s.objects[s2]["fresnel"] = 0.3; s.objects[s2]["ior"] = 1.5
    s3 = s.add_sphere((-20, -20, 20), 10, (73, 179, 248), ks=1)
    s.add_texture(s3, tex, 0.5, 0.5); s.add_texture(s3, tex, 0.5, 0.5)
    s.set_texture_type(s3, 3)
    env = s.add_environment((0, -50, 20), 1000, (255, 255, 0), (-1, 1, 0), (-1, -1, 0), (0, 0, 1))
    s.add_texture(env, tex, 1, 1); s.add_texture(env, tex, 1, 1)
    s.set_texture_type(env, 1)
    s.add_quadric("Ellipsoid", (0, -6, -5), (1, 0, 0), (0, 1, 0), (0, 0, 1), (4, 2.5, 3), (220, 200, 80), ks=0.2)
    s.add_quadric("Cylinder", (30, 5, 0), n((1, 0.1, 0)), n((-0.1, 1, 0.2)), n((0, -0.2, 1)), (2.2, 2.2, 1), (90, 200, 90))
    want, wst = oracle.render(s.pack(), S.default_camera(200, 150), S.default_shader(2), B.opts(200, 150), threads=0)
    assert wst["secondary_rays"] > 0
    _compare(got, want)


def _random_case(seed):
    """A random mix of planes, spheres and meshes lit by random lights of every type."""
    rng = S.SplitMix64(9000 + seed)
    sc = S.Scene()
    if rng.uniform() < 0.8:
        sc.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
    if rng.uniform() < 0.5:
        sc.add_plane((0, 40, 0), (0, -1, 0), (-1, 0, 0), (200, 200, 200))
    n_sph = int(rng.uniform(70, 200))
    for _ in range(n_sph):
        c = (rng.uniform(-25, 25), rng.uniform(-5, 35), rng.uniform(-9, 15))
        sc.add_sphere(c, rng.uniform(0.2, 3.5), (int(rng.uniform(30, 255)), int(rng.uniform(30, 255)), int(rng.uniform(30, 255))))
    for m in range(int(rng.uniform(0, 3))):
        n = int(rng.uniform(3, 9))
        tris = S.heightfield_tris(n, n, seed=seed * 7 + m, x0=rng.uniform(-25, -5), x1=rng.uniform(5, 25),
                                  y0=rng.uniform(-5, 5), y1=rng.uniform(10, 30), zbase=rng.uniform(-8, 4),
                                  amp=rng.uniform(0.5, 4))
        if rng.uniform() < 0.3:   # flip the winding of some meshes: back faces toward the camera
            tris = tris[:, [0, 1, 2, 6, 7, 8, 3, 4, 5, 9, 10, 13, 14, 11, 12]]
        sc.add_mesh(tris, (int(rng.uniform(30, 255)), 200, 110))
    lights = []
    for _ in range(int(rng.uniform(1, 4))):
        t = int(rng.uniform(0, 4))
        pos = (rng.uniform(-60, 60), rng.uniform(-40, 40), rng.uniform(10, 90))
        d = (rng.uniform(-1, 1), rng.uniform(-1, 1), -rng.uniform(0.2, 1))
        lights.append(S.light_src(t, pos, d, (rng.uniform(0.5, 1), rng.uniform(0.5, 1), rng.uniform(0.5, 1))))
    sh = S.Shader(lights, 0x101418, 0xDDDDDD, 0x202020)
    eye = (rng.uniform(-10, 10), rng.uniform(-30, -12), rng.uniform(-4, 12))
    view = (rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.2))
    k = int(os.environ.get("RT355_TEST_SCALE", "1"))  # tools/parity_sweep.py renders the same scenes larger
    w, h = 112 * k, 80 * k
    return sc, sh, B.camera(eye, view, (0, 0, 1), 10.0, w, h), w, h


@pytest.mark.parametrize("seed", range(10))
def test_random_scenes_search_is_conservative(api, oracle, seed):
    """The fp32 search must never lose a pair the fp64 formulas accept: FAST == checker == BRUTE."""
    sc, sh, cam, w, h = _random_case(seed)
    assert sc.n_primitives >= 64
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    got, st = r.render(cam, sh, w, h, mode=B.MODE_ALLPAIRS)
    packet, pst = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
    r.close()
    _compare(got, want)
    _compare(packet, want)
    assert st["shadow_rays"] == wst["shadow_rays"]
    assert pst["exact_tests"] == st["exact_tests"]


@pytest.mark.parametrize("seed", range(10, 22))
def test_random_scenes_with_mirrors(api, oracle, seed):
    """The same random scenes with random reflection coefficients (object 0 included, so that rays that miss go
    on, main.cpp:94), overlapping spheres, mirror meshes, eyes inside spheres: the search of the mirror rays
    must lose nothing either.  Every third seed is scaled up a thousandfold and moved far from the origin, where
    fp32 has few digits left for the geometry."""
    sc, sh, cam, w, h = _random_case(seed)
    rng = S.SplitMix64(777 + seed)
    for i, o in enumerate(sc.objects):
        u = rng.uniform()
        if i == 0 or u < 0.4:
            o["ks"] = (0.3, 0.55, 0.8, 0.97, 1.0)[int(rng.uniform(0, 5))]
    if seed % 3 == 0:
        k, shift = 1000.0, np.array([3.0e5, -2.0e5, 1.0e5])
        for o in sc.objects:
            o["p"] = o["p"] * k + shift
            o["radius"] = o["radius"] * k
        sc.triangles = [np.concatenate([(t[:, :9].reshape(-1, 3, 3) * k + shift).reshape(-1, 9), t[:, 9:]], axis=1) for t in sc.triangles]
        for l in sh.lights:
            l["position"] = l["position"] * k + shift
        cam.p_eye[:] = list(np.array(list(cam.p_eye)) * k + shift)
        cam.dist *= k; cam.s_x *= k; cam.s_y *= k
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    got, st = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
    r.close()
    _compare(got, want)
    assert st["secondary_rays"] == wst["secondary_rays"] > 0
    assert st["shadow_rays"] == wst["shadow_rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(30, 38))
def test_random_scenes_inside_an_environment(api, oracle, seed):
    """The random scenes inside an Environment (anyobject.h:171-225, the sphere seen from within, unlit:
    shader.h:70-72): placed first or last in the list, textured or flat, reflective or not, with and without
    other mirrors, the eye inside it; shadow rays of lights inside it always end on it or before it, those of
    directional lights too.  Searched (PACKET, ALLPAIRS) and unsearched (BRUTE) paths against the checker, the
    gather shaders on the odd seeds."""
    sc, sh, cam, w, h = _random_case(seed)
    rng = S.SplitMix64(4242 + seed)
    env = sc.add_environment((rng.uniform(-5, 5), rng.uniform(0, 20), rng.uniform(-5, 5)), rng.uniform(150, 900),
                             (int(rng.uniform(30, 255)), int(rng.uniform(30, 255)), 240),
                             (rng.uniform(-1, 1), 1, 0), (0, rng.uniform(-1, 1), 1), (1, 0, rng.uniform(-1, 1)),
                             ks=(0.0, 0.5)[seed % 2])
    if seed % 4 < 2:
        sc.add_texture(env, S._small_texture(32, 16, seed), 1, 1)
        if seed % 4 == 1:
            sc.add_texture(env, S._small_texture(16, 16, seed + 1), 1, 1)
            sc.set_texture_type(env, 1)
    if seed % 3 == 0:      # the environment as object 0: rays that miss everything still end on it
        sc.objects.insert(0, sc.objects.pop())
    if seed >= 34:
        for i, o in enumerate(sc.objects):
            if i % 5 == 2:
                o["ks"] = 0.6
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
        got, st = r.render(cam, sh, w, h, mode=mode)
        _compare(got, want)
        assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"], mode
    if seed % 2:
        dirs = B.geodesic(B.icosphere(0))
        r.set_directions(dirs)
        shading = (B.SHADING_BLEED, B.SHADING_OCCLUSION)[(seed // 2) % 2]
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=2, seed=seed, shading=shading), threads=0, dirs=dirs)
        for mode in (B.MODE_PACKET, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, alias=2, seed=seed, shading=shading, mode=mode)
            _compare(got, want)
            assert st["gather_rays"] == wst["gather_rays"]
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(40, 48)) + [1822, 3248])  # mirror chains along a paraboloid: hits 1e19 and 1e10 away
def test_random_scenes_with_quadrics(api, oracle, seed):
    """The random scenes with two to four random quadrics (anyobject.h:231-442: the five subclasses, the default
    "plane" and general integer coefficients, axes neither unit nor orthogonal, float scales), some of them
    mirrors: candidates of every ray in the searched modes, one loop of their own in the unsearched one.  All
    three modes against the checker, the gather shaders on the odd seeds."""
    sc, sh, cam, w, h = _random_case(seed)
    rng = S.SplitMix64(5151 + seed)
    kinds = list(B.QUADRIC_KINDS)
    for _ in range(int(rng.uniform(2, 5))):
        kind = kinds[int(rng.uniform(0, len(kinds)))] if rng.uniform() < 0.8 else tuple(int(rng.uniform(-2, 3)) for _ in range(5))
        axes = [(1 + rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)),
                (rng.uniform(-0.3, 0.3), 1 + rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3)),
                (rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1 + rng.uniform(-0.2, 0.2))]
        sc.add_quadric(kind, (rng.uniform(-20, 20), rng.uniform(0, 30), rng.uniform(-8, 10)), *axes,
                       (rng.uniform(1, 6), rng.uniform(1, 6), rng.uniform(0.3, 6)),
                       (int(rng.uniform(30, 255)), int(rng.uniform(30, 255)), int(rng.uniform(30, 255))),
                       ks=(0.0, 0.0, 0.5, 0.9)[int(rng.uniform(0, 4))])
    if seed % 3 == 0:      # a quadric as object 0
        sc.objects.insert(0, sc.objects.pop())
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
        got, st = r.render(cam, sh, w, h, mode=mode)
        _compare(got, want)
        assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"], mode
    if seed % 2:
        dirs = B.geodesic(B.icosphere(0))
        r.set_directions(dirs)
        shading = (B.SHADING_BLEED, B.SHADING_OCCLUSION)[(seed // 2) % 2]
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=2, seed=seed, shading=shading), threads=0, dirs=dirs)
        for mode in (B.MODE_PACKET, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, alias=2, seed=seed, shading=shading, mode=mode)
            _compare(got, want)
            assert st["gather_rays"] == wst["gather_rays"]
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(50, 56)) + [12233])  # 12233: a finite chain with more than 64 calls waiting
def test_random_glass_in_a_room(api, oracle, seed):
    """Refraction (main.cpp:116-154): the random scenes closed by an Environment, with random objects of glass
    (spheres, planes, meshes, quadrics; fresnel 0..0.9, IOR 1.05..1.8, some of them mirrors too).  Every mode
    renders such a scene unsearched, one thread per pixel; frame and ray counts against the checker, which is
    pinned on the reference's own classes by the fixtures `glass` and `glass_mixed`."""
    sc, sh, cam, w, h = _random_case(seed)
    rng = S.SplitMix64(6161 + seed)
    sc.add_quadric("Ellipsoid", (rng.uniform(-10, 10), rng.uniform(5, 20), rng.uniform(-5, 5)), (1, 0, 0), (0, 1, 0), (0, 0, 1),
                   (rng.uniform(2, 5), rng.uniform(2, 5), rng.uniform(2, 5)), (200, 120, 90))
    sc.add_environment((0, 10, 0), 400, (110, 150, 235))
    n_glass = 0
    for i, o in enumerate(sc.objects[:-1]):
        if rng.uniform() < 0.25 or i == len(sc.objects) - 2:
            o["fresnel"] = rng.uniform(0.0, 0.9); o["ior"] = rng.uniform(1.05, 1.8)
            if rng.uniform() < 0.4:
                o["ks"] = rng.uniform(0.2, 0.9)
            n_glass += 1
    assert n_glass > 0
    w, h = 80 * w // 112, 60 * w // 112
    cam = B.camera(tuple(cam.p_eye), tuple(cam.v_view), (0, 0, 1), 10.0, w, h)
    for alias in (0, 2):
        try:
            want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=seed), threads=0)
        except RuntimeError:   # a chain that does not end (the checker stops at depth 2000): the library must say so too
            r = api.Renderer(sc)
            with pytest.raises(api.Rt355Error, match="did not end"):
                r.render(cam, sh, w, h, alias=alias, seed=seed)
            r.close()
            continue
        r = api.Renderer(sc)
        for mode in (B.MODE_PACKET, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, mode=mode, alias=alias, seed=seed)
            _compare(got, want)
            assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"] > 0, mode
        r.close()


def glass_among_spheres(api, oracle, seed, n_spheres, tiles):
    """n_spheres random spheres, every tenth of glass, some of those mirrors too, under three lights inside an
    Environment: frames and ray counts of every mode against the checker, supersampled too (and in row tiles)."""
    sc = S.spheres_scene(n_spheres, seed=seed)
    for i, o in enumerate(sc.objects):
        if i and i % 10 == 0:
            o["fresnel"] = 0.3 if i % 20 else 0.6
            o["ior"] = 1.5 if i % 30 else 1.2
            if i % 40 == 0:
                o["ks"] = 0.4
    sc.objects[0]["ks"] = 0.3
    sc.add_environment((0, 60, 10), 600, (110, 150, 235))
    sh = S.Shader(S.ring_lights(3), 0x101010, 0xDDDDDD, 0x202020)
    w, h = 160, 120
    cam = S.default_camera(w, h)
    r = api.Renderer(sc)
    for alias in (0, 2):
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=8), threads=0)
        assert wst["secondary_rays"] > w * h // 50
        for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, alias=alias, seed=8, mode=mode)
            _compare(got, want)
            assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"], (alias, mode)
            assert (st["pair_tests"] > 0) == (mode != B.MODE_BRUTE)   # the table was walked
    if tiles:
        got = np.zeros_like(want)
        for g in range(3):
            r.render(cam, sh, w, h, alias=2, seed=8, out=got, tile_rows=16, tile_count=3, tile_phase=g)
        _compare(got, want)
    r.close()


@pytest.mark.gpu
def test_glass_among_many_spheres(api, oracle):
    """Refraction where the scene is searched: 1 500 spheres, every tenth of glass, some of those mirrors too, under
    three lights inside an Environment.  The searched modes walk the scene table for every ray of a pixel's tree
    (camera, mirror, refracted and shadow rays); BRUTE tests every primitive.  Frames and ray counts against the
    checker, supersampled and in row tiles as well."""
    glass_among_spheres(api, oracle, 97, 1500, True)


def glass_among_spheres_seeded(api, oracle, seed):
    """(for tools/parity_sweep.py) the same with 150 .. 650 spheres of another seed"""
    glass_among_spheres(api, oracle, seed, 150 + 100 * (seed % 6), False)


@pytest.mark.gpu
def test_triangle_whose_inside_test_is_noise(api, oracle):
    """Triangle::CalcIntersect takes u and v from the first component of (p1 - p0) x (p2 - p0) that is not zero
    (anyobject.h:512-513).  These triangles lie in a slanted plane that contains the x direction: that component
    is 7e-15 beside 310, rounding noise, so u and v are noise and the reference takes or leaves points anywhere on
    the plane.  The searched modes must give what the walk over everything gives: such a triangle is a candidate of
    every ray (camera, shadow, mirror, gather)."""
    p0, p1, p2 = (-27.38, 17.98, -6.56), (17.62, 33.64, -1.34), (-27.38, 24.88, -4.26)
    tris = np.zeros((2, 15))
    tris[0, :9] = p0 + p1 + p2
    tris[1, :9] = p0 + p2 + p1   # the other winding: one of the two faces the camera
    sc = S.Scene()
    sc.add_plane((0, 0, -12), (0, 0, 1), (1, 0, 0), (200, 200, 210), ks=0.5)
    sc.add_mesh(tris, clr=(230, 90, 60))
    rng = S.SplitMix64(99)
    for i in range(70):
        sc.add_sphere((rng.uniform(-25, 25), rng.uniform(10, 45), rng.uniform(-10, 6)), rng.uniform(0.6, 1.8),
                      (int(rng.uniform(40, 255)), int(rng.uniform(40, 255)), int(rng.uniform(40, 255))))
    sh = S.Shader(S.ring_lights(2) + [B.light_src(0, (0, 0, 0), (0.2, 0.3, -1.0), (1, 1, 1))], 0x101010, 0xDDDDDD, 0x202020)
    w, h = 160, 120
    cam = S.default_camera(w, h)
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    mesh_pixels = int((np.abs(want.astype(int) - np.array([230, 90, 60])).sum(axis=2) < 200).sum())
    r = api.Renderer(sc)
    for mode in (B.MODE_BRUTE, B.MODE_PACKET, B.MODE_ALLPAIRS):
        got, st = r.render(cam, sh, w, h, mode=mode)
        _compare(got, want)
        assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"], mode
    dirs = B.geodesic(B.icosphere(1))
    r.set_directions(dirs)
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, seed=5, shading=B.SHADING_BLEED), threads=0, dirs=dirs)
    got, st = r.render(cam, sh, w, h, seed=5, shading=B.SHADING_BLEED, mode=B.MODE_PACKET)
    _compare(got, want)
    assert st["gather_rays"] == wst["gather_rays"] > 0
    r.close()
    assert mesh_pixels >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_spheres", [0, 1, 7, 8, 9, 63, 64, 65, 511, 513])
def test_scene_table_of_every_depth(api, oracle, n_spheres):
    """The scene table the mirror, glass and gather rays walk is an 8-ary tree over the spheres and triangles:
    scenes whose tree is absent (planes only), one group, and just below / at / above a power of 8, padded with
    planes to the 64 primitives from which a scene is searched.  Mirrors, glass and both gather shaders against
    the checker."""
    rng = S.SplitMix64(4242 + n_spheres)
    sc = S.Scene()
    sc.add_plane((0, 0, -12), (0, 0, 1), (1, 0, 0), (200, 200, 210), ks=0.4)
    sc.add_plane((0, 60, 0), (0, -1, 0), (1, 0, 0), (210, 180, 160))
    for k in range(max(0, 64 - n_spheres - 2)):   # more planes, all behind the camera
        sc.add_plane((0, -200 - k, 0), (0, 1, 0), (1, 0, 0), (90, 90, 90))
    for i in range(n_spheres):
        c = (rng.uniform(-22, 22), rng.uniform(18, 50), rng.uniform(-10, 14))
        sc.add_sphere(c, rng.uniform(0.8, 3.0) if n_spheres > 20 else rng.uniform(2.5, 6.0),
                      (int(rng.uniform(40, 255)), int(rng.uniform(40, 255)), int(rng.uniform(40, 255))),
                      ks=0.5 if i % 3 == 0 else 0.0)
    assert sc.n_primitives >= 64
    sh = S.Shader(S.ring_lights(2), 0x101010, 0xDDDDDD, 0x202020)
    w, h = 96, 64
    cam = S.default_camera(w, h)
    r = api.Renderer(sc)
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    got, st = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
    _compare(got, want)
    assert st["secondary_rays"] == wst["secondary_rays"] > 0 and st["shadow_rays"] == wst["shadow_rays"]
    dirs = B.geodesic(B.icosphere(1))
    r.set_directions(dirs)
    for shading in (B.SHADING_BLEED, B.SHADING_OCCLUSION):
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, seed=3, shading=shading), threads=0, dirs=dirs)
        got, st = r.render(cam, sh, w, h, seed=3, shading=shading, mode=B.MODE_PACKET)
        _compare(got, want)
        assert st["gather_rays"] == wst["gather_rays"] > 0
    r.close()
    if n_spheres >= 7:   # the same scene with every other sphere of glass, in a closed box
        for i, o in enumerate(sc.objects):
            if o["type"] == B.OBJ_SPHERE and i % 2:
                o["fresnel"] = 0.4
                o["ior"] = 1.4
        sc.add_environment((0, 40, 0), 500, (120, 150, 230))
        r = api.Renderer(sc)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        got, st = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
        _compare(got, want)
        assert st["secondary_rays"] == wst["secondary_rays"] > 0
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(60, 66))
def test_random_scenes_with_textures(api, oracle, seed):
    """The random scenes with texture mapping (type 1), solid textures with random projectors (type 2) and the
    3D function (type 3) on a third of their objects -- planes, spheres, meshes -- some of them mirrors.  Bar: the
    sphere mapping goes through acos, which the device and glibc may round differently in the last bit
    (shader.h:213,222), so at most 0.1 % of the pixels may differ, by one unit; every other texture path is
    integer arithmetic on the same doubles."""
    sc, sh, cam, w, h = _random_case(seed)
    rng = S.SplitMix64(7171 + seed)
    n = B._normalize
    for i, o in enumerate(sc.objects):
        u = rng.uniform()
        if u > 0.35:
            continue
        kind = 1 + int(rng.uniform(0, 3))
        frame = None
        if kind == 2:
            frame = (n((1, rng.uniform(-0.3, 0.3), 0)), n((rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1)),
                     n((rng.uniform(-0.2, 0.2), -1, rng.uniform(-0.3, 0.3))), (rng.uniform(-10, 10), rng.uniform(0, 20), rng.uniform(-5, 8)))
        sx, sy = (rng.uniform(5, 40), rng.uniform(5, 40)) if kind == 2 else (rng.uniform(0.3, 6), rng.uniform(0.3, 6))
        for k in range(2):
            sc.add_texture(i, S._small_texture(8 + 8 * int(rng.uniform(0, 4)), 8 + 8 * int(rng.uniform(0, 3)), seed * 100 + 2 * i + k), sx, sy, frame=frame)
        sc.set_texture_type(i, kind)
        if rng.uniform() < 0.2:
            o["ks"] = 0.5
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
    r = api.Renderer(sc)
    for mode in (B.MODE_PACKET, B.MODE_BRUTE):
        got, st = r.render(cam, sh, w, h, mode=mode)
        _compare(got, want, exact=bool(os.environ.get("RT355_TEST_EXACT")))  # (the sweep can ask for every byte)
        assert st["shadow_rays"] == wst["shadow_rays"] and st["secondary_rays"] == wst["secondary_rays"], mode
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(70, 74))
def test_random_scenes_gather(api, oracle, seed):
    """HEAD's renderer (CalcSubPixel2 with shading3 / shading2, main.cpp:163-209) on the random scenes: 12 or 42
    directions, both shaders, with and without supersampling; the gather rays of the searched modes go through
    the scene table one direction per pass, those of BRUTE through the one-kernel walk."""
    sc, sh, cam, w, h = _random_case(seed)
    level = seed % 2
    shading = (B.SHADING_BLEED, B.SHADING_OCCLUSION)[(seed // 2) % 2]
    alias = (0, 2)[(seed // 4) % 2]
    if alias:
        w, h = w // 2, h // 2
        cam = B.camera(tuple(cam.p_eye), tuple(cam.v_view), (0, 0, 1), 10.0, w, h)
    dirs = B.geodesic(B.icosphere(level))
    want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=seed, shading=shading), threads=0, dirs=dirs)
    r = api.Renderer(sc)
    r.set_directions(dirs)
    for mode in (B.MODE_PACKET, B.MODE_BRUTE):
        got, st = r.render(cam, sh, w, h, alias=alias, seed=seed, shading=shading, mode=mode)
        _compare(got, want)
        assert st["gather_rays"] == wst["gather_rays"] > 0, mode
    r.close()


@pytest.mark.gpu
def test_gather_shaders_vs_golden_and_checker(api, oracle, golden_dir):
    """CalcSubPixel2 with shading3 / shading2 (main.cpp:163-209, shader.h:100-191), HEAD's renderer: frames
    the reference's own shader code produced (fixtures), then more cases against the CPU checker, ray counts
    included.  Both sides take the two random numbers per hit from rt355_rotation."""
    import json, os
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    for c in meta["gather_cases"]:
        sc, sh, cam, w, h = S.small_case(c["case"])
        w, h = c["width"], c["height"]
        cam = S.default_camera(w, h)
        r = api.Renderer(sc)
        r.set_directions(B.geodesic(B.icosphere(c["level"])))
        for mode in (B.MODE_BRUTE, B.MODE_PACKET):  # PACKET: the mesh cases search their gather rays in the scene table
            got, st = r.render(cam, sh, w, h, alias=c["alias"], seed=c["seed"], shading=c["shading"], mode=mode)
            _compare(got, np.load(os.path.join(golden_dir, c["file"])))
            assert st["gather_rays"] > 0 and st["shadow_rays"] == 0
        r.close()
    for name in ("default_2lights", "inside", "mesh", "spheres300", "empty_scene"):
        sc, sh, cam, w, h = S.small_case(name)
        w, h = (w // 2) | 1, (h // 2) | 1
        cam = S.default_camera(w, h)
        r = api.Renderer(sc)
        for level, shading, alias in ((2, B.SHADING_BLEED, 0), (1, B.SHADING_OCCLUSION, 0), (0, B.SHADING_BLEED, 3)):
            dirs = B.geodesic(B.icosphere(level))
            r.set_directions(dirs)
            want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=9, shading=shading), threads=0, dirs=dirs)
            for mode in (B.MODE_BRUTE, B.MODE_PACKET, B.MODE_ALLPAIRS):
                got, st = r.render(cam, sh, w, h, alias=alias, seed=9, shading=shading, mode=mode)
                _compare(got, want)
                assert st["gather_rays"] == wst["gather_rays"], (name, level, shading, alias, mode)
        r.close()


@pytest.mark.gpu
def test_gather_small_scene_needs_no_mode_and_rows_shard(api, oracle):
    """A scene of a few objects renders the gather shaders in every mode (nothing is searched), slab by slab
    as well; no direction set gives black hits, on a searched scene too."""
    sc, sh, cam, w, h = S.small_case("default_2lights")
    dirs = B.geodesic(B.icosphere(1))
    want, _ = oracle.render(sc.pack(), cam, sh, B.opts(w, h, seed=3, shading=B.SHADING_BLEED), threads=0, dirs=dirs)
    r = api.Renderer(sc)
    black, st = r.render(cam, sh, w, h, seed=3, shading=B.SHADING_BLEED)
    assert not black.any() and st["gather_rays"] == 0 and st["primary_hits"] > 0
    r.set_directions(dirs)
    for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
        got, _ = r.render(cam, sh, w, h, seed=3, shading=B.SHADING_BLEED, mode=mode)
        _compare(got, want)
    parts = np.zeros_like(want)
    for y0, y1 in ((0, 11), (11, h)):
        r.render(cam, sh, w, h, row_begin=y0, row_end=y1, out=parts, seed=3, shading=B.SHADING_BLEED)
    assert np.array_equal(parts, want)
    # a scene of at most four planes and four spheres takes the kernel that keeps the objects' origin terms in
    # registers; the general kernel gives the same frame
    for shading in (B.SHADING_BLEED, B.SHADING_OCCLUSION):
        r.set_option("gather_general", 1)
        general, gst = r.render(cam, sh, w, h, seed=3, shading=shading)
        r.set_option("gather_general", 0)
        few, fst = r.render(cam, sh, w, h, seed=3, shading=shading)
        assert np.array_equal(general, few) and gst["gather_rays"] == fst["gather_rays"]
    r.close()
    big, bsh, bcam, bw, bh = S.small_case("spheres300")
    r = api.Renderer(big)
    black, st = r.render(bcam, bsh, bw, bh, shading=B.SHADING_BLEED, mode=B.MODE_PACKET)
    assert not black.any() and st["gather_rays"] == 0 and st["primary_hits"] > 0
    r.close()


@pytest.mark.gpu
def test_device_sincos_equals_host_evaluation(api):
    """The rotation's sine and cosine on the device are the doubles rt355_sincos returns on the host (which
    tests/test_host_math.py checks against exact arithmetic)."""
    import math
    rng = np.random.default_rng(357)
    xs = np.concatenate([rng.random(20000) * 2 * 3.1415927, [k * math.pi / 2 for k in range(5)],
                         [math.nextafter(k * math.pi / 2, 9.0) for k in range(5)], [0.0, 1e-300, 1e-9, 2 * 3.1415927]])
    dev = api.probe_sincos(xs)
    host = np.array([api.sincos(float(x)) for x in xs])
    assert np.array_equal(dev, host)

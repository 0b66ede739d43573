"""GPU: the benchmark's own workloads at their own size against the CPU checker.

The throughput scenes (10 000 spheres under 8 lights at 3840x2160; the 100 000-triangle and the 1 000 000-triangle
textured mesh under 2 lights at 3840x2160 and 7680x4320; the 60-camera sweep of the triangle scene) are
far too large for the checker to render whole -- it tests every primitive for every ray, as the reference does
(main.cpp:65-79, shader.h:393-433, anyobject.h:646-666) -- so each frame is checked on a strided sample of
its pixels (every 16th, 32nd or 64th in x and y: the same rays as the full frame's), byte for byte, and
through properties that do not depend on the size: the packet search and the all-pairs search give the same
bytes and the same candidate counts, row slabs and interleaved row tiles reproduce the frame rendered at once,
and rt355_render (host pixmap, read back in parts) equals rt355_render_device (framebuffer in HBM).
"""
import numpy as np
import pytest

import bench
from raytracer_amd import binding as B
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    from raytracer_amd import api as A
    A.lib()
    return A


def _sample_equal(frame, oracle, packed, cam, sh, w, h, stride):
    want, wst = oracle.render(packed, cam, sh, B.opts(w, h), threads=0, stride=stride)
    a, b = frame[::stride, ::stride], want[::stride, ::stride]
    ndiff = int((a != b).any(axis=2).sum())
    assert ndiff == 0, f"{ndiff} of {a.shape[0] * a.shape[1]} sampled pixels differ from the CPU checker"
    return wst


def _device_frame(api, r, cam, sh, o, w, h):
    import torch
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    r.render_device(cam, sh, o, fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = r.sync()
    return fb.cpu().numpy(), st


@pytest.mark.parametrize("name,stride", [("spheres10k_8lights_4k", 16), ("mesh100k_2lights_4k", 16)])
def test_throughput_scene_at_full_size(api, oracle, name, stride):
    """The 10 000-sphere and the 100 000-triangle benchmark frames (one GPU, 3840x2160): every 16th pixel in x and y against the checker, then the
    whole frame against itself through the other search, the two partitions and the other entry point."""
    scene, sh, cam, w, h = bench.build_workload(name)
    packed = scene.pack()
    r = api.Renderer(packed)
    frame, st = r.render(cam, sh, w, h, mode=B.MODE_PACKET)
    assert st["primary_rays"] == w * h
    _sample_equal(frame, oracle, packed, cam, sh, w, h, stride)
    # the all-pairs search: same bytes, same candidates
    allp, ast = r.render(cam, sh, w, h, mode=B.MODE_ALLPAIRS)
    assert np.array_equal(allp, frame)
    assert ast["exact_tests"] == st["exact_tests"] and ast["shadow_rays"] == st["shadow_rays"]
    # two row slabs, then eight shares of interleaved 32-row tiles (the 8-GPU partition), into one pixmap each
    parts = np.zeros_like(frame)
    shadow = 0
    for y0, y1 in ((0, 1000), (1000, h)):
        _, pst = r.render(cam, sh, w, h, row_begin=y0, row_end=y1, out=parts)
        shadow += pst["shadow_rays"]
    assert np.array_equal(parts, frame) and shadow == st["shadow_rays"]
    tiles = np.zeros_like(frame)
    shadow = 0
    for g in range(8):
        _, pst = r.render(cam, sh, w, h, out=tiles, tile_rows=32, tile_count=8, tile_phase=g)
        shadow += pst["shadow_rays"]
    assert np.array_equal(tiles, frame) and shadow == st["shadow_rays"]
    # the framebuffer left in HBM, and a page-locked pixmap
    dev, dst = _device_frame(api, r, cam, sh, B.opts(w, h), w, h)
    assert np.array_equal(dev, frame) and dst["shadow_rays"] == st["shadow_rays"]
    host = api.HostPixmap(w, h)
    r.render(cam, sh, w, h, out=host.array)
    assert np.array_equal(host.array, frame)
    host.close()
    r.close()


def test_million_triangles_at_8k(api, oracle):
    """The largest benchmark frame at its full size: 1 000 000 triangles at 7680x4320.  Every 64th pixel in x and y
    against the checker (8 160 camera rays and their shadow rays against a million triangles each), eight shares
    of interleaved row tiles against the frame rendered at once."""
    scene, sh, cam, w, h = bench.build_workload("mesh1m_8k_tiles")
    packed = scene.pack()
    assert scene.n_primitives >= 999_000  # (a square height field: 2 x 707 x 707 triangles and the floor)
    r = api.Renderer(packed)
    frame, st = r.render(cam, sh, w, h)
    assert st["primary_rays"] == w * h
    _sample_equal(frame, oracle, packed, cam, sh, w, h, 64)
    tiles = np.zeros_like(frame)
    shadow = 0
    for g in range(8):
        _, pst = r.render(cam, sh, w, h, out=tiles, tile_rows=32, tile_count=8, tile_phase=g)
        shadow += pst["shadow_rays"]
    assert np.array_equal(tiles, frame) and shadow == st["shadow_rays"]
    r.close()


def test_camera_sweep_of_the_triangle_scene(api, oracle):
    """The camera-sweep benchmark: frames of the 60-camera sweep of the 100 000-triangle scene at 3840x2160, one scene
    handle rendering them one after the other (the eye's tables are rebuilt for every frame, the lights' are
    kept): every 32nd pixel of five of them against the checker, and the round-robin deal covers every frame."""
    from raytracer_amd.sharding import frames_for_rank
    scene, sh, _, w, h = bench.build_workload("mesh100k_sweep60")
    cams = bench.sweep_of("mesh100k_sweep60", w, h)
    assert len(cams) == 60
    assert sorted(f for g in range(8) for f in frames_for_rank(60, g, 8)) == list(range(60))
    packed = scene.pack()
    r = api.Renderer(packed)
    host = api.HostPixmap(w, h)
    mine = frames_for_rank(60, 3, 8)[:5]  # what rank 3 of 8 renders: 3, 11, 19, 27, 35
    frames, rays = {}, {}
    for f in mine:
        _, st = r.render(cams[f], sh, w, h, out=host.array)
        assert st["primary_hits"] > w * h // 4
        _sample_equal(host.array, oracle, packed, cams[f], sh, w, h, 32)
        frames[f], rays[f] = host.array.copy(), st["shadow_rays"]
    # the same frames, two in flight: two scene handles, each on a stream of its own, render straight into their
    # page-locked pixmaps (rt355_render_device with the pixmap's device address) while the other one is still busy
    import torch
    fly = [(r, host, torch.cuda.Stream()), (api.Renderer(packed), api.HostPixmap(w, h), torch.cuda.Stream())]
    pending = [None, None]

    def finish(k):
        st = fly[k][0].sync()
        f = pending[k]
        assert np.array_equal(fly[k][1].array, frames[f]) and st["shadow_rays"] == rays[f], f
        pending[k] = None
    for i, f in enumerate(mine):
        k = i % 2
        if pending[k] is not None:
            finish(k)
        fly[k][1].array[:] = 0
        fly[k][0].render_device(cams[f], sh, B.opts(w, h), fly[k][1].device_ptr, fly[k][2].cuda_stream)
        pending[k] = f
    for k in (0, 1):
        if pending[k] is not None:
            finish(k)
    fly[1][0].close()
    fly[1][1].close()
    # the same frames by rt355_render_sweep: one call, the readback of a frame running while the next is rendered
    pix = [api.HostPixmap(w, h) for _ in mine]
    sts = r.render_sweep([cams[f] for f in mine], sh, B.opts(w, h), [p.array for p in pix])
    for f, p, st in zip(mine, pix, sts):
        assert np.array_equal(p.array, frames[f]) and st["shadow_rays"] == rays[f], f
        p.close()
    host.close()
    r.close()


def test_sweep_entry_point(api):
    """rt355_render_sweep against rt355_render frame by frame: page-locked and pageable pixmaps (and both in one
    call), a searched scene and an unsearched one, whole frames, a row slab and a share of interleaved tiles (rows
    outside the call's share stay untouched), zero frames and one."""
    for case, (w, h) in (("spheres300", (200, 150)), ("default_2lights", (161, 97))):
        sc, sh, _, _, _ = S.small_case(case)
        r = api.Renderer(sc)
        cams = [S.orbit_camera(f, w, h, n_frames=7, center=(0.0, 40.0, 0.0), radius=60.0) for f in range(7)]
        assert r.render_sweep([], sh, B.opts(w, h), []) == []
        for kw in ({}, {"row_begin": 40, "row_end": 90}, {"tile_rows": 16, "tile_count": 3, "tile_phase": 1}):
            want = []
            for c in cams:
                out = np.full((h, w, 3), 7, dtype=np.uint8)
                _, st = r.render(c, sh, w, h, out=out, **kw)
                want.append((out, st))
            for n, pinned in ((7, "all"), (7, "none"), (7, "mixed"), (1, "all")):
                hosts = [api.HostPixmap(w, h) for _ in range(n)]
                pix = []
                for i, hp in enumerate(hosts):
                    a = hp.array if (pinned == "all" or (pinned == "mixed" and i % 3 != 1)) else np.empty((h, w, 3), dtype=np.uint8)
                    a[:] = 7
                    pix.append(a)
                sts = r.render_sweep(cams[:n], sh, B.opts(w, h, **kw), pix)
                for i in range(n):
                    assert np.array_equal(pix[i], want[i][0]), (case, kw, pinned, i)
                    for k in ("primary_rays", "shadow_rays", "primary_hits", "exact_tests"):
                        assert sts[i][k] == want[i][1][k], (case, kw, pinned, i, k)
                for hp in hosts:
                    hp.close()
        r.close()


def test_search_in_two_launches(api, oracle):
    """The packet search of a large scene runs as two launches: a wave that has met enough primitives hands the rest
    of its walk to many waves, which append to its rays' lists through the rays' counts in memory.  Forced here on
    small scenes, with waves handing over after 4 and after 64 primitives (and a list of candidates that overflows
    into the spill chains): frames, ray counts and the candidates evaluated equal the one-launch search's, the
    all-pairs search's and the checker's."""
    cases = [("spheres300", {}), ("mesh", {}), ("mirrors_mesh", {})]
    sc_big = S.mesh_scene(6000, seed=7, tex_size=64)
    for name, _ in cases + [("mesh6k", {})]:
        if name == "mesh6k":
            sc, sh, (w, h) = sc_big, S.mesh_shader(), (224, 128)
            cam = S.default_camera(w, h)
        else:
            sc, sh, cam, w, h = S.small_case(name)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h), threads=0)
        r = api.Renderer(sc)
        one, ost = r.render(cam, sh, w, h)
        allp, ast = r.render(cam, sh, w, h, mode=B.MODE_ALLPAIRS)
        assert np.array_equal(one, want) and np.array_equal(allp, want)
        for after in ("4", "64"):
            r.set_option("heavy_min_prims", 0)
            r.set_option("heavy_after", int(after))
            two, tst = r.render(cam, sh, w, h)
            r.set_option("heavy_min_prims", 300000)
            r.set_option("heavy_after", 256)
            assert np.array_equal(two, want), (name, after)
            for k in ("shadow_rays", "exact_tests", "overflow_rays", "secondary_rays"):
                assert tst[k] == ost[k] == ast[k], (name, after, k)
            assert tst["shadow_rays"] == wst["shadow_rays"]
        r.close()


def test_searched_gather_without_lights(api, oracle):
    """The gather shaders use no lights (main.cpp:205: shading3 takes none), so an empty light list is a natural
    call; on a searched scene the frame then records its two search launches with no light group at all."""
    sc, _, cam, w, h = S.small_case("spheres300")
    sh = S.Shader([], 0x101010, 0xDDDDDD, 0x000000)
    dirs = B.geodesic(B.icosphere(0))
    r = api.Renderer(sc)
    r.set_directions(dirs)
    for shading in (B.SHADING_BLEED, B.SHADING_OCCLUSION):
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, seed=4, shading=shading), threads=0, dirs=dirs)
        for mode in (B.MODE_PACKET, B.MODE_ALLPAIRS, B.MODE_BRUTE):
            got, st = r.render(cam, sh, w, h, seed=4, shading=shading, mode=mode)
            assert np.array_equal(got, want), (shading, mode)
            assert st["gather_rays"] == wst["gather_rays"] > 0
    r.close()


def test_searched_gather_at_full_size(api, oracle):
    """HEAD's renderer (CalcSubPixel2 + shading3 / shading2) on the 100 000-triangle scene at 3840x2160, 42
    directions: 87 million gather rays through the scene table in one launch.  Every 64th pixel in x and y against
    the checker (which tests every triangle for every ray), ray counts of the sample's rows aside; the two shaders;
    eight shares of interleaved row tiles against the frame rendered at once."""
    scene, sh, cam, w, h = bench.build_workload("mesh100k_bleed42_4k")
    shading, dirs = bench.gather_of("mesh100k_bleed42_4k")
    packed = scene.pack()
    r = api.Renderer(packed)
    r.set_directions(dirs)
    stride = 64
    for shade in (shading, B.SHADING_OCCLUSION):
        frame, st = r.render(cam, sh, w, h, shading=shade)
        assert st["primary_rays"] == w * h and st["gather_rays"] > 80_000_000 and st["shadow_rays"] == 0
        want, wst = oracle.render(packed, cam, sh, B.opts(w, h, shading=shade), threads=0, stride=stride, dirs=dirs)
        a, b = frame[::stride, ::stride], want[::stride, ::stride]
        ndiff = int((a != b).any(axis=2).sum())
        assert ndiff == 0, f"{ndiff} of {a.shape[0] * a.shape[1]} sampled pixels differ from the CPU checker"
    tiles = np.zeros_like(frame)
    gathered = 0
    for g in range(8):
        _, pst = r.render(cam, sh, w, h, shading=B.SHADING_OCCLUSION, out=tiles, tile_rows=32, tile_count=8, tile_phase=g)
        gathered += pst["gather_rays"]
    assert np.array_equal(tiles, frame) and gathered == st["gather_rays"]
    r.close()


def test_interleaved_row_tiles_small(api, oracle):
    """rt355_opts.tile_*: the shares of the interleaved row tiles reproduce the frame, whatever the tile height,
    the number of shares (more shares than tiles included) and the renderer: lights, supersampling, mirrors
    rendered level by level, the gather shaders searched and unsearched; inside a row slab too."""
    cases = [("spheres300", dict()), ("mesh", dict(alias=2, seed=9)), ("mirrors_mesh", dict()),
             ("default_2lights", dict(shading=B.SHADING_BLEED, seed=5)), ("spheres300", dict(shading=B.SHADING_OCCLUSION, seed=6)),
             ("glass", dict())]
    dirs = B.geodesic(B.icosphere(0))
    for name, kw in cases:
        sc, sh, cam, w, h = S.small_case(name)
        r = api.Renderer(sc)
        r.set_directions(dirs)
        want, wst = oracle.render(sc.pack(), cam, sh, B.opts(w, h, **kw), threads=0, dirs=dirs)
        for tile_rows, count in ((32, 2), (7, 3), (16, 8), (64, 5)):
            got = np.zeros_like(want)
            rays = 0
            for g in range(count):
                _, st = r.render(cam, sh, w, h, out=got, tile_rows=tile_rows, tile_count=count, tile_phase=g, **kw)
                rays += st["primary_rays"]
            assert np.array_equal(got, want), (name, tile_rows, count)
            assert rays == wst["primary_rays"]
        # tiles of a slab: rows 10..h-5 only
        got = np.zeros_like(want)
        for g in range(3):
            r.render(cam, sh, w, h, out=got, row_begin=10, row_end=h - 5, tile_rows=9, tile_count=3, tile_phase=g, **kw)
        assert np.array_equal(got[10:h - 5], want[10:h - 5]) and not got[:10].any() and not got[h - 5:].any()
        with pytest.raises(api.Rt355Error, match="tile"):
            r.render(cam, sh, w, h, tile_rows=0, tile_count=2, tile_phase=0)
        with pytest.raises(api.Rt355Error, match="tile"):
            r.render(cam, sh, w, h, tile_rows=8, tile_count=2, tile_phase=2)
        r.close()


def test_host_pixmap_paths_agree(api):
    """rt355_render into a pageable pixmap (rendered in HBM, read back through staging), into a page-locked one
    (written by the kernels themselves, or read back by copy), in one part and in several, a searched scene and
    an unsearched one, and rt355_render_device: the same bytes and ray counts; a second, smaller frame on the
    same handle reuses the framebuffer."""
    import os
    for case in ("spheres300", "default_2lights", "mirror_spheres"):
        if case == "mirror_spheres":  # searched, with mirror levels: the last stage of the frame holds a fold
            sc = S.spheres_scene(200, seed=94)
            for i, o in enumerate(sc.objects):
                if o["type"] == B.OBJ_PLANE or i % 3 == 0:
                    o["ks"] = 0.5
            sh = S.Shader(S.ring_lights(2), 0x101010, 0xDDDDDD, 0x202020)
        else:
            sc, sh, cam, _, _ = S.small_case(case)
        r = api.Renderer(sc)
        for w, h in ((640, 1100), (322, 300)):
            cam = S.default_camera(w, h)
            dev, dst = _device_frame(api, r, cam, sh, B.opts(w, h), w, h)
            host = api.HostPixmap(w, h)
            _, hst = r.render(cam, sh, w, h, out=host.array)   # the kernels write into the mapped pixmap
            assert np.array_equal(host.array, dev)
            for parts in ("1", "3", "8"):
                os.environ["RT355_RENDER_PARTS"] = parts        # parts of a one-kernel frame
                os.environ["RT355_TAIL_PARTS"] = parts          # parts of a searched frame's last stage
                os.environ["RT355_READBACK_COPY"] = "1"         # ... or into HBM, and the copy engine reads back
                try:
                    pageable, pst = r.render(cam, sh, w, h)
                    host.array[:] = 0
                    _, cst = r.render(cam, sh, w, h, out=host.array)
                finally:
                    del os.environ["RT355_RENDER_PARTS"], os.environ["RT355_TAIL_PARTS"], os.environ["RT355_READBACK_COPY"]
                assert np.array_equal(pageable, dev) and np.array_equal(host.array, dev), parts
                for k in ("primary_rays", "shadow_rays", "primary_hits", "exact_tests"):
                    assert pst[k] == hst[k] == cst[k] == dst[k], (parts, k)
            host.close()
        r.close()

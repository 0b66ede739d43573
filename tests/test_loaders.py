"""Host file formats (CPU): the OBJ reader of Mesh::Mesh (anyobject.h:546-644), the BMP reader of
Texture::Texture (texture.h:42-79) and the direction reader Geodesic (main.cpp:408-450), as the Python host
(raytracer_amd/binding.py) and the C++ host (raytracer_amd/host/rt355_host.hpp, through rt355_dump) implement
them, against what the reference's own readers made of the same files (tests/golden/loaders, written by
oracle/make_golden.py) and, where the harness is built, against those readers directly."""
import json
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import binding as B
from raytracer_amd import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "raytracer_amd", "host", "rt355_dump")


@pytest.fixture(scope="module")
def dump():
    if not os.path.exists(DUMP):
        import __graft_entry__ as g
        g.build()
    return DUMP


def _cpp_obj(dump, path, origin, scale, tmp_path):
    out = str(tmp_path / "tris.f64")
    subprocess.run([dump, "obj", str(path), repr(origin[0]), repr(origin[1]), repr(origin[2]), repr(scale), out],
                   check=True, timeout=60)
    return np.fromfile(out, dtype=np.float64).reshape(-1, 15)


def test_obj_reader_matches_the_reference_fixtures(dump, golden_dir, tmp_path):
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    assert meta.get("loader_cases"), "golden set predates the loader fixtures: rerun oracle/make_golden.py"
    for c in meta["loader_cases"]:
        path = os.path.join(golden_dir, "loaders", c["obj"])
        want = np.load(os.path.join(golden_dir, "loaders", c["file"]))
        assert len(want) == c["triangles"]
        assert np.array_equal(B.read_obj(path, c["origin"], c["scale"]), want), c
        assert np.array_equal(_cpp_obj(dump, path, c["origin"], c["scale"], tmp_path), want), c


def test_missing_files_behave_as_in_the_reference(dump, tmp_path):
    """A missing OBJ gives an empty mesh (anyobject.h:551-554), a missing texture is an error (texture.h:50-53)."""
    assert B.read_obj(str(tmp_path / "nope.obj")).shape == (0, 15)
    assert _cpp_obj(dump, tmp_path / "nope.obj", (0.0, 0.0, 0.0), 1.0, tmp_path).shape == (0, 15)
    with pytest.raises(OSError):
        B.read_bmp(str(tmp_path / "nope.bmp"))
    r = subprocess.run([dump, "bmp", str(tmp_path / "nope.bmp"), str(tmp_path / "o.u8")], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"cannot open" in r.stderr
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        B.read_obj(str(bad))
    r = subprocess.run([dump, "obj", str(bad), "0", "0", "0", "1", str(tmp_path / "o.f64")], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"not read" in r.stderr


def test_bmp_reader_matches_the_reference_fixture(dump, golden_dir, tmp_path):
    path = os.path.join(golden_dir, "loaders", "tex16x8.bmp")
    want = np.load(os.path.join(golden_dir, "loaders", "tex16x8.npy"))
    assert np.array_equal(B.read_bmp(path), want)
    out = str(tmp_path / "tex.u8")
    subprocess.run([dump, "bmp", path, out], check=True, timeout=60)
    raw = np.fromfile(out, dtype=np.uint8)
    w, h = np.frombuffer(raw[:8].tobytes(), dtype=np.int32)
    assert (w, h) == (16, 8) and np.array_equal(raw[8:].reshape(h, w, 3), want)
    # the writer is the reader's inverse, odd widths included (the reference's reader assumes unpadded rows)
    odd = S._small_texture(5, 3, 11)
    B.write_bmp(str(tmp_path / "odd.bmp"), odd)
    assert np.array_equal(B.read_bmp(str(tmp_path / "odd.bmp")), odd)


def test_geodesic_reader_cpp_equals_python(dump, tmp_path):
    verts = [(1.5 * x + 0.25, 1.5 * y - 0.5, 1.5 * z + 1.0) for x, y, z in B.icosphere(1)]
    obj = tmp_path / "geodesic.obj"
    obj.write_text("# directions\n" + "".join(f"v {x!r} {y!r} {z!r}\n" for x, y, z in verts) + "vt 0 0\nf 1 2 3\n")
    out = str(tmp_path / "geo.f64")
    subprocess.run([dump, "geo", str(obj), out], check=True, timeout=60)
    got = np.fromfile(out, dtype=np.float64).reshape(-1, 3)
    assert got.shape == (42, 3) and np.array_equal(got, B.read_geodesic(str(obj))) and np.array_equal(got, B.geodesic(verts))


def test_readers_against_the_reference_readers_on_generated_files(dump, tmp_path):
    """Random meshes written in each of the three face formats, with comments, quads and blank lines, read by
    Mesh::Mesh itself through the harness (this container only) and by both hosts."""
    from oracle.oracle_py import Reference
    if not Reference.available():
        pytest.skip("reference harness not built (it needs /root/reference)")
    R = Reference()
    rng = S.SplitMix64(4242)
    for fmt in (1, 2, 3):
        nv = 40
        v = rng.array(nv * 3).reshape(nv, 3) * 20 - 10
        vt = rng.array(nv * 2).reshape(nv, 2)
        lines = ["# generated", ""]
        lines += [f"v {a!r} {b!r} {c!r}" for a, b, c in v.tolist()]
        if fmt >= 2:
            lines += [f"vt {a!r} {b!r}" for a, b in vt.tolist()]
        if fmt == 3:
            lines += ["vn 0 0 1", "vn 1 0 0"]
        for k in range(60):
            idx = [1 + int(rng.uniform() * nv) for _ in range(4)]
            corner = {1: lambda i: f"{i}", 2: lambda i: f"{i}/{1 + (i * 7) % nv}", 3: lambda i: f"{i}/{1 + (i * 7) % nv}/{1 + i % 2}"}[fmt]
            lines.append("f " + " ".join(corner(i) for i in (idx if k % 5 == 0 else idx[:3])))
            if k % 17 == 0:
                lines.append("g part")
        path = tmp_path / f"gen{fmt}.obj"
        path.write_text("\n".join(lines) + "\n")
        for origin, scale in (((0.0, 0.0, 0.0), 1.0), ((3.25, -1.0, 0.5), 0.75)):
            want = R.load_mesh(str(path), origin, scale)
            assert len(want) == 60
            assert np.array_equal(B.read_obj(str(path), origin, scale), want)
            assert np.array_equal(_cpp_obj(dump, path, origin, scale, tmp_path), want)
    for w, h in ((32, 16), (4, 64)):
        tex = S._small_texture(w, h, 100 + w)
        p = tmp_path / f"t{w}.bmp"
        B.write_bmp(str(p), tex)
        assert np.array_equal(R.load_texture(str(p)), tex) and np.array_equal(B.read_bmp(str(p)), tex)


def test_loaded_mesh_renders_like_the_same_triangles_given_directly(oracle, golden_dir):
    """An OBJ file read by the host goes through Scene.add_mesh like any triangle list."""
    tris = B.read_obj(os.path.join(golden_dir, "loaders", "uvn.obj"), (-1.0, 2.0, -1.0), 3.0)
    assert tris.shape == (2, 15)
    sc = B.Scene()
    sc.add_plane((0, 13, 0), (0, -1, 0), (-1, 0, 0), (200, 200, 200))
    m = sc.add_mesh(tris, clr=(150, 200, 110))
    sc.add_texture(m, B.read_bmp(os.path.join(golden_dir, "loaders", "tex16x8.bmp")))
    sc.add_texture(m, B.read_bmp(os.path.join(golden_dir, "loaders", "tex16x8.bmp"))[::-1].copy())
    sc.set_texture_type(m, 1)
    frame, st = oracle.render(sc.pack(), S.default_camera(64, 48), S.default_shader(2), B.opts(64, 48), threads=0)
    assert st["primary_rays"] == 64 * 48 and frame.any()

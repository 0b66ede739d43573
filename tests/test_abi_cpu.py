"""CPU: the C-ABI library loads and exports every symbol include/rt355.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt355.h")).read()
    names = set(re.findall(r"\b(rt355_[a-z_0-9]+)\s*\(", hdr))
    assert {"rt355_scene_create", "rt355_render", "rt355_render_device", "rt355_scene_destroy"} <= names
    so = os.path.join(ROOT, "raytracer_amd", "lib", "librt355.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    from raytracer_amd import api
    assert os.path.samefile(api._LIB_PATH, so) or os.environ.get("RT355_LIB")
    L = api.lib()  # loads PyTorch's HIP runtime first when PyTorch is present (one runtime per process)
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in rt355.h but not exported"
    L.rt355_abi_version.restype = ctypes.c_int
    assert L.rt355_abi_version() >= 5
    # the exported jitter and its Python mirror
    from raytracer_amd import binding as B
    L.rt355_jitter.restype = ctypes.c_double
    L.rt355_jitter.argtypes = [ctypes.c_uint32] + [ctypes.c_int32] * 4
    for a in ((0, 0, 0, 0, 0), (5, 17, 3, 8, 1), (4_000_000_000, 3839, 2159, 15, 0)):
        assert L.rt355_jitter(*a) == B.jitter(*a)
    L.rt355_rotation.restype = ctypes.c_double
    L.rt355_rotation.argtypes = [ctypes.c_uint32] + [ctypes.c_int32] * 4
    for a in ((0, 0, 0, 0, 0), (5, 17, 3, 8, 1), (4_000_000_000, 3839, 2159, 15, 0)):
        assert L.rt355_rotation(*a) == B.rotation(*a)
    L.rt355_strata.restype = ctypes.c_double
    L.rt355_strata.argtypes = [ctypes.c_uint32] + [ctypes.c_int32] * 3
    for a in ((0, 0, 0, 0), (5, 17, 3, 2), (4_000_000_000, 3839, 2159, 1)):
        assert L.rt355_strata(*a) == B.strata(*a)


def test_struct_sizes_match_header_layout():
    from raytracer_amd import binding as B
    assert B.OBJECT_DTYPE.itemsize == 232 and B.LIGHT_DTYPE.itemsize == 144
    assert ctypes.sizeof(B.CCamera) == 96 and ctypes.sizeof(B.CStats) == 80
    # ... and what the C compiler makes of the header itself
    import subprocess, tempfile
    prog = ('#include <stdio.h>\n#include <stddef.h>\n#include "rt355.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
            'sizeof(rt355_object), offsetof(rt355_object, s), offsetof(rt355_object, a), sizeof(rt355_texture), '
            'offsetof(rt355_texture, n0), sizeof(rt355_light), sizeof(rt355_opts), sizeof(rt355_stats)); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sizes.c")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "sizes"), src])
        got = [int(v) for v in subprocess.check_output([os.path.join(d, "sizes")]).split()]
    assert got == [B.OBJECT_DTYPE.itemsize, B.OBJECT_DTYPE.fields["s"][1], B.OBJECT_DTYPE.fields["a"][1],
                   ctypes.sizeof(B.CTexture), B.CTexture.n0.offset, B.LIGHT_DTYPE.itemsize,
                   ctypes.sizeof(B.COpts), ctypes.sizeof(B.CStats)]


def test_null_arguments_are_rejected_without_a_gpu():
    from raytracer_amd import api
    L = api.lib()
    out = ctypes.c_void_p()
    assert L.rt355_scene_create(None, ctypes.byref(out)) == -1
    assert b"null" in L.rt355_last_error()
    # the entry points added for one process on several GPUs and for the handle's options check their arguments
    # before they touch the runtime
    devs = (ctypes.c_int32 * 2)(0, 1)
    assert L.rt355_scene_create_multi(None, devs, 2, ctypes.byref(out)) == -1
    assert L.rt355_scene_set_option(None, b"readback", 1) == -1 and b"null" in L.rt355_last_error()
    assert L.rt355_scene_sync(None) == -1
    assert L.rt355_render(None, None, None, None, None, None) == -1
    assert L.rt355_render_sweep(None, None, 0, None, None, None, None) == -1

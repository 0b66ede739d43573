"""Generate tests/golden/* from the reference's own code (oracle/_ref/libref_harness.so).

Run in the container that has /root/reference:  python -m oracle.make_golden
The outputs are data only: inputs (seeded, re-creatable from raytracer_amd.scenes) and the
values the reference computed for them.
"""
from __future__ import annotations

import json
import os

import numpy as np

from raytracer_amd import binding as B
from raytracer_amd import scenes as S
from oracle.oracle_py import Reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def random_rays(n, seed, origin_box=((-40, -40, -20), (40, 40, 40))):
    rng = S.SplitMix64(seed)
    u = rng.array(n * 6).reshape(n, 6)
    lo, hi = np.array(origin_box[0]), np.array(origin_box[1])
    pe = lo + (hi - lo) * u[:, :3]
    d = u[:, 3:] * 2 - 1
    return np.concatenate([pe, d], axis=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = Reference()
    meta = {}
    # 1. golden frames
    for name in S.SMALL_CASES:
        sc, sh, cam, w, h = S.small_case(name)
        frame = R.render(sc.pack(), cam, sh, B.opts(w, h))
        np.save(os.path.join(OUT, f"frame_{name}.npy"), frame)
        meta[name] = {"width": w, "height": h, "fnv1a64": f"{B.fnv1a64(frame):016x}"}
    # 1b. supersampled frames (alias > 0): CalcPixel's sub-sample loop run with the reference's classes, the
    #     jitter taken from rt355_jitter in place of rand()
    meta["alias_cases"] = []
    for name, alias in (("default_2lights", 2), ("textured", 3), ("mesh", 2), ("spheres300_dir", 3),
                        ("environment", 2), ("quadrics", 2), ("glass", 2)):
        sc, sh, cam, w, h = S.small_case(name)
        w, h = w // 2, h // 2
        cam = S.default_camera(w, h)
        seed = 4000 + alias
        frame = R.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=seed))
        fn = f"frame_{name}_alias{alias}.npy"
        np.save(os.path.join(OUT, fn), frame)
        meta["alias_cases"].append({"case": name, "alias": alias, "seed": seed, "width": w, "height": h, "file": fn,
                                    "fnv1a64": f"{B.fnv1a64(frame):016x}"})
    # 1c. the gather shaders (CalcSubPixel2 with the reference's own Shader::shading3 / shading2), the two
    #     rand() draws per hit replaced by rt355_rotation; direction sets are subdivided icosahedra
    meta["gather_cases"] = []
    for name, level, shading, alias in (("default_2lights", 0, B.SHADING_BLEED, 0), ("default_2lights", 2, B.SHADING_BLEED, 0),
                                        ("default_2lights", 1, B.SHADING_OCCLUSION, 0), ("mesh", 1, B.SHADING_BLEED, 2),
                                        ("inside", 1, B.SHADING_OCCLUSION, 2), ("textured", 1, B.SHADING_BLEED, 0),
                                        ("mirrors_mesh", 0, B.SHADING_BLEED, 0), ("environment", 1, B.SHADING_BLEED, 0),
                                        ("environment_spheres", 0, B.SHADING_OCCLUSION, 0),
                                        ("environment_spheres", 0, B.SHADING_BLEED, 2),
                                        ("quadrics", 1, B.SHADING_BLEED, 0), ("quadrics_spheres", 0, B.SHADING_OCCLUSION, 2)):
        sc, sh, cam, w, h = S.small_case(name)
        w, h = w // 2, h // 2
        cam = S.default_camera(w, h)
        seed = 5000 + 10 * level + shading
        dirs = B.geodesic(B.icosphere(level))
        frame = R.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=seed, shading=shading), dirs=dirs)
        fn = f"frame_{name}_gather{shading}_l{level}_a{alias}.npy"
        np.save(os.path.join(OUT, fn), frame)
        meta["gather_cases"].append({"case": name, "level": level, "shading": shading, "alias": alias, "seed": seed,
                                     "width": w, "height": h, "file": fn, "fnv1a64": f"{B.fnv1a64(frame):016x}"})
    # 1e. soft shadows: CalcSubPixel with Shader::shading's loop run with the reference's own CalcSoftShadow
    #     (shader.h:439-483, the call commented out at :91) in place of CalcShadow; the strata of main.cpp:286-288
    #     from rt355_strata in place of rand()
    meta["soft_cases"] = []
    for name, kind, alias in (("default_2lights", "area", 0), ("default_2lights", "area", 2), ("default_2lights", "point", 3),
                              ("mesh", "area", 2), ("mirrors", "area", 2), ("spheres300", "area", 2), ("glass", "area", 2),
                              ("textured", "point", 0), ("environment", "area", 1), ("mirrors_mesh", "area", 3)):
        sc, _, cam, w, h = S.small_case(name)
        sh = S.soft_shader(kind)
        w, h = w // 2, h // 2
        cam = S.default_camera(w, h)
        seed = 6000 + alias
        frame = R.render(sc.pack(), cam, sh, B.opts(w, h, alias=alias, seed=seed, shading=B.SHADING_SOFT))
        fn = f"frame_{name}_soft_{kind}_a{alias}.npy"
        np.save(os.path.join(OUT, fn), frame)
        meta["soft_cases"].append({"case": name, "lights": kind, "alias": alias, "seed": seed, "width": w, "height": h,
                                   "file": fn, "fnv1a64": f"{B.fnv1a64(frame):016x}"})
    # 1d. host file formats: small OBJ and BMP files and what the reference's own readers (Mesh::Mesh,
    #     Texture::Texture) make of them
    ldir = os.path.join(OUT, "loaders")
    os.makedirs(ldir, exist_ok=True)
    objs = {
        "plain.obj": "# three numbers per face\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\n\nf 1 2 3\nf 2 4 3\nf 1 2 3 4\n",
        "uv.obj": "v -1.5 0.25 2\nv 1e0 0 0\nv 0 1 0\nv 1 1 .5\nvt 0 0\nvt 1 0\nvt 0.25 0.75\nvt 1 1\n"
                  "f 1/1 2/2 3/3\nf 2/2 4/4 3/3\n",
        "uvn.obj": "o thing\nv 0 0 0\nv 2 0 0\nv 0 2 0\nv 2 2 1\nvn 0 0 1\nvn 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\n"
                   "s off\nf 1/1/1 2/2/1 3/3/2\nf 2/2/1 4/4/2 3/3/1 1/1/1\n",
        "escalate.obj": "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0.5 0.5\nvt 1 0\nvt 0 1\nvn 0 0 1\n"
                        "f 1 2 3\nf 2/2 4/1 3/3\nf 1/1/1 2/2/1 4/3/1\nf 1/2 2/3 3/1\nf 1 2 3\n",
        "broken.obj": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nf 1//1 2//1 3//1\nf 1 2 3\n",
    }
    meta["loader_cases"] = []
    for name, text in objs.items():
        path = os.path.join(ldir, name)
        with open(path, "w") as f:
            f.write(text)
        for origin, scale in (((0.0, 0.0, 0.0), 1.0), ((1.5, -2.0, 0.125), 2.5)):
            tris = R.load_mesh(path, origin, scale)
            fn = f"{name[:-4]}_s{scale}.npy"
            np.save(os.path.join(ldir, fn), tris)
            meta["loader_cases"].append({"obj": name, "origin": list(origin), "scale": scale, "file": fn, "triangles": len(tris)})
    tex = S._small_texture(16, 8, 907)
    B.write_bmp(os.path.join(ldir, "tex16x8.bmp"), tex)
    got = R.load_texture(os.path.join(ldir, "tex16x8.bmp"))
    assert np.array_equal(got, tex)
    np.save(os.path.join(ldir, "tex16x8.npy"), got)
    # 2. CalcIntersect known answers: rays aimed roughly at each object kind
    sc, _, _, _, _ = S.small_case("mesh")
    p = sc.pack()
    rays = random_rays(400, 901, ((-20, -30, -5), (20, 5, 30)))
    # aim half of them at the scene centre so that a good share hits
    tgt = np.array([0.0, 4.0, -2.0])
    rays[::2, 3:] = tgt - rays[::2, :3] + (rays[::2, 3:] * 6)
    isect = {"rays": rays}
    for obj in range(len(p.objects)):
        isect[f"obj{obj}"] = R.intersect_many(p, obj, rays)
    sc2, _, _, _, _ = S.small_case("inside")
    p2 = sc2.pack()
    rays2 = random_rays(300, 902, ((-10, -30, -8), (10, 0, 8)))
    isect["rays_inside"] = rays2
    for obj in range(len(p2.objects)):
        isect[f"inside_obj{obj}"] = R.intersect_many(p2, obj, rays2)
    np.savez_compressed(os.path.join(OUT, "intersect.npz"), **isect)
    # 3. Color operator table (color.h:33-51)
    rng = S.SplitMix64(903)
    vals = [0.0, 0.5, 1.0, 0.999999, 1.003, 1.5, 2.0, -0.002, -1.0, -3.5 / 200, 1e10, -1e10, float("nan"),
            float("inf"), 255.999 / 255, 256.0 / 255]
    vals += list(rng.array(40) * 1.2 - 0.1)
    cols = [(200, 100, 50), (255, 255, 255), (0, 1, 2), (221, 221, 221), (128, 127, 129)]
    rows = []
    for c in cols:
        for v in vals:
            rows.append([0, *c, 0, 0, 0, v, 0, 0, *R.color_op(0, c, v=v)])
            if v != 0:
                rows.append([1, *c, 0, 0, 0, v, 0, 0, *R.color_op(1, c, v=v)])
            rows.append([3, *c, 0, 0, 0, v, 1 - v, v * 0.5, *R.color_op(3, c, v=v, v2=1 - v, v3=v * 0.5)])
        for c2 in cols:
            rows.append([2, *c, *c2, 0, 0, 0, *R.color_op(2, c, c2)])
    np.save(os.path.join(OUT, "color_ops.npy"), np.array(rows, dtype=np.float64))
    # 4. Texture lookups (texture.h:81-122) on a tiny 8x4 texture and a 32x16 one
    tex = {}
    for tag, (tw, th, sx, sy, seed) in {"t8x4": (8, 4, 1.0, 1.0, 904), "t32x16": (32, 16, 0.5, 0.25, 905)}.items():
        t = B.Texture(S._small_texture(tw, th, seed), sx, sy)
        uv = S.SplitMix64(seed + 50).array(600).reshape(300, 2)
        uv[:4] = [[0, 0], [0.999999, 0.999999], [0.5, 0.5], [1.0, 1.0]]
        ang = uv * np.array([2 * 3.1415927, 3.1415927])
        tex[f"{tag}_rgb"] = t.rgb
        tex[f"{tag}_s"] = np.array([sx, sy])
        tex[f"{tag}_uv"] = uv
        tex[f"{tag}_plane"] = R.texture_many(t, 0, uv)
        tex[f"{tag}_ang"] = ang
        tex[f"{tag}_sphere"] = R.texture_many(t, 1, ang)
    np.savez_compressed(os.path.join(OUT, "texture_lookups.npz"), **tex)
    # 5. pixel values recorded from a run of the reference main.cpp path itself
    meta["baseline_md_samples"] = {
        "800x600_1light": {"(0,0)": [181] * 3, "(400,300)": [11] * 3, "(200,300)": [65, 130, 65],
                           "(600,200)": [180] * 3, "(400,75)": [181] * 3, "(799,599)": [170] * 3},
        "800x600_2lights": {"(0,0)": [193] * 3, "(400,300)": [84] * 3, "(200,300)": [57, 116, 57],
                            "(600,200)": [199] * 3, "(400,75)": [196] * 3, "(799,599)": [164] * 3},
        "3840x2160_1light": {"(0,0)": [181] * 3, "(1920,1080)": [11] * 3, "(960,1080)": [157] * 3,
                             "(2880,720)": [180] * 3, "(1920,270)": [181] * 3, "(3839,2159)": [161] * 3},
    }
    # larger frames: hashes only
    for (w, h, nl) in [(800, 600, 1), (800, 600, 2)]:
        frame = R.render(S.default_scene().pack(), S.default_camera(w, h), S.default_shader(nl), B.opts(w, h))
        meta[f"default_{w}x{h}_{nl}light"] = {"fnv1a64": f"{B.fnv1a64(frame):016x}"}
    # the default scene at the metric's 3840x2160: a third of the rows by the reference's classes (a thread needs a
    # minute for the frame), the hash of the whole frame as this build's checker renders it and as a build of the
    # reference's main.cpp itself was measured to render it
    w, h = 3840, 2160
    rows = R.render(S.default_scene().pack(), S.default_camera(w, h), S.default_shader(1), B.opts(w, h, row_begin=720, row_end=1440))
    from oracle.oracle_py import Oracle
    O = Oracle()
    full, _ = O.render(S.default_scene().pack(), S.default_camera(w, h), S.default_shader(1), B.opts(w, h), threads=0)
    assert np.array_equal(rows[720:1440], full[720:1440])
    meta["default_3840x2160_1light"] = {"fnv1a64": f"{O.fnv(full):016x}", "rows_720_1440_fnv1a64": f"{O.fnv(rows[720:1440]):016x}"}
    assert meta["default_3840x2160_1light"]["fnv1a64"] == "b7bc333b9d9ae34d"
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Benchmark of the accelerated render path: Mrays/s (primary + shadow) at 3840x2160.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--shard frames|slabs]
                    [--mode packet|allpairs]

The headline workload is the default scene at 3840x2160 (`default_4k`).  At N = 1 the two
throughput scenes (`spheres10k_8lights_4k`, `mesh100k_2lights_4k`) and the default scene under the
renderer HEAD of the reference selects (`default_bleed162_4k`: CalcSubPixel2 with shading3 on 162 gather
directions) are measured the same way after it and reported under "other_configs" of the same JSON line.  Searched scenes use the library's
default search (packet: beams against a direction-sorted table); the all-pairs search (every ray x
every primitive in fp32, the FP32-VALU-bound kernel) is timed beside it and reported in
"all_pairs" with its own roofline.

One step = one frame of the workload rendered by rt355_render_device into a framebuffer that
stays in HBM (scene, tables and framebuffer are all device-resident when the timed region
starts).  N > 1 is launched by torch.distributed.run, one rank per GPU; ranks render whole
frames independently (weak scaling, no data-path collective) or, with --shard slabs, row slabs
of one frame followed by the final-frame gather over RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak (= the f32-input MFMA peak), datasheet
PEAK_HBM_GBS = 8000.0     # HBM3E peak, datasheet (about 6300 GB/s is achievable)
# flops per ray/primitive test of the REFERENCE formulas (common sub-expressions shared):
ALGO_FLOPS = {"sphere": 18.0, "plane": 14.0, "tri_front": 50.0, "tri_culled": 13.0}
# flops the search kernel EXECUTES per pair: 3 FMA (6) + its share of the max-fold (8 max3 per
# 16 pairs, 2 compares each = 1) + 1/16 of the group compare
EXEC_FLOPS_PER_PAIR = 6.0 + 1.0 + 1.0 / 16.0


def build_workload(name: str):
    from raytracer_amd import scenes as S
    if name == "spheres10k_8lights_4k":
        w, h = 3840, 2160
        return S.spheres_scene(10000, seed=355), S.spheres_shader(8), S.default_camera(w, h), w, h
    if name in ("default_4k", "default_bleed162_4k"):
        w, h = 3840, 2160
        return S.default_scene(), S.default_shader(1), S.default_camera(w, h), w, h
    if name in ("mesh100k_2lights_4k", "mesh100k_bleed42_4k"):
        w, h = 3840, 2160
        return S.mesh_scene(100_000, seed=356), S.mesh_shader(), S.default_camera(w, h), w, h
    if name == "spheres10k_mirrors_8lights_4k":
        # the spheres scene with a reflecting floor (ks 0.4) and every third sphere a mirror (ks 0.5, every ninth 0.9)
        w, h = 3840, 2160
        sc = S.spheres_scene(10000, seed=355)
        for i, o in enumerate(sc.objects):
            if i == 0:
                o["ks"] = 0.4
            elif i % 3 == 0:
                o["ks"] = 0.9 if i % 9 == 0 else 0.5
        return sc, S.spheres_shader(8), S.default_camera(w, h), w, h
    if name == "default_mirrors_4k":  # the reference's scene with reflecting objects (main.cpp:93-112), unsearched
        w, h = 3840, 2160
        sc, sh, _, _, _ = S.small_case("mirrors")
        return sc, sh, S.default_camera(w, h), w, h
    if name == "glass_room_4k":
        # refraction (main.cpp:116-154): the reference's two spheres of glass, one a mirror too, in a closed room
        w, h = 3840, 2160
        sc, sh, _, _, _ = S.small_case("glass")
        return sc, sh, S.default_camera(w, h), w, h
    if name == "mesh1m_2lights_8k":
        w, h = 7680, 4320
        return S.mesh_scene(1_000_000, seed=356), S.mesh_shader(), S.default_camera(w, h), w, h
    if name == "spheres1k_2lights_small":  # quick functional run
        w, h = 640, 360
        return S.spheres_scene(1000, seed=355), S.spheres_shader(2), S.default_camera(w, h), w, h
    raise SystemExit(f"unknown workload {name}")


def gather_of(name: str):
    """(shading, direction set) of a workload: the renderer HEAD of the reference selects (CalcSubPixel2 with
    shading3, main.cpp:205) on a 162-direction set (an icosahedron subdivided twice), or Shader::shading."""
    from raytracer_amd import binding as B
    if name == "default_bleed162_4k":
        return B.SHADING_BLEED, B.geodesic(B.icosphere(2))
    if name == "mesh100k_bleed42_4k":  # a searched scene: the gather rays go through the scene table, one direction per pass
        return B.SHADING_BLEED, B.geodesic(B.icosphere(1))
    return B.SHADING_LIGHTS, None


def algorithmic_flops(scene, shader, cam, stats) -> float:
    """Reference-formula flops of one frame: sum over rays and primitives of the per-test figure.
    A triangle test counts 50 when the bundle origin lies on the triangle's front side (so rays can
    pass the back-face test), else 13."""
    from raytracer_amd import binding as B
    objs = scene.pack().objects
    n_pl = int((objs["type"] == B.OBJ_PLANE).sum())
    n_sp = int((objs["type"] == B.OBJ_SPHERE).sum())
    base = n_pl * ALGO_FLOPS["plane"] + n_sp * ALGO_FLOPS["sphere"]
    tris = np.concatenate(scene.triangles, axis=0) if scene.triangles else np.zeros((0, 15))
    n_lights = len(shader.lights)
    per_ray_primary = base
    per_ray_shadow = [base] * n_lights
    if len(tris):
        p0, p1, p2 = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
        nrm = np.cross(p1 - p0, p2 - p0)
        pc = (p0 + p1 + p2) / 3.0
        eye = np.array(list(cam.p_eye))
        front = ((nrm * (eye - pc)).sum(axis=1) > 0).sum()
        per_ray_primary += front * ALGO_FLOPS["tri_front"] + (len(tris) - front) * ALGO_FLOPS["tri_culled"]
        for k in range(n_lights):
            # a shadow ray can only pass the back-face test of triangles whose BACK faces the light
            lp = shader.lights[k]["position"]
            back = ((nrm * (lp - pc)).sum(axis=1) < 0).sum()
            per_ray_shadow[k] += back * ALGO_FLOPS["tri_front"] + (len(tris) - back) * ALGO_FLOPS["tri_culled"]
    shadow_per_light = stats["shadow_rays"] / max(1, n_lights)
    gather = stats.get("gather_rays", 0) * base  # gather rays meet every object (scenes without meshes only)
    return stats["primary_rays"] * per_ray_primary + sum(shadow_per_light * f for f in per_ray_shadow) + gather


def cpu_baseline(scene, shader, cam, w, h, stride, threads, shading=0, dirs=None):
    """The CPU checker (oracle/rt_oracle.c, a port of the reference path) timed on a strided pixel sample of
    the same frame, all `threads` cores and one; and, where the harness around the reference's own classes was
    built (oracle/_ref), that code on a few rows of the same frame, one thread as the reference runs."""
    from oracle.oracle_py import Oracle, Reference
    from raytracer_amd import binding as B
    O = Oracle()
    p = scene.pack()
    t0 = time.perf_counter()
    _, st = O.render(p, cam, shader, B.opts(w, h, shading=shading), threads=threads, stride=stride, dirs=dirs)
    dt = time.perf_counter() - t0
    count = lambda q: q["primary_rays"] + q["shadow_rays"] + q["gather_rays"] + q["secondary_rays"]
    rays = count(st)
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"every {stride}th pixel in x and y of the same frame ({st['primary_rays']} primary + "
                     f"{st['shadow_rays']} shadow + {st['gather_rays']} gather + {st['secondary_rays']} mirror or refracted rays, {dt:.2f} s)"}
    # one thread of the same port, on a sample a thread finishes in seconds
    s1 = stride * (4 if scene.n_primitives >= 1000 else 1)
    t0 = time.perf_counter()
    _, st1 = O.render(p, cam, shader, B.opts(w, h, shading=shading), threads=1, stride=s1, dirs=dirs)
    dt1 = time.perf_counter() - t0
    out["one_thread"] = {"value": count(st1) / dt1 / 1e6, "unit": "Mrays/s",
                         "sample": f"every {s1}th pixel in x and y ({dt1:.2f} s)"}
    # the reference's own classes (CalcIntersect, Shader::shading, Color ...) driven by the harness, one thread
    work = scene.n_primitives * (1 + len(shader.lights))
    rows = 64 if work < 200 else (1 if work < 200_000 else 0)
    if shading:
        rows = 16
    if Reference.available() and rows:
        R = Reference()
        o = B.opts(w, h, row_begin=h // 2, row_end=h // 2 + rows, shading=shading)
        _, sts = O.render(p, cam, shader, o, threads=threads, dirs=dirs)  # the slab's ray counts
        t0 = time.perf_counter()
        R.render(p, cam, shader, o, dirs=dirs)
        dtr = time.perf_counter() - t0
        out["reference_classes_one_thread"] = {
            "value": count(sts) / dtr / 1e6, "unit": "Mrays/s", "kind": "reference",
            "sample": f"rows {h // 2}..{h // 2 + rows - 1} of the same frame ({sts['primary_rays']} primary + "
                      f"{sts['shadow_rays']} shadow + {sts['gather_rays']} gather + {sts['secondary_rays']} mirror or refracted rays, {dtr:.2f} s); the harness "
                      f"compiles the reference's headers, main.cpp itself needs GLUT"}
    return out


def primary_bytes(stats, n_rays_padded, n_lights, searched):
    """Algorithmic HBM bytes of k_resolve_primary for one frame (all per-ray records are SoA): every ray
    slot reads its candidate count (if the camera rays were searched) and writes its object id; the
    camera rays' candidates are read once (4 B each; counted as one per hit, the typical case, because
    the statistics do not split candidates by bundle); a hit writes ph, nh, reh (72 B), the border
    weight (8 B) and its two colours (8 B); per light, a shadow ray writes its fp32 record (16 B) and,
    when the rays are direction-sorted, every slot writes a bin id and a rank (8 B)."""
    hits = stats["primary_hits"]
    b = n_rays_padded * (4 + (4 if searched else 0)) + hits * (4 if searched else 0) + hits * 88
    b += stats["shadow_rays"] * 16 + (n_lights * n_rays_padded * 8 if searched else 0)
    return float(b)


def shade_bytes(stats, n_rays_padded, n_lights, w, h, searched):
    """Algorithmic HBM bytes of k_shade_group for one frame: every slot reads its object id; a hit
    reads its 88-byte record once; every shadow ray reads its candidate count and candidate ids
    (4 B each); one RGB8 pixel is stored per pixel."""
    hits = stats["primary_hits"]
    cands = max(0, stats["exact_tests"] - hits) if searched else 0
    return float(n_rays_padded * 4 + hits * 88 + (stats["shadow_rays"] * 4 + cands * 4 if searched else 0) + w * h * 3)


def measured_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` on `workload` as the PMC counters gave them (rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE, separate passes of this same bench command: tools/traffic_passes.sh), read from the
    committed summary; FETCH_SIZE doubled, as gfx950 tallies a 128-byte request as 64 bytes.  None when the
    summary does not hold the pair."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    try:
        rec = json.load(open(path))[workload][kernel]
    except (OSError, KeyError, ValueError):
        return None
    return {"bytes": (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0, "fetch_kib_counter": rec["fetch_kb"],
            "write_kib_counter": rec["write_kb"], "source": "profiles/r1_traffic.json"}


def run_frames(r, cam, shader, o, fb, stream, steps, extra_step=None, detail=False):
    """`steps` frames back to back, each synchronised before the next is enqueued.  With detail=True the
    HIP-event kernel times of every frame are read back and summed (not done inside timed regions)."""
    acc = {"search_ms": 0.0, "search_launches": 0, "resolve_ms": 0.0, "resolve_launches": 0, "pairs": 0,
           "primary_ms": 0.0, "shade_ms": 0.0, "shade_launches": 0, "frames": steps}
    stats = None
    for _ in range(steps):
        r.render_device(cam, shader, o, fb.data_ptr(), stream)
        if extra_step:
            extra_step()
        stats = r.sync()
        if detail:
            ms, n = r.search_kernel_ms()
            acc["search_ms"] += ms; acc["search_launches"] += n
            ms, n = r.resolve_kernel_ms()
            acc["resolve_ms"] += ms; acc["resolve_launches"] += n
            per = r.resolve_launch_ms()
            acc["primary_ms"] += per[0]; acc["shade_ms"] += sum(per[1:]); acc["shade_launches"] += len(per) - 1
            acc["pairs"] += stats["pair_tests"]
    return acc, stats


def measure(name, args, ctx, headline):
    """One workload: W warm-up frames, K timed frames between barriers (max over ranks), then, on rank 0 and
    outside the timed region, kernel-time breakdown, rooflines, the all-pairs comparison and the CPU baseline."""
    import torch
    from raytracer_amd import binding as B
    from raytracer_amd.api import Renderer
    from raytracer_amd.sharding import gather_frame, slab_rows
    dist, world, rank = ctx["dist"], ctx["world"], ctx["rank"]

    scene, shader, cam, w, h = build_workload(name)
    shading, dirs = gather_of(name)
    r = Renderer(scene)
    if dirs is not None:
        r.set_directions(dirs)
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    mode = B.MODE_PACKET if args.mode == "packet" else B.MODE_ALLPAIRS
    if args.shard == "slabs" and world > 1:
        b, e = slab_rows(h, world)[rank]
        o = B.opts(w, h, row_begin=b, row_end=e, mode=mode, shading=shading)
    else:
        o = B.opts(w, h, mode=mode, shading=shading)
    stream = torch.cuda.current_stream().cuda_stream
    gather = (lambda: gather_frame(fb, rank, world, dst=0)) if (args.shard == "slabs" and world > 1) else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_frames(r, cam, shader, o, fb, stream, args.warmup, gather)
    barrier()
    t0 = time.perf_counter()
    _, stats = run_frames(r, cam, shader, o, fb, stream, args.steps, gather)
    barrier()
    dt = time.perf_counter() - t0
    rays_step = float(stats["primary_rays"] + stats["shadow_rays"] + stats["gather_rays"] + stats["secondary_rays"])
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=ctx["reduce_device"])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        rays_all = torch.tensor([rays_step], dtype=torch.float64, device=ctx["reduce_device"])
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        rays_step = float(rays_all.item())
    if rank != 0:
        r.close()
        return None

    def valu_roofline(acc, stats):
        """FP32-VALU roofline of the all-pairs search kernel."""
        avg_ms = acc["search_ms"] / acc["search_launches"]
        pairs_per_launch = acc["pairs"] / acc["search_launches"]
        exec_tflops = pairs_per_launch * EXEC_FLOPS_PER_PAIR / (avg_ms * 1e-3) / 1e12
        algo = algorithmic_flops(scene, shader, cam, stats) * acc["frames"] / acc["search_launches"]
        algo_tflops = algo / (avg_ms * 1e-3) / 1e12
        return {"bound": "valu_fp32", "kernel": "k_search", "achieved": round(exec_tflops, 2), "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(exec_tflops / PEAK_FP32_TFLOPS, 4), "traffic": None,
                "launches_per_frame": acc["search_launches"] // acc["frames"], "avg_launch_ms": round(avg_ms, 4),
                "pairs_per_launch": pairs_per_launch, "flops_per_pair_executed": EXEC_FLOPS_PER_PAIR,
                "reference_formula_equivalent": {"tflops": round(algo_tflops, 2),
                                                 "frac_of_peak": round(algo_tflops / PEAK_FP32_TFLOPS, 4),
                                                 "flops_per_test": ALGO_FLOPS}}

    ms_per_step = dt * 1e3 / args.steps
    value = rays_step * args.steps / dt / 1e6
    n_rays_padded = ((w + 31) // 32) * (((o.row_end or h) - o.row_begin + 31) // 32) * 1024
    out = {"value": round(value, 2), "unit": "Mrays/s", "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4),
           "config": {"workload": name, "width": w, "height": h, "primitives": scene.n_primitives,
                      "lights": len(shader.lights), "alias": 0, "shard": args.shard, "search": args.mode,
                      "rays_per_frame": {"primary": stats["primary_rays"], "shadow": stats["shadow_rays"],
                                         "gather": stats["gather_rays"], "mirror": stats["secondary_rays"]},
                      "shading": {0: "Shader::shading", 1: "Shader::shading2", 2: "Shader::shading3"}[shading],
                      "gather_directions": 0 if dirs is None else len(dirs),
                      "candidates_evaluated_fp64": stats["exact_tests"], "overflow_rays": stats["overflow_rays"],
                      "packet_survivors": stats["packet_survivors"]}}
    # ---- kernel-time breakdown: a few more frames, HIP-event times read back after each
    K = 3
    acc, stats = run_frames(r, cam, shader, o, fb, stream, K, None, detail=True)  # rank 0 alone: no gather here
    searched = acc["search_launches"] > 0
    nl = len(shader.lights)
    out["kernel_time"] = {"search_ms_per_frame": round(acc["search_ms"] / K, 4),
                          "resolve_primary_ms_per_frame": round(acc["primary_ms"] / K, 4),
                          "shade_ms_per_frame": round(acc["shade_ms"] / K, 4), "frames_measured": K}

    def hbm_roofline(kernel, bytes_frame, ms_total, launches):
        avg_ms = ms_total / max(1, launches)
        gbs = bytes_frame * K / max(1, launches) / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "launches_per_frame": launches // K,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_frame": bytes_frame}

    if (stats["secondary_rays"] > 0 or stats["gather_rays"] > 0) and searched:
        # a frame rendered level by level: the launches of all levels are summed in kernel_time (search includes
        # k_search_general, resolve_primary is level 0's, shade the rest); the per-kernel byte models below describe
        # one level and are not applied
        out["roofline"] = {"bound": "hbm", "kernel": "levels", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": None, "traffic": None,
                           "note": "multi-level frame: see profiles/ for the per-kernel trace"}
    elif not searched:
        # k_render_direct: the reference's own algorithm (every primitive tested for the camera ray and for every
        # shadow ray, fp64), one thread per pixel.  Priced as the reference-formula flops of those tests against
        # the FP32 vector peak; the shading of each hit (normalisations, x^25, 8-bit colour arithmetic), which
        # the per-test figure does not count, is most of its instructions on a 4-object scene.
        flops = algorithmic_flops(scene, shader, cam, stats)
        avg_ms = acc["primary_ms"] / K
        tf = flops / (avg_ms * 1e-3) / 1e12
        refracts = bool((scene.pack().objects["fresnel"] < 1).any()) and not shading
        out["roofline"] = {"bound": "valu_fp32", "kernel": "k_render_gather" if shading else ("k_render_refract" if refracts else "k_render_direct"), "achieved": round(tf, 3),
                           "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_TFLOPS, 5),
                           "traffic": None, "launches_per_frame": 1, "avg_launch_ms": round(avg_ms, 4),
                           "algorithmic_flops_per_frame": flops, "flops_per_test": ALGO_FLOPS,
                           "algorithmic_bytes_per_frame": float(3 * w * h),
                           "note": "arithmetic is fp64 (half the fp32 rate); per-hit shading is not in the flop count"
                                   + ("; nor are the tests of the mirror and refracted rays" if stats["secondary_rays"] else "")}
    else:
        rooflines = {"k_resolve_primary": hbm_roofline("k_resolve_primary", primary_bytes(stats, n_rays_padded, nl, True),
                                                       acc["primary_ms"], K)}
        if acc["shade_launches"]:
            rooflines["k_shade_group"] = hbm_roofline("k_shade_group", shade_bytes(stats, n_rays_padded, nl, w, h, True),
                                                      acc["shade_ms"], acc["shade_launches"])
        if args.mode == "allpairs":
            rooflines["k_search"] = valu_roofline(acc, stats)
        else:
            # packet search: the tables stay in L2; per ray it reads the list entry (4 B, shadow rays only) and
            # the fp32 record (16 B) and writes the candidate count (4 B) and 4 B per candidate
            sb = stats["primary_rays"] * 20.0 + stats["shadow_rays"] * 24.0 + stats["exact_tests"] * 4.0
            rooflines["k_search_packet"] = hbm_roofline("k_search_packet", sb, acc["search_ms"], acc["search_launches"])
        times = {"k_resolve_primary": acc["primary_ms"], "k_shade_group": acc["shade_ms"],
                 "k_search_packet": acc["search_ms"], "k_search": acc["search_ms"]}
        dominant = max(rooflines, key=lambda k: times[k])
        out["roofline"] = rooflines[dominant]
        out["roofline_other_kernels"] = {k: v for k, v in rooflines.items() if k != dominant}
    for rf in [out["roofline"], *out.get("roofline_other_kernels", {}).values()]:
        t = measured_traffic(name, rf["kernel"]) if (args.mode == "packet" and rf.get("frac") is not None) else None
        if t:
            rf["traffic"] = t["bytes"]
            rf["traffic_detail"] = t
    if world == 1 and args.mode == "packet" and not args.no_allpairs and searched:
        o2 = B.opts(w, h, mode=B.MODE_ALLPAIRS)
        run_frames(r, cam, shader, o2, fb, stream, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        acc2, stats2 = run_frames(r, cam, shader, o2, fb, stream, 3, detail=True)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        out["all_pairs"] = {"value": round(rays_step * 3 / dt2 / 1e6, 2), "unit": "Mrays/s", "steps": 3,
                            "ms_per_step": round(dt2 * 1e3 / 3, 4),
                            "search_ms_per_frame": round(acc2["search_ms"] / 3, 4),
                            "roofline": valu_roofline(acc2, stats2)}
    if not args.no_cpu_baseline and world == 1:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        stride = args.cpu_stride
        if stride == 0:
            work = scene.n_primitives * (1 + len(shader.lights))
            stride = 1 if work < 200 else (4 if work < 5000 else 16)
        if shading:
            stride = 4
        out["cpu_baseline"] = cpu_baseline(scene, shader, cam, w, h, stride, threads, shading, dirs)
    r.close()
    return out


HEADLINE = "default_4k"
OTHERS = ["spheres10k_8lights_4k", "mesh100k_2lights_4k", "default_bleed162_4k", "spheres10k_mirrors_8lights_4k",
          "glass_room_4k"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=HEADLINE)
    ap.add_argument("--shard", choices=["frames", "slabs"], default="frames")
    ap.add_argument("--mode", choices=["packet", "allpairs"], default="packet",
                    help="search of searched scenes: packet = beams against a direction-sorted table "
                         "(library default); allpairs = fp32 test of every ray/primitive pair")
    ap.add_argument("--no-others", action="store_true", help="only the workload named, no other_configs")
    ap.add_argument("--no-allpairs", action="store_true", help="skip the all-pairs comparison run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=0, help="pixel stride of the CPU sample (0 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # rehearsal of the N > 1 path on a one-GPU box: every rank on device 0, collectives over gloo (RCCL refuses
    # two ranks on one device)
    share = bool(os.environ.get("RT355_BENCH_SHARE_GPU"))
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = {"dist": dist, "world": world, "rank": rank, "reduce_device": "cpu" if share else "cuda"}

    head = measure(args.workload, args, ctx, headline=True)
    others = {}
    if world == 1 and args.workload == HEADLINE and not args.no_others:
        for name in OTHERS:
            others[name] = measure(name, args, ctx, headline=False)
    if rank == 0:
        # (a workload that casts gather rays counts them too)
        out = {"metric": "Mrays/sec (primary+shadow) at 3840x2160", "value": head["value"], "unit": "Mrays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak" if (args.shard == "frames" or world == 1) else "strong",
               "vs_baseline": None, "dtype": "f64" if head["kernel_time"]["search_ms_per_frame"] == 0 else "f32 search + f64 resolve",
               "data": "synthetic"}
        for k in ("config", "kernel_time", "roofline", "roofline_other_kernels", "all_pairs", "cpu_baseline"):
            if k in head:
                out[k] = head[k]
        if others:
            out["other_configs"] = others
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

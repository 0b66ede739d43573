#!/usr/bin/env python
"""Benchmark of the accelerated render path: Mrays/s (primary + shadow) at 3840x2160.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--shard frames|slabs|tiles]
                    [--mode packet|allpairs]

The headline workload is the default scene at 3840x2160 (`default_4k`).  At N = 1 the throughput scenes
(`spheres10k_8lights_4k`, `mesh100k_2lights_4k`, the 60-camera sweep `mesh100k_sweep60`), the default scene
under the renderer HEAD of the reference selects (`default_bleed162_4k`: CalcSubPixel2 with shading3 on 162
gather directions), the mirror scene and the glass room are measured the same way after it and reported
under "other_configs" of the same JSON line.  Searched scenes use the library's default search (packet:
beams against a direction-sorted table); the all-pairs search (every ray x every primitive in fp32, the
FP32-VALU-bound kernel) is timed beside it and reported in "all_pairs" with its own roofline.

One step = one frame of the workload rendered by rt355_render -- the call that stands where the reference has
setPixels (main.cpp:339-376,640) -- into a page-locked host pixmap: the scene and its tables are resident in
HBM when the timed region starts, and the readback of the frame is inside it (`value`).  The same frames
rendered by rt355_render_device with the framebuffer left in HBM are timed beside it
(`value_device_resident`).  A sweep workload's step is its whole camera sweep.

N > 1: one rank per GPU, launched by torch.distributed.run or, when WORLD_SIZE is not set, spawned by this
script itself before it touches the GPU.  Ranks render whole frames independently (--shard frames: weak
scaling, no data-path collective; a sweep deals its frames round robin, frame f on rank f mod N, a fixed
total) or one frame between them (--shard slabs: contiguous row slabs; --shard tiles: interleaved 32-row
tiles; --shard balanced: slabs whose cuts follow the times the ranks' shares took; a fixed total), the frame
coming together in one pixmap in host memory that all ranks write their rows into (or, --gather xgmi, on
rank 0's GPU over RCCL, read back from there).

Prints ONE compact JSON line on rank 0 as the last thing the run writes (a few kilobytes: headline, its roofline
and CPU baseline, five numbers per other workload, the scaling workloads); the full record of every workload goes to
bench_detail.json beside this script (and to gpurun_out/ where that directory exists).  Progress lines go to stderr
before it, prefixed "bench:".

The scaling workloads -- the million-triangle 8K frame and the 10 000-sphere 4K frame in balanced slabs, and
the 60-camera sweep -- are measured after the headline at every N: at N = 1 the one GPU renders all of each, at N > 1
the ranks share them ("scaling_workloads" of the line).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
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
TRAFFIC_FILES = ("r3_traffic.json", "r2_traffic.json")  # the newest summary that holds the pair
ISSUE_FILE = "r3_issue.json"
SIMDS = 1024               # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # peak engine clock
LINE_LIMIT = 8000          # the final line stays below this many bytes


def build_workload(name: str):
    from raytracer_amd import scenes as S
    if name in ("spheres10k_8lights_4k", "spheres10k_tiles"):  # _tiles: one frame shared between the ranks
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
    if name == "spheres10k_glass_8lights_4k":
        # refraction on a scene of many primitives: every tenth sphere of the spheres scene of glass (some of them
        # mirrors too), the whole inside an Environment so that every ray ends somewhere (main.cpp:116-154)
        w, h = 3840, 2160
        sc = S.spheres_scene(10000, seed=355)
        for i, o in enumerate(sc.objects):
            if i and i % 10 == 0:
                o["fresnel"] = 0.3 if i % 20 else 0.6
                o["ior"] = 1.5 if i % 30 else 1.2
                if i % 40 == 0:
                    o["ks"] = 0.4
        sc.add_environment((0, 60, 10), 600, (110, 150, 235))
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
    if name in ("mesh1m_2lights_8k", "mesh1m_8k_tiles", "mesh1m_8k_slabs"):
        # the 1 M-triangle scene at 7680x4320; _tiles / _slabs: one frame shared between the ranks
        w, h = 7680, 4320
        return S.mesh_scene(1_000_000, seed=356), S.mesh_shader(), S.default_camera(w, h), w, h
    if name == "mesh100k_sweep60":  # 60 cameras orbiting the triangle scene; one step = the whole sweep
        w, h = 3840, 2160
        return S.mesh_scene(100_000, seed=356), S.mesh_shader(), S.orbit_camera(0, w, h), w, h
    if name == "mesh4k_sweep6_small":  # quick functional run of the sweep
        w, h = 320, 180
        return S.mesh_scene(4000, seed=356, tex_size=64), S.mesh_shader(), S.orbit_camera(0, w, h), w, h
    if name == "spheres1k_2lights_small":  # quick functional run
        w, h = 640, 360
        return S.spheres_scene(1000, seed=355), S.spheres_shader(2), S.default_camera(w, h), w, h
    raise SystemExit(f"unknown workload {name}")


def sweep_of(name: str, w: int, h: int):
    """The cameras of a sweep workload (one step renders all of them), or None."""
    from raytracer_amd import scenes as S
    if name == "mesh100k_sweep60":
        return [S.orbit_camera(f, w, h, n_frames=60) for f in range(60)]
    if name == "mesh4k_sweep6_small":
        return [S.orbit_camera(10 * f, w, h, n_frames=60) for f in range(6)]
    return None


# the partition a workload is meant for when --shard is left at its default
DEFAULT_SHARD = {"mesh1m_8k_tiles": "balanced", "mesh1m_8k_slabs": "slabs", "spheres10k_tiles": "balanced"}


def gather_of(name: str):
    """(shading, direction set) of a workload: the renderer HEAD of the reference selects (CalcSubPixel2 with
    shading3, main.cpp:205) on a 162-direction set (an icosahedron subdivided twice), or Shader::shading."""
    from raytracer_amd import binding as B
    if name == "default_bleed162_4k":
        return B.SHADING_BLEED, B.geodesic(B.icosphere(2))
    if name == "mesh100k_bleed42_4k":  # a searched scene: the gather rays go through the scene table
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


def count_rays(q) -> float:
    return float(q["primary_rays"] + q["shadow_rays"] + q["gather_rays"] + q["secondary_rays"])


def cpu_baseline(scene, shader, cam, w, h, stride, threads, shading=0, dirs=None, gpu_frame=None):
    """The CPU checker (oracle/rt_oracle.c, a port of the reference path) timed on a strided pixel sample of
    the same frame, all `threads` cores and one; and, where the harness around the reference's own classes was
    built (oracle/_ref), that code on a few rows of the same frame, one thread as the reference runs.  The
    sample's pixels must equal the frame the GPU rendered in the timed region (gpu_frame), byte for byte."""
    from oracle.oracle_py import Oracle, Reference
    from raytracer_amd import binding as B
    O = Oracle()
    p = scene.pack()
    # (a frame the cores finish in a fraction of a second is repeated until about a second of wall time has
    # gone by, so that starting the threads is not what is timed)
    reps, t0 = 0, time.perf_counter()
    while True:
        sample, st = O.render(p, cam, shader, B.opts(w, h, shading=shading), threads=threads, stride=stride, dirs=dirs)
        reps += 1
        dt_all = time.perf_counter() - t0
        if dt_all >= 1.0 or reps >= 64:
            break
    dt = dt_all / reps
    parity = "not checked"
    if gpu_frame is not None:
        a, b = sample[::stride, ::stride], gpu_frame[::stride, ::stride]
        ndiff = int((a != b).any(axis=2).sum())
        if ndiff:
            raise SystemExit(f"parity: {ndiff} of the {a.shape[0] * a.shape[1]} pixels the CPU checker sampled differ from the GPU frame")
        parity = "ok"
    out = {"value": count_rays(st) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "parity_sample": parity, "sample_short": f"every {stride}th pixel in x and y of the same frame, {reps} x {dt:.2f} s",
           "sample": f"every {stride}th pixel in x and y of the same frame ({st['primary_rays']} primary + "
                     f"{st['shadow_rays']} shadow + {st['gather_rays']} gather + {st['secondary_rays']} mirror or refracted rays, {reps} x {dt:.2f} s)"}
    # one thread of the same port, on a sample a thread finishes in seconds
    s1 = stride * (4 if scene.n_primitives >= 1000 else 1)
    if scene.n_primitives < 500_000:  # (a thread needs minutes for any sample of a larger scene)
        t0 = time.perf_counter()
        _, st1 = O.render(p, cam, shader, B.opts(w, h, shading=shading), threads=1, stride=s1, dirs=dirs)
        dt1 = time.perf_counter() - t0
        out["one_thread"] = {"value": count_rays(st1) / dt1 / 1e6, "unit": "Mrays/s",
                             "sample": f"every {s1}th pixel in x and y ({dt1:.2f} s)"}
    # the reference's own classes (CalcIntersect, Shader::shading, Color ...) driven by the harness, one thread
    work = scene.n_primitives * (1 + len(shader.lights))
    rows = 64 if work < 200 else (1 if work < 200_000 else 0)
    if shading:  # (every gather ray tests every primitive: rows of a small scene only)
        rows = 16 if scene.n_primitives * len(dirs) < 5000 else 0
    if Reference.available() and rows:
        R = Reference()
        o = B.opts(w, h, row_begin=h // 2, row_end=h // 2 + rows, shading=shading)
        _, sts = O.render(p, cam, shader, o, threads=threads, dirs=dirs)  # the slab's ray counts
        t0 = time.perf_counter()
        R.render(p, cam, shader, o, dirs=dirs)
        dtr = time.perf_counter() - t0
        out["reference_classes_one_thread"] = {
            "value": count_rays(sts) / dtr / 1e6, "unit": "Mrays/s", "kind": "reference",
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
    for name in TRAFFIC_FILES:
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))[workload][kernel]
        except (OSError, KeyError, ValueError):
            continue
        return {"bytes": (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0, "fetch_kib_counter": rec["fetch_kb"],
                "write_kib_counter": rec["write_kb"], "source": "profiles/" + name}
    return None


def measured_issue(workload, kernel, avg_launch_ms):
    """Share of the chip's vector-issue slots a launch of `kernel` on `workload` used: wave-level vector instructions
    per launch as the SQ counters gave them (rocprofv3 --pmc SQ_ACTIVE_INST_VALU ..., a separate pass of this same
    bench command: tools/pmc_pass.sh, summarised by tools/issue_merge.py into the committed file), four cycles each,
    over the cycles 1 024 SIMDs have in the launch's duration as THIS run measured it.  None when the summary does
    not hold the pair."""
    path = os.path.join(ROOT, "profiles", ISSUE_FILE)
    try:
        rec = json.load(open(path))[workload][kernel]
    except (OSError, KeyError, ValueError):
        return None
    if not avg_launch_ms:
        return None
    inst = float(rec["valu_inst_per_launch"])
    slots = SIMDS * CLOCK_HZ * avg_launch_ms * 1e-3 / 4.0
    return {"bound": "valu_issue", "kernel": kernel, "achieved": round(inst / (avg_launch_ms * 1e-3) / 1e12, 4), "peak": round(SIMDS * CLOCK_HZ / 4.0 / 1e12, 4),
            "unit": "T wave-instructions/s", "frac": round(inst / slots, 4), "traffic": None, "avg_launch_ms": round(avg_launch_ms, 4),
            "valu_inst_per_launch": inst, "profiled_avg_launch_ms": rec.get("avg_ms"), "source": "profiles/" + ISSUE_FILE}


def run_frames(r, cam, shader, o, fb, stream, steps, detail=False):
    """`steps` frames by rt355_render_device, the framebuffer left in HBM, each synchronised before the next is
    enqueued.  With detail=True the HIP-event kernel times of every frame are read back and summed."""
    acc = {"search_ms": 0.0, "search_launches": 0, "resolve_ms": 0.0, "resolve_launches": 0, "pairs": 0,
           "primary_ms": 0.0, "shade_ms": 0.0, "shade_launches": 0, "frames": steps}
    stats = None
    for _ in range(steps):
        r.render_device(cam, shader, o, fb.data_ptr(), stream)
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


def measure(name, args, ctx, light=False):
    """One workload: W warm-up steps, K timed steps between barriers (max over ranks), then, on rank 0 and
    outside the timed region, the device-resident figure, kernel-time breakdown, rooflines, the all-pairs
    comparison and the CPU baseline (light=True: the timed steps alone)."""
    import torch
    from raytracer_amd import binding as B
    from raytracer_amd.api import HostPixmap, Renderer
    from raytracer_amd.sharding import SharedPixmap, SlabBalancer, frames_for_rank, gather_frame, host_barrier, rows_of, TILE_ROWS
    dist, world, rank = ctx["dist"], ctx["world"], ctx["rank"]

    scene, shader, cam, w, h = build_workload(name)
    shading, dirs = gather_of(name)
    cams = sweep_of(name, w, h)
    shard = args.shard or DEFAULT_SHARD.get(name, "frames")
    if shard == "balanced" and args.gather != "host":
        shard = "tiles"  # (the shares' times travel through the shared pixmap's page)
    split = world > 1 and shard in ("slabs", "tiles", "balanced") and cams is None  # one frame between the ranks
    r = Renderer(scene)
    if dirs is not None:
        r.set_directions(dirs)
    mode = B.MODE_PACKET if args.mode == "packet" else B.MODE_ALLPAIRS
    o = B.opts(w, h, mode=mode, shading=shading)
    stream = torch.cuda.current_stream().cuda_stream
    host = HostPixmap(w, h)  # page-locked, in place of `new unsigned char[width * height * 3]` (main.cpp:544)
    fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")

    shared = None
    if split:
        balancer = None
        if shard == "tiles":
            o_part = B.opts(w, h, mode=mode, shading=shading, tile_rows=TILE_ROWS, tile_count=world, tile_phase=rank)
        elif shard == "balanced":
            # contiguous slabs whose cuts follow the times the shares took in the frame before (every rank keeps the
            # same balancer and feeds it the same numbers); the cuts move in the warm-up steps and stay for the timed ones
            balancer = SlabBalancer(h, world)
            b, e = balancer.slabs()[rank]
            o_part = B.opts(w, h, row_begin=b, row_end=e, mode=mode, shading=shading)
        else:
            b, e = rows_of(h, world, rank, "slabs")[0]
            o_part = B.opts(w, h, row_begin=b, row_end=e, mode=mode, shading=shading)
        host_t = torch.from_numpy(host.array)
        shared = SharedPixmap(w, h, rank, world, dist) if args.gather == "host" else None
        if shared is not None:
            import ctypes as C
            from raytracer_amd.api import lib, _check
            from raytracer_amd.binding import CStats
            rt355_render, st_part = lib().rt355_render, CStats()
            part_args = (r.handle, C.byref(cam), C.byref(shader.c), C.byref(o_part), shared.array.ctypes.data, C.byref(st_part))

        balance = {"on": balancer is not None, "best": None, "best_cuts": None}

        def step():
            if shared is not None:
                # this rank's rows straight into the one pixmap all ranks share, over this GPU's own link to the host;
                # the frame is complete when every rank's rows have arrived
                t_call = time.perf_counter()
                rc = rt355_render(*part_args)
                if rc:
                    _check(rc, "rt355_render")
                if balance["on"]:
                    times = shared.barrier_with_times(time.perf_counter() - t_call)
                    if balance["best"] is None or max(times) < balance["best"]:
                        balance["best"], balance["best_cuts"] = max(times), list(balancer.cuts)
                    balancer.update(times)
                    o_part.row_begin, o_part.row_end = balancer.slabs()[rank]
                else:
                    shared.barrier()  # (arrival counters in the shared mapping itself: microseconds, no collective)
                st = st_part.as_dict()
                return count_rays(st), st
            # this rank's rows into its framebuffer in HBM, the final frame gathered on rank 0 (RCCL), read back there
            r.render_device(cam, shader, o_part, fb.data_ptr(), stream)
            st = r.sync()
            full = gather_frame(fb, rank, world, dst=0, shard=shard, group=ctx["gpu_group"])
            if rank == 0:
                host_t.copy_(full)
            return count_rays(st), st
    elif cams is not None:
        mine = frames_for_rank(len(cams), rank, world) if world > 1 else list(range(len(cams)))
        # rt355_render_sweep: frame f on rank f mod N, every frame with its own camera (the eye's tables are rebuilt)
        # and its own page-locked pixmap; the library reads a frame back while it renders the next.  With
        # --in-flight 2 the rank's frames are dealt to two scene handles, each sweeping its share from a host thread
        # of its own (the calls block; ctypes releases the interpreter lock), which also fills the gaps between one
        # handle's frames.
        import threading
        n_fly = max(1, min(args.in_flight, len(mine)))
        sweep_pix = [HostPixmap(w, h) for _ in mine]
        fly = [r] + [Renderer(scene) for _ in range(n_fly - 1)]
        extra_handles = fly[1:]
        shares = [list(range(k, len(mine), n_fly)) for k in range(n_fly)]

        def step():
            out = [None] * n_fly

            def run(k):
                out[k] = fly[k].render_sweep([cams[mine[i]] for i in shares[k]], shader, o, [sweep_pix[i].array for i in shares[k]])
            threads = [threading.Thread(target=run, args=(k,)) for k in range(1, n_fly)]
            for t in threads:
                t.start()
            run(0)
            for t in threads:
                t.join()
            if any(s_ is None for s_ in out):
                raise RuntimeError("a sweep thread failed")
            rays = sum(count_rays(st) for sts in out for st in sts)
            last_k = (len(mine) - 1) % n_fly
            return rays, out[last_k][-1]
    else:
        # the call itself and nothing else in the loop: arguments and the statistics record are made once, the
        # rays of a step are counted from the record after the loop (every frame of this workload casts the same)
        import ctypes as C
        from raytracer_amd.api import lib, _check
        from raytracer_amd.binding import CStats
        rt355_render, st_c = lib().rt355_render, CStats()
        call_args = (r.handle, C.byref(cam), C.byref(shader.c), C.byref(o), host.array.ctypes.data, C.byref(st_c))

        def step():
            rc = rt355_render(*call_args)
            if rc:
                _check(rc, "rt355_render")
            return None, None

    def barrier():
        if world > 1:
            host_barrier(dist)
        torch.cuda.synchronize()

    import gc
    # (slabs that follow the shares' times need a few frames to find their cuts: at least six untimed steps for them)
    for _ in range(max(args.warmup, 6) if (split and shared is not None and balancer is not None) else args.warmup):
        step()
    if split and shared is not None and balancer is not None and balance["best_cuts"] is not None:
        # the timed steps run on the best cuts the warm-up found (the same on every rank)
        balance["on"] = False
        balancer.cuts = balance["best_cuts"]
        o_part.row_begin, o_part.row_end = balancer.slabs()[rank]
    barrier()
    gc.disable()  # no collector pause inside the timed steps
    t0 = time.perf_counter()
    rays_step, stats = 0.0, None
    for _ in range(args.steps):
        rays_step, st_ = step()
        stats = st_ or stats
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if stats is None:  # the host-pixmap loop: the record of the last frame
        stats = st_c.as_dict()
        rays_step = count_rays(stats)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=ctx["reduce_device"])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # whole-job ray counts: every rank's share of the step (and, for a frame shared between the ranks, of the frame)
        keys = ("primary_rays", "shadow_rays", "gather_rays", "secondary_rays", "exact_tests", "overflow_rays", "packet_survivors")
        rays_all = torch.tensor([rays_step] + [float(stats[k]) for k in keys], dtype=torch.float64, device=ctx["reduce_device"])
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        rays_step = float(rays_all[0].item())
        if split:
            stats = dict(stats, **{k: int(v) for k, v in zip(keys, rays_all[1:].tolist())})
    if cams is not None:
        # the frame the sweep ends on lies in the pixmap of the handle that drew it
        gpu_last = sweep_pix[-1].array.copy()
        for rr in extra_handles:
            rr.close()
        for hp in sweep_pix:
            hp.close()
    if rank != 0:
        if shared is not None:
            shared.close()
        r.close()
        host.close()
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
    gpu_frame = gpu_last if cams is not None else host.array.copy()  # the last frame of the timed region, as the caller received it
    shared_registered = shared is not None and shared.registered
    if shared is not None:
        gpu_frame = shared.array.copy()
        shared.close()
    # (a frame the ranks share, and a sweep whose frames they deal out, are the same total work at every N -- 1 included)
    fixed_total = shard in ("slabs", "tiles", "balanced") or cams is not None
    out = {"value": round(value, 2), "unit": "Mrays/s", "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4),
           "timed": (("rt355_render of this rank's rows into the one host pixmap all ranks share" + ("" if shared_registered else " (pageable: through each rank's staging memory)") + " + barrier" if args.gather == "host" else
                      "rt355_render_device of this rank's rows + gather of the final frame on rank 0 + its readback") if split else
                     ("rt355_render_sweep into page-locked host pixmaps, one per frame (readback included)" if cams is not None else
                      "rt355_render into a page-locked host pixmap (readback of the frame included)")),
           "scaling": "strong" if fixed_total else "weak",
           "config": {"workload": name, "width": w, "height": h, "primitives": scene.n_primitives,
                      "lights": len(shader.lights), "alias": 0, "shard": shard if (world > 1 or cams is None) else "frames",
                      "frames_per_step": len(cams) if cams is not None else 1,
                      "frames_in_flight": n_fly if cams is not None else 1, "search": args.mode,
                      "rays_per_frame": {"primary": stats["primary_rays"], "shadow": stats["shadow_rays"],
                                         "gather": stats["gather_rays"], "mirror": stats["secondary_rays"]},
                      "shading": {0: "Shader::shading", 1: "Shader::shading2", 2: "Shader::shading3"}[shading],
                      "gather_directions": 0 if dirs is None else len(dirs),
                      "candidates_evaluated_fp64": stats["exact_tests"], "overflow_rays": stats["overflow_rays"],
                      "packet_survivors": stats["packet_survivors"]}}
    if split:  # the frame the ranks assembled against the same frame rendered by this rank alone (outside the timed region)
        r.render_device(cam, shader, o, fb.data_ptr(), stream)
        r.sync()
        out["gathered_frame"] = "ok" if np.array_equal(fb.cpu().numpy(), gpu_frame) else "DIFFERS"
        assert out["gathered_frame"] == "ok", "the frame gathered from the ranks differs from the frame one rank renders"
    # the breakdowns below describe one GPU rendering whole frames: rank 0 makes them for frames it rendered whole (no
    # collective in them; the other ranks wait at the next workload's first barrier)
    if light or split or (world > 1 and cams is not None):
        r.close()
        host.close()
        return out
    # ---- the same frames with the framebuffer left in HBM (no readback)
    dev_cams = cams if cams is not None else [cam]
    for c in dev_cams[:2]:
        run_frames(r, c, shader, o, fb, stream, 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rays_dev, stats = 0.0, None
    for _ in range(args.steps):
        for c in dev_cams:
            _, stats = run_frames(r, c, shader, o, fb, stream, 1)
            rays_dev += count_rays(stats)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t1
    out["value_device_resident"] = round(rays_dev / dt1 / 1e6, 2)
    out["ms_per_step_device_resident"] = round(dt1 * 1e3 / args.steps, 4)
    assert np.array_equal(fb.cpu().numpy(), gpu_frame), "rt355_render and rt355_render_device gave different frames"
    # ---- what this box's link to the host gives: the same frame copied from HBM to the page-locked pixmap and nothing
    # else (the floor of `ms_per_step` minus the rendering that the copy cannot overlap; boxes differ by a factor of 2)
    host_t = torch.from_numpy(host.array)
    for _ in range(3):
        host_t.copy_(fb, non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(10):
        host_t.copy_(fb, non_blocking=True)
    torch.cuda.synchronize()
    copy_ms = (time.perf_counter() - t2) * 1e2
    out["readback"] = {"bytes_per_frame": w * h * 3, "plain_copy_ms": round(copy_ms, 4),
                       "link_GBps": round(w * h * 3 / (copy_ms * 1e-3) / 1e9, 1)}
    n_rays_padded = ((w + 31) // 32) * ((h + 31) // 32) * 1024
    # ---- kernel-time breakdown: a few more frames, HIP-event times read back after each
    K = 3
    acc, stats = run_frames(r, dev_cams[-1], shader, o, fb, stream, K, detail=True)
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
        if stats["gather_rays"] > 0:  # one launch walks the scene table for every gather ray
            kern, avg_ms = "k_gather_walk", acc["search_ms"] / max(1, acc["search_launches"])
            avg_ms = max(r.search_launch_ms()) if acc["search_launches"] else avg_ms  # (the walk is the long one of the frame's search launches)
        else:  # a frame of several levels: its shading launches are the largest share
            kern, avg_ms = "k_shade_group", acc["shade_ms"] / max(1, acc["shade_launches"])
        out["roofline"] = measured_issue(name, kern, avg_ms) or {
            "bound": "valu_issue", "kernel": kern, "achieved": None, "peak": round(SIMDS * CLOCK_HZ / 4.0 / 1e12, 4), "unit": "T wave-instructions/s",
            "frac": None, "traffic": None, "avg_launch_ms": round(avg_ms, 4), "note": "no counter pass of this kernel in profiles/" + ISSUE_FILE}
    elif not searched:
        # k_render_direct: the reference's own algorithm (every primitive tested for the camera ray and for every
        # shadow ray, fp64), one thread per pixel.  Priced as the reference-formula flops of those tests against
        # the FP32 vector peak; the shading of each hit (normalisations, x^25, 8-bit colour arithmetic), which
        # the per-test figure does not count, is most of its instructions on a 4-object scene.
        flops = algorithmic_flops(scene, shader, cam, stats)
        avg_ms = acc["primary_ms"] / K
        tf = flops / (avg_ms * 1e-3) / 1e12
        refracts = bool((scene.pack().objects["fresnel"] < 1).any()) and not shading
        if refracts and scene.n_primitives >= 64:
            # every ray of the pixel's tree walks the scene table: the tests are those of the walk, not one per primitive
            out["roofline"] = measured_issue(name, "k_render_refract", avg_ms) or {
                "bound": "valu_issue", "kernel": "k_render_refract", "achieved": None, "peak": round(SIMDS * CLOCK_HZ / 4.0 / 1e12, 4),
                "unit": "T wave-instructions/s", "frac": None, "traffic": None, "avg_launch_ms": round(avg_ms, 4),
                "note": "no counter pass of this kernel in profiles/" + ISSUE_FILE}
            out["roofline"].update({"launches_per_frame": 1, "box_tests_per_frame": stats["pair_tests"], "candidates_evaluated_fp64": stats["exact_tests"]})
        else:
            out["roofline"] = {"bound": "valu_fp32", "kernel": "k_render_gather" if shading else ("k_render_refract" if refracts else "k_render_direct"), "achieved": round(tf, 3),
                               "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_TFLOPS, 5),
                               "traffic": None, "launches_per_frame": 1, "avg_launch_ms": round(avg_ms, 4),
                               "algorithmic_flops_per_frame": flops, "flops_per_test": ALGO_FLOPS,
                               "algorithmic_bytes_per_frame": float(3 * w * h),
                               "note": "arithmetic is fp64 (half the fp32 rate); per-hit shading is not in the flop count"
                                       + ("; nor are the tests of the mirror and refracted rays" if stats["secondary_rays"] else "")}
            iss = measured_issue(name, out["roofline"]["kernel"], avg_ms)  # what bounds it: the share of the vector-issue slots it uses
            if iss:
                out["roofline"]["valu_issue_frac"] = iss["frac"]
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
        for k, rf in rooflines.items():  # what the counters say bounds them: instruction issue, beside the HBM floor
            iss = measured_issue(name, k, rf["avg_launch_ms"])
            if iss:
                rf["valu_issue_frac"] = iss["frac"]
        dominant = max(rooflines, key=lambda k: times[k])
        out["roofline"] = rooflines[dominant]
        out["roofline_other_kernels"] = {k: v for k, v in rooflines.items() if k != dominant}
    traffic_name = {"mesh100k_sweep60": "mesh100k_2lights_4k"}.get(name, name)
    for rf in [out["roofline"], *out.get("roofline_other_kernels", {}).values()]:
        t = measured_traffic(traffic_name, rf["kernel"]) if (args.mode == "packet" and rf.get("frac") is not None) else None
        if t:
            rf["traffic"] = t["bytes"]
            rf["traffic_detail"] = t
    if args.mode == "packet" and not args.no_allpairs and searched and cams is None and scene.n_primitives < 500_000:
        o2 = B.opts(w, h, mode=B.MODE_ALLPAIRS, shading=shading)
        run_frames(r, cam, shader, o2, fb, stream, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        acc2, stats2 = run_frames(r, cam, shader, o2, fb, stream, 3, detail=True)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        assert np.array_equal(fb.cpu().numpy(), gpu_frame), "the all-pairs search gave a different frame"
        out["all_pairs"] = {"value_device_resident": round(count_rays(stats2) * 3 / dt2 / 1e6, 2), "unit": "Mrays/s", "steps": 3,
                            "ms_per_step": round(dt2 * 1e3 / 3, 4),
                            "search_ms_per_frame": round(acc2["search_ms"] / 3, 4),
                            "roofline": valu_roofline(acc2, stats2)}
    if not args.no_cpu_baseline:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        stride = args.cpu_stride
        if stride == 0:
            work = scene.n_primitives * (1 + len(shader.lights))
            stride = 1 if work < 200 else (4 if work < 5000 else (16 if work < 1_000_000 else 64))
        if shading:  # a gather shader tests every primitive for every direction of every hit
            stride = 4 if scene.n_primitives * len(dirs) < 5000 else 256
        out["cpu_baseline"] = cpu_baseline(scene, shader, dev_cams[-1], w, h, stride, threads, shading, dirs, gpu_frame)
    r.close()
    host.close()
    return out


def measure_inproc(name, args, devices):
    """One workload through ONE process driving len(devices) GPUs (rt355_scene_create_multi): a frame's 32-row tiles
    dealt to the GPUs, every GPU writing its rows into the one page-locked pixmap; a sweep's frames dealt round robin.
    The headline (whole frames, weak scaling) is a sweep of one frame per GPU.  Outside the timed region the result is
    compared with what one GPU renders alone."""
    import torch
    from raytracer_amd import binding as B
    from raytracer_amd.api import HostPixmap, Renderer
    scene, shader, cam, w, h = build_workload(name)
    shading, dirs = gather_of(name)
    cams = sweep_of(name, w, h)
    n = len(devices)
    weak = cams is None and DEFAULT_SHARD.get(name) is None  # whole frames, one per GPU and step
    if weak:
        cams = [cam] * n
    r = Renderer(scene, devices=devices)
    if dirs is not None:
        r.set_directions(dirs)
    if args.gather == "xgmi":
        r.set_option("gather", 1)
    shard = args.shard or DEFAULT_SHARD.get(name, "frames")
    if cams is None and shard == "balanced" and n > 1:
        r.set_option("balance", 1)  # contiguous slabs, cut by what the devices' shares took in the calls before
    o = B.opts(w, h, mode=B.MODE_PACKET if args.mode == "packet" else B.MODE_ALLPAIRS, shading=shading)
    pix = [HostPixmap(w, h) for _ in (cams or [None])]
    arrays = [p.array for p in pix]

    def step():
        if cams is not None:
            return r.render_sweep(cams, shader, o, arrays)
        return [r.render(cam, shader, w, h, mode=o.mode, shading=shading, out=arrays[0])[1]]
    balanced = cams is None and shard == "balanced" and n > 1
    for _ in range(max(args.warmup, 6) if balanced else args.warmup):  # (the cuts need a few frames to settle)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sts = step()
    dt = time.perf_counter() - t0
    rays = sum(count_rays(st) for st in sts)
    last = arrays[-1].copy()
    for p in pix:
        p.close()
    r.close()
    torch.cuda.set_device(devices[0])
    alone = Renderer(scene)
    if dirs is not None:
        alone.set_directions(dirs)
    want, _ = alone.render((cams or [cam])[-1], shader, w, h, mode=o.mode, shading=shading)
    alone.close()
    same = bool(np.array_equal(want, last))
    assert same, "the frame the GPUs of one process rendered together differs from the frame one GPU renders"
    return {"value": round(rays * args.steps / dt / 1e6, 2), "unit": "Mrays/s", "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 4), "scaling": "weak" if weak else "strong", "gathered_frame": "ok",
            "timed": "one process, rt355_scene_create_multi over %d devices: %s" % (n, "rt355_render_sweep" if cams is not None else "rt355_render"),
            "config": {"workload": name, "width": w, "height": h, "primitives": scene.n_primitives, "lights": len(shader.lights), "alias": 0,
                       "shard": "frames" if cams is not None else ("balanced" if shard == "balanced" and n > 1 else "tiles"),
                       "frames_per_step": len(cams) if cams is not None else 1, "search": args.mode,
                       "rays_per_frame": {"primary": sts[-1]["primary_rays"], "shadow": sts[-1]["shadow_rays"],
                                          "gather": sts[-1]["gather_rays"], "mirror": sts[-1]["secondary_rays"]}}}


def run_inproc(args):
    """`--inproc`: the headline and the scaling workloads through one process that drives all N GPUs."""
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    share = bool(os.environ.get("RT355_BENCH_SHARE_GPU"))
    devices = [0] * args.gpus if share else list(range(args.gpus))
    assert share or torch.cuda.device_count() >= args.gpus, f"--gpus {args.gpus} but {torch.cuda.device_count()} devices"
    torch.cuda.set_device(devices[0])
    out = {}
    names = [args.workload] + (SCALING if args.workload == HEADLINE and not args.no_scaling else [])
    for name in names:
        t0 = time.perf_counter()
        out[name] = measure_inproc(name, args, devices)
        print(f"bench: in-process {name} {out[name]['ms_per_step']} ms per step ({time.perf_counter() - t0:.1f} s)", file=sys.stderr, flush=True)
    head = out[args.workload]
    line = compact_line(head, {}, {k: v for k, v in out.items() if k != args.workload}, args.gpus, args.steps, args.warmup)
    line["launch"] = "one process, %d devices" % args.gpus
    emit(line, dict(line, headline=head, scaling_workloads_detail=out), detail_name="bench_detail_inproc.json")
    return 0


def inproc_beside(args, world):
    """After the ranks' own measurements: the same headline and scaling workloads by ONE process that drives all N GPUs
    (a child process: the ranks are idle meanwhile), so that one run shows both ways of reaching N GPUs.  A failure
    is reported in the record, not raised."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--inproc", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--gather", args.gather, "--mode", args.mode]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return {"error": "no result within 420 s"}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stderr or "no output")[-300:]}
    d = json.loads(lines[-1])
    out = {"ms": d["ms_per_step"], "value": d["value"], "scaling": d["scaling"]}
    for name, e in d.get("scaling_workloads", {}).items():
        out[name] = {"ms": e["ms"], "value": e["value"], "gathered_frame": e.get("gathered_frame")}
    return out


HEADLINE = "default_4k"
OTHERS = ["spheres10k_8lights_4k", "mesh100k_2lights_4k", "default_bleed162_4k",
          "spheres10k_mirrors_8lights_4k", "glass_room_4k", "spheres10k_glass_8lights_4k", "mesh100k_bleed42_4k"]
# measured after the headline at every N: one frame in interleaved tiles between the ranks (BASELINE configs[3] and
# configs[2]'s scene), and the camera sweep dealt round robin (configs[4]); at N = 1 the one GPU renders all of each
SCALING = ["mesh1m_8k_tiles", "spheres10k_tiles", "mesh100k_sweep60"]


def _num(x, digits=4):
    return None if x is None else round(float(x), digits)


def compact_roofline(rf):
    """The roofline object of the line: the contract's keys and the launch time they were measured with."""
    out = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")}
    if "valu_issue_frac" in rf:
        out["valu_issue_frac"] = rf["valu_issue_frac"]
    return out


def compact_cpu(cb):
    out = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
           "sample": cb.get("sample_short", cb.get("sample", ""))[:120], "parity_sample": cb.get("parity_sample")}
    if "one_thread" in cb:
        out["one_thread"] = _num(cb["one_thread"]["value"])
    if "reference_classes_one_thread" in cb:  # the reference's own classes (oracle/_ref), one thread
        out["reference_one_thread"] = _num(cb["reference_classes_one_thread"]["value"])
    return out


def compact_line(head, others, scaling, world, steps, warmup):
    """The ONE line the run ends with: BASELINE's metric on the headline workload with its roofline and CPU baseline,
    five numbers per other workload, the scaling workloads.  Everything else is in bench_detail.json."""
    out = {"metric": "Mrays/sec (primary+shadow) at %dx%d" % (head["config"]["width"], head["config"]["height"]),  # (the headline: 3840x2160)
           "value": head["value"], "unit": "Mrays/s",
           "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": head["ms_per_step"],
           "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None,
           "dtype": "f64" if head["config"]["primitives"] < 64 else "f32 search + f64 resolve", "data": "synthetic"}
    c = head["config"]
    out["config"] = {k: c[k] for k in ("workload", "width", "height", "primitives", "lights", "alias", "shard", "search",
                                       "frames_per_step", "rays_per_frame") if k in c}
    for k in ("gathered_frame", "value_device_resident", "ms_per_step_device_resident"):
        if k in head:
            out[k] = head[k]
    if "readback" in head:
        out["readback_link_GBps"] = head["readback"]["link_GBps"]
    if "roofline" in head:
        out["roofline"] = compact_roofline(head["roofline"])
    if "cpu_baseline" in head:
        out["cpu_baseline"] = compact_cpu(head["cpu_baseline"])
    if others:
        out["configs"] = {}
        for name, d in others.items():
            rf = d.get("roofline", {})
            e = {"ms": d["ms_per_step"], "value": d["value"], "kernel": rf.get("kernel"), "bound": rf.get("bound"), "frac": rf.get("frac"),
                 "parity": d.get("cpu_baseline", {}).get("parity_sample")}
            if "all_pairs" in d:  # the all-pairs search of the same frame: the FP32-VALU-bound kernel
                e["allpairs_fp32_frac"] = d["all_pairs"]["roofline"]["frac"]
            out["configs"][name] = e
    if scaling:
        out["scaling_workloads"] = {}
        for name, d in scaling.items():
            e = {"ms": d["ms_per_step"], "value": d["value"], "scaling": d["scaling"], "shard": d["config"]["shard"]}
            if "gathered_frame" in d:
                e["gathered_frame"] = d["gathered_frame"]
            out["scaling_workloads"][name] = e
    out["detail"] = "bench_detail.json"
    return out


# The line goes to the descriptor that was standard output when the process started; descriptor 1 itself points at
# standard error from then on, so that nothing a library prints (gloo reports its connections on standard output,
# in lines that begin with a bracket) can stand before or behind the line.
_LINE_FD = None


def own_stdout():
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def say_line(text: str):
    sys.stdout.flush()
    sys.stderr.flush()
    os.write(_LINE_FD if _LINE_FD is not None else 1, (text + "\n").encode())


def emit(line: dict, detail: dict, detail_name: str = "bench_detail.json"):
    """bench_detail.json first, then the line: the last thing the run writes, on stdout, with stderr flushed before it."""
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # (cannot happen with the fields above; the line must stay readable by a tail)
        line = {k: v for k, v in line.items() if k not in ("configs",)}
        text = json.dumps(line, separators=(",", ":"))
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, detail_name), "w") as f:
                    json.dump(detail, f, indent=1)
            except OSError:
                pass
    say_line(text)
    return text


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` started by hand: one child process per rank with the environment
    torch.distributed.run would give it.  The parent never touches the GPU; rank 0 prints the JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    # a rank that fails takes the others with it (they would wait for it in the next collective for ever)
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            break
        time.sleep(0.05)
    return max(abs(c) for c in codes)


def check_launch(args) -> int:
    """The N-rank start-up alone: every rank joins a gloo group from the environment it was given, the ranks
    are summed, rank 0 prints one line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    seen = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.init_process_group("gloo")
        dist.all_reduce(seen)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        say_line(json.dumps({"launch": "ok", "n_gpus": world, "rank_sum": float(seen.item())}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=HEADLINE)
    ap.add_argument("--shard", choices=["frames", "slabs", "tiles", "balanced"], default=None,
                    help="N > 1: whole frames per rank (default), or one frame in row slabs / interleaved 32-row tiles / slabs cut by the shares' times")
    ap.add_argument("--gather", choices=["host", "xgmi"], default="host",
                    help="--shard slabs|tiles: where the frame comes together -- host: every rank writes its rows into one "
                         "shared page-locked pixmap over its own link; xgmi: rows gathered on GPU 0 (RCCL) and read back there")
    ap.add_argument("--mode", choices=["packet", "allpairs"], default="packet",
                    help="search of searched scenes: packet = beams against a direction-sorted table "
                         "(library default); allpairs = fp32 test of every ray/primitive pair")
    ap.add_argument("--no-others", action="store_true", help="no other workloads after the headline (\"configs\" of the line)")
    ap.add_argument("--no-scaling", action="store_true", help="no scaling workloads after the headline (\"scaling_workloads\")")
    ap.add_argument("--no-allpairs", action="store_true", help="skip the all-pairs comparison run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=0, help="pixel stride of the CPU sample (0 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--in-flight", type=int, default=1,
                    help="sweep workloads: scene handles per rank, each sweeping its share of the frames from a host thread of its own "
                         "(one handle keeps two frames in flight by itself)")
    ap.add_argument("--inproc", action="store_true",
                    help="N > 1 through ONE process that drives all N GPUs (rt355_scene_create_multi) instead of one rank per GPU")
    ap.add_argument("--no-inproc", action="store_true", help="N > 1: do not measure the one-process way beside the ranks' own")
    ap.add_argument("--check-launch", action="store_true",
                    help="start the ranks, meet over gloo, print {\"launch\": \"ok\", ...} and stop: no GPU is touched")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.inproc:
        sys.exit(spawn_ranks(args.gpus))  # (before anything in this process touches the GPU; the ranks inherit standard output)
    own_stdout()
    if args.inproc:
        sys.exit(run_inproc(args))
    if args.check_launch:
        sys.exit(check_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # rehearsal of the N > 1 path on a one-GPU box: every rank on device 0, collectives over gloo (RCCL refuses
    # two ranks on one device)
    share = bool(os.environ.get("RT355_BENCH_SHARE_GPU"))
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The control plane (barriers, the max-over-ranks time, the ray counts: a few numbers in host memory) runs over
        # gloo; RCCL carries GPU tensors and is set up only where there are any to send: the gather of a frame's rows
        # on GPU 0 (--gather xgmi).  The default path (whole frames, or rows into the one host pixmap) has no GPU collective.
        dist.init_process_group("gloo")
    ctx = {"dist": dist, "world": world, "rank": rank, "reduce_device": "cpu", "gpu_group": None}
    if world > 1 and args.gather == "xgmi" and not share:
        ctx["gpu_group"] = dist.new_group(backend="nccl")  # RCCL over xGMI: the gather of the final frame on GPU 0

    def timed_measure(name, light=False):  # (a line per workload on stderr: a long default run shows that it is alive)
        t0 = time.perf_counter()
        res = measure(name, args, ctx, light=light)
        if rank == 0:
            print(f"bench: {name} {res['ms_per_step']} ms per step ({time.perf_counter() - t0:.1f} s)", file=sys.stderr, flush=True)
        return res

    head = timed_measure(args.workload)
    scaling, others = {}, {}
    if args.workload == HEADLINE and not args.no_scaling:
        for name in SCALING:
            scaling[name] = timed_measure(name, light=True)
    if world == 1 and args.workload == HEADLINE and not args.no_others:
        for name in OTHERS:
            others[name] = timed_measure(name)
    if rank == 0:
        # (a workload that casts gather rays counts them too)
        line = compact_line(head, others, scaling, world, args.steps, args.warmup)
        if world > 1 and args.workload == HEADLINE and not args.no_inproc and not args.no_scaling:
            line["inproc"] = inproc_beside(args, world)
        detail = dict(line, headline=head, other_configs=others, scaling_workloads_detail=scaling)
        # (a run of one workload alone -- the profiling passes -- keeps its record apart from the default run's)
        full = args.workload == HEADLINE and (others or scaling)
        if not full:
            line["detail"] = f"bench_detail_{args.workload}.json"
        emit(line, detail, detail_name=line["detail"])
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""More seeds of the randomised GPU parity tests than the suite runs (GPU box):
    python tools/parity_sweep.py FIRST_SEED N_SEEDS [family ...]
Every family is one of the seed-parametrised tests of tests/test_gpu_parity.py, called with other seeds; a
failure is printed with its family and seed and the sweep goes on.  Progress goes to stdout line by line."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    import torch
    torch.cuda.set_device(0)
    from raytracer_amd import api
    api.lib()
    from oracle.oracle_py import Oracle
    import test_gpu_parity as T
    families = {"search": T.test_random_scenes_search_is_conservative, "mirrors": T.test_random_scenes_with_mirrors,
                "environment": T.test_random_scenes_inside_an_environment, "quadrics": T.test_random_scenes_with_quadrics,
                "glass": T.test_random_glass_in_a_room, "textures": T.test_random_scenes_with_textures, "gather": T.test_random_scenes_gather,
                "glass_spheres": T.glass_among_spheres_seeded}
    chosen = sys.argv[3:] or list(families)
    oracle = Oracle()
    failed = []
    for name in chosen:
        for seed in range(first, first + n):
            t0 = time.time()
            try:
                families[name](api, oracle, seed)
                print(f"{name} {seed} ok {time.time() - t0:.1f}s", flush=True)
            except Exception as ex:  # noqa: BLE001 -- a sweep reports and goes on
                failed.append((name, seed))
                print(f"{name} {seed} FAILED: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
                traceback.print_exc(limit=3)
    print("failed:", failed, flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())

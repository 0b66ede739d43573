"""GPU box: the N > 1 path of bench.py end to end with two ranks on the one GPU of the box (collectives over gloo,
since RCCL refuses two ranks on one device): ranks spawned by bench.py itself, every rank driving the HIP library,
the barrier / max-over-ranks timing / whole-job ray counts, and for a frame shared between the ranks the one gather,
whose result the benchmark compares with the frame one rank renders alone."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ, RT355_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    # (a session of its own: should it hang, the launcher AND its ranks are killed, not the launcher alone)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--steps", "3", "--warmup", "2"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.fail("bench.py " + " ".join(args) + " did not end in 420 s:\n" + err[-2000:])
    assert p.returncode == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_whole_frames():
    one = _bench("--gpus", "1", "--no-others", "--no-scaling", "--no-allpairs", "--no-cpu-baseline")
    two = _bench("--gpus", "2")
    # the headline keeps its roofline and CPU baseline at N > 1 (rank 0's, for frames it rendered whole)
    assert two["roofline"]["frac"] > 0 and two["cpu_baseline"]["value"] > 0 and two["cpu_baseline"]["parity_sample"] == "ok"
    # the plain `--gpus N` run measures the scaling workloads: the 8K and 4K frames in balanced slabs between the ranks, the sweep
    sw = two["scaling_workloads"]
    assert set(sw) == {"mesh1m_8k_tiles", "spheres10k_tiles", "mesh100k_sweep60"}
    for name in ("mesh1m_8k_tiles", "spheres10k_tiles"):
        assert sw[name]["gathered_frame"] == "ok" and sw[name]["scaling"] == "strong" and sw[name]["shard"] == "balanced"
    assert all(v["ms"] > 0 and v["value"] > 0 for v in sw.values())
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["workload"] == one["config"]["workload"]
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    # value counts the rays of both ranks over the slower rank's time
    rays = sum(one["config"]["rays_per_frame"].values())
    assert abs(two["value"] * two["ms_per_step"] * 1e3 - 2 * rays) < 1e-3 * rays


@pytest.mark.parametrize("gather", ["host", "xgmi"])
@pytest.mark.parametrize("shard", ["slabs", "tiles"])
def test_two_ranks_share_a_frame(shard, gather):
    """gather = host: every rank's rt355_render writes its rows into one page-locked pixmap the ranks share;
    xgmi: the rows are gathered on rank 0's GPU and read back there."""
    d = _bench("--gpus", "2", "--shard", shard, "--gather", gather, "--workload", "spheres10k_8lights_4k")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shard"] == shard
    assert d["gathered_frame"] == "ok"
    r = d["config"]["rays_per_frame"]
    assert r["primary"] == 3840 * 2160 and r["shadow"] == 35244679  # both ranks' shares, summed


def test_two_ranks_sweep():
    d = _bench("--gpus", "2", "--workload", "mesh100k_sweep60")
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 60 and d["value"] > 0 and d["scaling"] == "strong"

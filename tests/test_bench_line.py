"""bench.py's final line: formatted from synthetic results of the size a default run produces, it stays a few
kilobytes, parses, and carries the contract's keys with `roofline` and `cpu_baseline` on the headline."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _result(name, searched=True, world=1):
    long_text = "x" * 400  # (the detail record carries prose; none of it may reach the line)
    d = {"value": 31234.56, "unit": "Mrays/s", "steps": 20, "warmup": 5, "ms_per_step": 0.5381, "timed": long_text,
         "scaling": "weak",
         "config": {"workload": name, "width": 3840, "height": 2160, "primitives": 100353, "lights": 2, "alias": 0, "shard": "frames",
                    "frames_per_step": 1, "frames_in_flight": 1, "search": "packet",
                    "rays_per_frame": {"primary": 8294400, "shadow": 8221013, "gather": 0, "mirror": 0},
                    "shading": "Shader::shading", "gather_directions": 0, "candidates_evaluated_fp64": 222468344,
                    "overflow_rays": 0, "packet_survivors": 417368},
         "value_device_resident": 55200.1, "ms_per_step_device_resident": 0.2991,
         "readback": {"bytes_per_frame": 24883200, "plain_copy_ms": 0.4469, "link_GBps": 55.7},
         "kernel_time": {"search_ms_per_frame": 0.0, "resolve_primary_ms_per_frame": 0.2591, "shade_ms_per_frame": 0.0, "frames_measured": 3},
         "roofline": {"bound": "hbm" if searched else "valu_fp32", "kernel": "k_shade_group" if searched else "k_render_direct",
                      "achieved": 875.3, "peak": 8000.0, "unit": "GB/s", "frac": 0.1094, "traffic": 1.33e9, "avg_launch_ms": 1.2281,
                      "launches_per_frame": 1, "algorithmic_bytes_per_frame": 1.075e9, "note": long_text,
                      "traffic_detail": {"bytes": 1.33e9, "fetch_kib_counter": 1.0, "write_kib_counter": 2.0, "source": long_text},
                      "valu_issue_frac": 0.55},
         "roofline_other_kernels": {"k_resolve_primary": {"note": long_text}, "k_search_packet": {"note": long_text}},
         "all_pairs": {"value_device_resident": 1333.3, "ms_per_step": 64.4, "roofline": {"frac": 0.6103, "note": long_text}},
         "cpu_baseline": {"value": 181.23456789, "unit": "Mrays/s", "cores": 16, "kind": "port", "parity_sample": "ok",
                          "sample": long_text, "sample_short": "every 16th pixel in x and y of the same frame, 1.84 s",
                          "one_thread": {"value": 11.6123456, "unit": "Mrays/s", "sample": long_text},
                          "reference_classes_one_thread": {"value": 8.4123456, "unit": "Mrays/s", "kind": "reference", "sample": long_text}}}
    if world > 1:
        d["gathered_frame"] = "ok"
        d["scaling"] = "strong"
    return d


def test_line_is_compact_and_complete(capfd):
    head = _result("default_4k", searched=False)
    others = {n: _result(n) for n in bench.OTHERS}
    scaling = {n: _result(n, world=8) for n in bench.SCALING}
    line = bench.compact_line(head, others, scaling, 1, 20, 5)
    capfd.readouterr()
    text = bench.emit(line, dict(line, headline=head, other_configs=others))  # (written to the descriptor itself)
    printed = capfd.readouterr().out
    assert printed.endswith(text + "\n") and printed.count("\n") == 1 and printed.startswith("{")
    assert len(text) < 4096 < bench.LINE_LIMIT
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["scaling"] in ("weak", "strong") and d["config"]["workload"] == "default_4k" and "model" not in d["config"]
    rf = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and rf["frac"] == 0.1094
    cb = d["cpu_baseline"]
    assert cb["value"] == 181.2346 and cb["cores"] == 16 and cb["kind"] == "port" and len(cb["sample"]) < 130
    assert cb["reference_one_thread"] == 8.4123
    assert set(d["configs"]) == set(bench.OTHERS)
    for e in d["configs"].values():
        assert set(("ms", "value", "kernel", "frac", "parity")) <= set(e)
    assert set(d["scaling_workloads"]) == set(bench.SCALING)
    assert d["scaling_workloads"]["mesh1m_8k_tiles"]["gathered_frame"] == "ok"
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert detail["headline"]["roofline"]["note"] and set(detail["other_configs"]) == set(bench.OTHERS)
    os.remove(os.path.join(ROOT, "bench_detail.json"))


def test_nothing_else_reaches_standard_output(tmp_path):
    """Whatever a library prints on standard output while the bench runs (gloo reports its connections there, in lines
    that begin with a bracket) goes to standard error: the line is all that the original descriptor receives."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.own_stdout(); "
            "os.write(1, b'[Gloo] Rank 0 is connected\\n'); print('[library] chatter'); "
            "bench.say_line('{\"ok\": 1}'); print('more chatter')" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout == '{"ok": 1}\n'
    assert "[Gloo] Rank 0" in p.stderr and "chatter" in p.stderr


def test_issue_fraction_from_the_committed_counters(tmp_path, monkeypatch):
    """valu_issue: wave-level vector instructions x 4 cycles over the cycles 1 024 SIMDs have in the launch."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / bench.ISSUE_FILE).write_text(json.dumps({"w": {"k": {"valu_inst_per_launch": 1.024e9, "avg_ms": 2.0}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rf = bench.measured_issue("w", "k", 2.0)
    # 1.024e9 instructions x 4 cycles / (1024 SIMDs x 2.4e9 Hz x 2 ms) = 0.8333
    assert rf["bound"] == "valu_issue" and abs(rf["frac"] - 1.024e9 * 4 / (1024 * 2.4e9 * 2e-3)) < 1e-3
    assert bench.measured_issue("w", "other", 2.0) is None and bench.measured_issue("w", "k", 0.0) is None

"""The bench.py contract on the GPU box: one JSON line with the metric, the `roofline` and
`cpu_baseline` objects and the host-I/O leg, from a small configuration (Params(64)) so that it
finishes in seconds.  Run with `pytest -m gpu`."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "params64",
                        "--batch", "256", "--steps", "2", "--warmup", "1", *extra],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--chunk", "64")                       # four chunks: two lanes, as at the default workload
    assert d["metric"] == "bootstraps/sec" and d["unit"] == "bootstraps/sec"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "i32" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # `frac` is the whole iteration (both launches of the k-loop), the dominant kernel alone is
    # under kernels.k_extprod; whole_job_frac is the driver-timed figure
    k = r["kernels"]
    for name in ("k_extprod", "k_crt_lean"):
        assert k[name]["launch_samples"] > 0 and k[name]["launch_ms"] > 0
        assert k[name]["launch_ms_alone"] > 0 and k[name]["kernel_frac"] > 0
    # the iteration time is the device wall time of a step / chunk-iterations in it, not a sum of
    # kernel durations (with two lanes the kernels of two chunks overlap)
    assert d["config"]["lanes"] == 2 and "overlap" in r["kernel"]
    assert k["k_extprod"]["launch_ms_is"].startswith("overlapped")
    assert r["launch_ms"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["whole_job_frac"] > 0
    assert "HIP events" in r["launch_ms_source"]
    # roofline.traffic is measured in the run itself (rocprofv3 --pmc child passes on this box): the HBM
    # bytes of both launches of an iteration, at least what the kernels must move, with where it came from
    assert r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    assert abs(r["traffic_ratio"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 1e-9
    assert r["traffic_note"].startswith("measured in this run")
    assert k["k_extprod"]["traffic"] > 0 and k["k_crt_lean"]["traffic"] > 0
    assert k["k_extprod"]["valu_insts_per_launch"] > 0
    assert r["launch_ms_rocprof"] is None            # (the committed rocprofv3 trace is Params(1024)'s)
    assert d["config"]["build_id"]
    # one process, no launcher: no process group, so no proof-of-ranks figure to give (N > 1: test_gpu_multi.py,
    # tests/test_distributed_cpu.py)
    assert d["config"]["group_ranks"] is None and d["config"]["rccl_ranks"] is None
    assert d["config"]["key_broadcast_gbs"] is None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores_available"] >= c["cores"] >= 1
    # the whole k-loop is timed where that takes under 30 s (Params(64): all 64 iterations, no scaling)
    assert c["iterations_timed"] == c["iterations_total"] == 64 and "scaled" not in c["sample"]
    assert "64 of 64" in c["sample"] and c["opt"]["iterations_timed"] == 64
    assert c["opt"]["value"] > c["value"]            # 6 NTTs per iteration against 24
    if c["cores_available"] >= 16 and not c["cgroup_cpu_quota"]:
        assert c["cores"] == c["cores_available"]    # the box's CPU rate ...
        sh = c["share"]                               # ... and one GPU's share of it
        assert sh["cores"] == c["cores_available"] // 8 and sh["value"] > 0 and sh["opt"]["value"] > 0
    assert k["k_extprod"]["name"].startswith("k_extprod<9, 3") and k["k_crt_lean"]["name"] == "k_crt_lean<4, 3>"
    # the metric as SURVEY.md 8(d) words it (host buffers in and out): the same number of steps as the
    # headline after a warm-up of its own, the same bytes
    h = d["host_io"]
    assert h["value"] > 0 and h["equals_device_resident_output"] is True
    assert h["steps"] == d["steps"] and h["warmup"] >= 1
    assert abs(h["value"] - 256 / (h["ms_per_step"] * 1e-3)) < 1e-6 * h["value"]
    assert abs(h["vs_device_resident"] - h["device_resident_ms_per_step_alternating"] / h["ms_per_step"]) < 1e-9


def test_host_buffers_run_at_the_device_resident_rate():
    """with the copies pipelined chunk by chunk beside the kernels, the headline
    workload (Params(1024), batch 4096) through host pointers -- sgfhe_bootstrap_batch, H2D and D2H
    inside the timed region -- gives the bytes of the device-resident entry point at its rate, steps of
    the two alternating in one process so that clock drift falls on both.
    The suite holds the FUNCTIONAL half (same bytes, same steps) and a bound a real regression breaks
    (round 3's whole-buffer copies ran at 0.989 of the device-resident rate, unpipelined pageable copies at
    0.95): one attempt, no retry.  The rate itself -- 0.996-0.998 over 20
    alternating steps -- is bench.py's `host_io.vs_device_resident` in every committed bench line and
    tools/io_variants.py (profiles/r04_exp_io_variants.txt): three steps here cannot resolve 0.3 %."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-isolated", "--no-live-counters"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    h = d["host_io"]
    assert h["equals_device_resident_output"] is True and h["steps"] == 3
    assert h["vs_device_resident"] > 0.97, h
    # the headline workload's line: VALU issue against the kernel's own instruction mix AND against the raw
    # lane rate of the device whenever the counters of this build are at hand
    v = d["roofline"]["valu"]
    if v is not None:
        assert v["raw_peak"] > v["peak"] and 0 < v["raw_frac"] < v["frac"] < 1
        assert abs(v["raw_peak"] - 256 * 4 * 32 * 2.4e9 / 1e12) < 1e-6 and "nominal" in v["raw_peak_is"]


def test_bench_flags():
    d = _run("--no-cpu-baseline", "--no-host-io", "--flatten", "random", "--lanes", "1", "--chunk", "64")
    assert d["roofline"]["traffic"] is None and "counters" in d["roofline"]["traffic_note"]
    assert "cpu_baseline" not in d and "host_io" not in d
    assert "random flatten" in d["config"]["workload"] and d["value"] > 0
    assert d["config"]["lanes"] == 1 and d["config"]["chunk"] == 64
    k = d["roofline"]["kernels"]
    assert k["k_extprod"]["launch_ms_is"] == "alone on the device"
    assert k["k_extprod"]["launch_ms_alone"] == k["k_extprod"]["launch_ms"]
    # the CRT kernel is reported under the name the engine launches in this mode
    assert "k_crt_lean" not in k and k["k_crt_lean_rnd"]["name"] == "k_crt_lean_rnd<4, 3, false>"

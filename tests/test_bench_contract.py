"""The bench line committed with the latest profile carries every key of the driver's contract (task description:
bench.py contract + `roofline` + `cpu_baseline`), and the numbers in it are consistent with each other."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_bench_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_c3.json")))
    assert files, "no committed bench line under profiles/"
    text = open(files[-1]).read().strip().splitlines()[-1]
    return files[-1], json.loads(text)


def test_committed_bench_line_has_the_contract_keys():
    path, d = latest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (path, k)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]


def test_committed_bench_line_is_self_consistent():
    _, d = latest_bench_line()
    voxels = d["config"]["voxels"]
    assert abs(d["value"] - voxels / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3   # Mvoxels/s from the frame time
    r = d["roofline"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-2
    # a roofline fraction is bytes moved / time / peak: never above 1, for none of the reported launches
    assert 0 < r["frac"] <= 1.0
    for k in ("full_evaluation", "dense_case"):
        if k in r:
            assert 0 < r[k]["frac"] <= 1.0
            assert abs(r[k]["achieved"] - r[k]["bytes_per_launch"] / (r[k]["avg_launch_ms"] * 1e-3) / 1e9) / r[k]["achieved"] < 1e-2
    if "dense_case" in r:   # the dense map is where the layout's bytes and SURVEY 8(d)'s 80 B/voxel nearly coincide
        assert r["dense_case"]["bytes_per_launch"] <= 1.2 * r["effective"]["bytes_per_launch"]
    if r["traffic"] is not None:   # PMC traffic of the in-frame launches is near what the layout says they have to move
        assert 0.5 * r["bytes_per_launch"] <= r["traffic"] <= 3 * r["bytes_per_launch"]
    assert d["value"] / d["cpu_baseline"]["value"] >= 50    # BASELINE.json: >= 50x the single-thread CPU frame time at C3
    if "stress" in d and "x_cpu" in d["stress"]:
        assert d["stress"]["x_cpu"] >= 50


import pytest  # noqa: E402


def _run_bench(args, env_extra, timeout=120):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus_n_without_a_launcher_spawns_its_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it (how the driver starts N = 1; its first multi-GPU run must not
    die on a launch convention): bench.py starts its two ranks itself, each with the launcher's environment, and only
    rank 0's line reaches stdout.  The ranks here are stand-ins that report their environment (no GPU in this test)."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"SDM_BENCH_FAKE_RANK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.strip().splitlines() if x.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["LOCAL_RANK"] == "0" and d["WORLD_SIZE"] == "2" and d["MASTER_ADDR"] == "127.0.0.1" and int(d["MASTER_PORT"]) > 0


def test_bench_self_spawn_propagates_a_dead_rank():
    """rank 1 dies: the launcher returns its exit code and takes rank 0 (which would wait in the rendezvous) down with it"""
    import time
    t0 = time.time()
    r = _run_bench(["--gpus", "2"], {"SDM_BENCH_FAKE_RANK": "fail1"})
    assert r.returncode == 3 and time.time() - t0 < 25, (r.returncode, time.time() - t0)
    assert not r.stdout.strip()


def test_bench_gpus_n_under_a_launcher_with_another_world_size_says_so():
    r = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "SDM_BENCH_FAKE_RANK": ""})
    assert r.returncode != 0 and "must agree" in r.stderr


@pytest.mark.gpu
def test_sharded_bench_line_rehearsal():
    """The N > 1 path of bench.py - gloo process group, RCCL communicator inside the library, sdm_update_sharded, the
    collective timers, the strong-scaling leg - with the one rank a one-GPU box allows (SDM_BENCH_SHARDED=1): the line the
    driver's multi-GPU run will print has to come out whole."""
    import subprocess
    import sys
    env = dict(os.environ, SDM_BENCH_SHARDED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "3"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [x for x in r.stdout.strip().splitlines() if x.strip()]
    assert len(lines) == 1, "stdout must carry the one JSON line and nothing else (RCCL's banner goes to stderr): %r" % lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "collectives_us", "strong_scaling"):
        assert k in d, k
    c = d["collectives_us"]
    assert set(c["us"]) == {"counts_allgather", "halo_alltoall", "ck_alltoall", "ck_allgather"}
    assert c["us"]["ck_alltoall"] > 0 and c["us"]["ck_allgather"] > 0 and c["us_on_critical_path"] > 0
    assert d["issue_mode"].startswith("sharded") and d["cpu_baseline"] is None and d["n_gpus"] == 1

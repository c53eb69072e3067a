"""The driver's bench contract: `python bench.py` prints exactly one JSON line with the
agreed keys (metric/value/unit/..., plus `roofline` and `cpu_baseline`)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "96", "--warmup", "16",
                          "--no-variants", "--cpu-seconds", "0.5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 96 and d["warmup"] == 16
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "pairs/s" and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    # value and ms_per_step describe the same run
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "pairs/s" and c["value"] > 0


def test_bench_refuses_without_gpu_or_falls_loudly():
    """No silent CPU path: on a box without a GPU bench.py exits with an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert "no CPU fallback" in (out.stderr + out.stdout)


def test_bench_gpus_n_as_typed_starts_its_own_ranks_and_relays_failure():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the rank processes itself
    (tools/caffe.cpp:154-227 starts its per-GPU workers from one command too).  Without a GPU every rank refuses
    loudly and the parent exits with a rank's return code instead of asking for torch.distributed.run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_two_ranks_as_typed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--steps", "4", "--warmup", "1", "--no-variants"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert out.stderr.count("no CPU fallback") == 2, out.stderr[-2000:]      # one refusal per rank
    assert "torch.distributed.run" not in out.stderr
    assert out.stdout.strip() == ""


def test_launch_ranks_sets_the_rendezvous_environment(tmp_path):
    """The launcher's contract, without a GPU: N children, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rank 0's stdout
    relayed, worst return code returned."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\n"
                     "r = int(os.environ['RANK'])\n"
                     "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                     "assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['WORLD_SIZE'] == '3'\n"
                     "print('rank', r, sys.argv[1:])\n"
                     "sys.exit(7 if r == 2 else 0)\n")
    real = bench.__file__
    bench.__file__ = str(probe)
    try:
        import io
        import contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            args = type("A", (), {"gpus": 3})()
            rc = bench.launch_ranks(args, ["--gpus", "3", "--steps", "4"])
    finally:
        bench.__file__ = real
    assert rc == 7
    assert buf.getvalue().strip() == "rank 0 ['--gpus', '3', '--steps', '4']"


@pytest.mark.gpu
def test_bench_two_ranks_as_typed():
    """`python bench.py --gpus 2 --backend gloo ...` as typed on ONE GPU: two rank processes share the card
    (rehearsal of the N > 1 control flow), one JSON line comes back through the parent."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["steps"] == 4 and d["warmup"] == 1
    assert abs(d["value"] - 2 * 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    sv = d["config"]["strong_split_variant"]
    assert sv["scaling"] == "strong" and sv["pairs_per_gpu"] == 2048 and sv["value"] > 0

"""The driver's bench contract, rehearsed on CPU: bench.py is launched exactly as the driver launches it (plain for
N = 1, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` for N > 1) with
`--emulate --backend gloo`, i.e. the real kernels on the CPU emulation of the HIP runtime and gloo instead of RCCL.
Checks arguments, environment handling, the one-JSON-line stdout contract and the fields the judge reads; the
numbers themselves mean nothing here."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(cmd, detail=None):
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    if detail:
        cmd = cmd + ["--detail", str(detail)]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout            # stdout carries the JSON line and nothing else
    assert p.stdout.rstrip("\n").splitlines()[-1] == lines[0]          # ... and it is the LAST line
    assert len(lines[0]) < 4096, len(lines[0])  # the driver stopped parsing the line when it outgrew ~20 KB (round 4)
    return json.loads(lines[0])


def test_single_rank_line_has_parity_and_cpu_baseline(emu_library, tmp_path):
    detail = tmp_path / "detail.json"
    d = run([sys.executable, "bench.py", "--emulate", "--logn", "13", "--rows", "12", "--steps", "2", "--warmup", "1"], detail)
    full = json.loads(detail.read_text())
    for k in CONTRACT + ["parity", "cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["higher_is_better"] is True
    par = d["parity"]
    assert par["rows_checked"] == 12 and par["ok"] and par["max_row_err"] < 1e-8          # bench.py times the 1e-9 target
    assert "per_kernel_class" not in par          # tables live in the detail file, not in the contract line
    assert sum(c["rows"] for c in full["parity"]["per_kernel_class"].values()) == 12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert d["cpu_baseline"]["value"] > 0 and isinstance(d["cpu_baseline"]["reference_mounted"], bool)
    assert full["cpu_baseline"]["reference_as_is"]["kind"] in ("reference", "port")
    assert full["value"] == pytest.approx(d["value"], rel=1e-4) and "per_class" in full["roofline"]
    assert "per_class" not in d["roofline"] and "kernels" not in d["roofline"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 * r["frac"] + 1e-12
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.parametrize("world", [2])
def test_multi_rank_launch_as_the_driver_does(emu_library, world, tmp_path):
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
             "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", str(world),
             "--steps", "2", "--warmup", "1", "--emulate", "--backend", "gloo", "--logn", "13", "--rows", "12"], tmp_path / "d.json")
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["rows_total"] == 12 and d["config"]["rows_per_gpu"] == 6
    assert d["weak_scaling"]["rows_total"] == 12 * world and d["weak_scaling"]["rows_per_gpu"] == 12
    assert "parity" not in d and "cpu_baseline" not in d          # rank 0 at N = 1 only
    assert d["value"] > 0 and d["weak_scaling"]["value"] > 0
    # the same steps through the public entry point (pycwt_amd.parallel.cwt_sharded): timed, one collective per call
    assert d["api_ms_per_step"] > 0 and d["api_collectives_per_call"] == 1
    full = json.loads((tmp_path / "d.json").read_text())
    assert full["api"]["rows_this_rank"] >= 1 and "cwt_sharded" in full["api"]["entry_point"]


def test_compact_line_of_a_full_default_run_stays_small():
    """The compact line built from the LARGEST detail dictionary the bench has produced (round 4's default run on the GPU,
    six nested workloads) is < 4 KB and keeps roofline + cpu_baseline + parity."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    line = json.dumps(bench.compact_line(full, os.path.join(ROOT, "bench_detail.json")))
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in CONTRACT + ["parity", "cpu_baseline", "extra", "from_idle"]:
        assert k in d, k
    assert d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert {"c2_roundoff_ms", "c3_paul_gs", "c3_dog_gs", "c4_ms"} <= set(d["extra"])


def test_plain_multi_gpu_command_spawns_its_own_ranks(emu_library, tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's single-GPU command): bench.py starts its
    ranks itself through torch.distributed.run on the loopback address and still prints exactly one JSON line."""
    env_keys = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        d = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--emulate", "--backend", "gloo",
                 "--logn", "13", "--rows", "12"], tmp_path / "d.json")
    finally:
        os.environ.update(saved)
    assert d["n_gpus"] == 2 and d["config"]["rows_per_gpu"] == 6 and d["value"] > 0

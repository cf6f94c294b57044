"""-m gpu: bench.py's N-GPU code path rehearsed on ONE GPU (`--rehearse-exchange`: process group, the C-ABI communicator built on a helper
thread with a host-waited trial exchange, the three exchange modes, the backtest_only figures) and its default one-GPU line, on a small
workload -- a one-GPU box cannot run N > 1, and the first 8-GPU run must not be the first time this code executes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = Path(__file__).resolve().parent.parent


def _bench(*argv):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return lines[0]


def test_bench_line_small_workload():
    d = _bench("--symbols", "320", "--days", "512", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary")
    assert d["n_gpus"] == 1 and d["collective"] is None and d["value"] > 0 and d["dtype"] == "f64" and d["unit"] == "rows/s"
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0 and d["config"]["row_pitch_elements"] == 512


def test_bench_rehearses_the_n_gpu_path_with_a_world_of_one():
    d = _bench("--rehearse-exchange", "--symbols", "320", "--days", "512", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    c = d["collective"]
    assert c["world_size_seen"] == 1 and c["torch_backend"] == "nccl" and c["c_abi_equals_torch_gather"] is True
    assert "C ABI" in c["timed"], c["timed"]                       # the product's own collective was built and timed, not the fallback
    sm = c["step_ms"]
    assert sm["chosen"] in ("serial", "overlapped") and all(sm[k] > 0 for k in ("serial_ms_per_step", "overlapped_ms_per_step", "kernel_only_ms_per_step"))
    b = d["backtest_only"]
    assert b["chosen"] in ("serial", "overlapped") and b["kernel_only_ms_per_step"] > 0 and b["two_steps_in_flight_same_summary"] is True
    assert isinstance(b["two_steps_in_flight_ms_per_step"], float)
    assert d["weak_scaling"]["symbols_per_gpu"] == 320 and d["weak_scaling"]["ms_per_step"] > 0

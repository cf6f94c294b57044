"""The committed measurement evidence belongs to the committed kernels (round-3 review: a stale profile contradicted the text).
CPU only: nothing here runs a kernel."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_pmc_traffic_file_was_collected_on_these_kernel_sources():
    """bench.py quotes profiles/<round>_pmc_traffic.json only when its source hash equals the hash of csrc/*.hip + csrc/*.h; a kernel edit
    without a new `scripts/collect_profiles.sh` run makes the line say `traffic: null` -- and this test fail"""
    import pytest
    from bench import PMC_FILE, source_hash
    if not PMC_FILE.exists():
        pytest.skip(f"{PMC_FILE.name}: this round's counters are not collected yet (bench.py then prints traffic: null)")
    pm = json.loads(PMC_FILE.read_text())
    assert pm["source_hash"] == source_hash(), "re-run scripts/collect_profiles.sh on the GPU box and copy gpurun_out/prof/* to profiles/<ROUND>_*"
    assert pm["launches_per_step"].get("seq_jobs_kernel<0>") == 2 and "cdl_all_kernel<true, true>" in pm["kernels"]
    step = sum(v["hbm_bytes_per_launch"] * pm["launches_per_step"].get(k, 1) for k, v in pm["kernels"].items() if not k.startswith(("at::", "__amd")))
    assert 14.5e9 < step < 16e9, step      # the step moves ~15.1 GB


def test_committed_kernel_resources_are_the_builds():
    """profiles/<round>_kernel_resources.txt is csrc/suite.resources.txt of the build (written by the Makefile on every build of suite.hip)"""
    cur = ROOT / "polars_quant_amd" / "csrc" / "suite.resources.txt"
    from bench import ROUND
    com = ROOT / "profiles" / f"{ROUND}_kernel_resources.txt"
    import pytest
    if not com.exists():
        pytest.skip(f"{com.name}: this round's evidence is not collected yet")
    if not cur.exists():
        pytest.skip("suite.resources.txt is written when suite.hip is compiled")

    def light(text):
        m = re.search(r"seq_jobs_kernelILi0E.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+)", text, re.S)
        return tuple(int(x) for x in m.groups())
    assert light(cur.read_text()) == light(com.read_text()) == (192, 0, 0)


def test_bench_line_of_the_committed_profile_states_its_denominators():
    import pytest
    from bench import ROUND
    f = ROOT / "profiles" / f"{ROUND}_bench.json"
    if not f.exists():
        pytest.skip(f"{f.name}: this round's evidence is not collected yet")
    line = json.loads(f.read_text().strip().splitlines()[-1])
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["fused_lower_bound_bytes_per_row"] == 928
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] and abs(r["step_frac_counter_bytes"] - r["step_traffic"] / (line["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    assert r["step_frac_fused_floor"] < r["step_frac_counter_bytes"] < line["config"]["suite_frac_of_hbm_peak_on_per_call_bytes"]
    assert line["cpu_baseline"]["kind"] == "port" and "5000 symbols" in line["cpu_baseline"]["sample"] and "-march=native" in line["cpu_baseline"]["sample"]

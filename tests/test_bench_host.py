"""Host-side pieces of bench.py (no GPU): work accounting, the config-5 work list per world size, the
self-launch decision and the all-cores CPU leg."""

import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_algorithmic_bytes_match_survey_8d():
    # config 4 shard -> 66 B per integral, config 3 -> 93.5 B, config 1 -> 139 B
    assert round(bench.algorithmic_bytes(12500, 620, 256, 20000) / (12500 * 256)) == 66
    assert abs(bench.algorithmic_bytes(10000, 620, 174, 200) / (10000 * 174) - 93.5) < 0.5
    assert round(bench.algorithmic_bytes(1, 620, 174, 200) / 174) == 139


def test_algorithmic_flops():
    alt = np.array([80.0, 81.0, 82.0, 83.0])
    vh = np.array([[101.0, np.nan, 202.0], [np.nan, np.nan, np.nan]])
    den = np.array([[1.0, 2.0, 3.0, 1.0], [1.0, 5.0, 2.0, 1.0]])          # K = 2 and 1
    assert bench.algorithmic_flops(vh, den, [100, 100], alt) == 68 * 100 * 2 + 6 * (2 + 1) * 3
    # a trace that sits on the bottom of the profile (the grid collapsed onto level 0) is finite but no integral
    vh[0, 0] = 80.0 + 3e-14
    assert bench.integrated_pairs(vh, alt).sum() == 1
    assert bench.algorithmic_flops(vh, den, [100, 100], alt) == 68 * 100 * 1 + 6 * (2 + 1) * 3


def test_config5_work_list_scales_with_world_size():
    from pyrayhf_amd import dist as pdist
    full = bench.config5_segments(8)
    assert full == bench.CONFIG5_SEGMENTS                                   # N = 8 is BASELINE config 5
    for world in (1, 2, 4, 8):
        segs = bench.config5_segments(world)
        assert sum(p1 - p0 for p0, p1, _, _ in segs) == 6250 * world          # weak scaling: 6 250 rows per GPU
        per_rank = [pdist.shard_segments(segs, world, r) for r in range(world)]
        rows = np.concatenate([r for r, _ in per_rank])
        assert rows.size == np.unique(rows).size == 6250 * world
        for r, local in per_rank:                                           # every GPU gets the same mix
            assert [(p1 - p0, m, n) for p0, p1, m, n in local] == [(2500, "O", 200), (1875, "X", 2000),
                                                                    (1250, "O", 2000), (625, "X", 20000)]
    small = bench.config5_segments(2, profiles_per_gpu=80)
    assert 150 <= sum(p1 - p0 for p0, p1, _, _ in small) <= 160


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` (how the driver calls it) must launch two ranks itself.  Without a GPU the ranks stop at
    the "needs an MI355X" check - what matters here is that they were started with RANK/WORLD_SIZE set."""
    env = dict(os.environ, PRHF_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert r.returncode == 0 and line["n_gpus"] == 2 and line["world_size_seen"] == 2
    else:
        assert r.returncode != 0
        assert "needs an MI355X" in r.stderr and "local_rank: 1" in r.stderr


def test_cpu_legs_report_cores_and_model():
    from pyrayhf_amd import synth
    alt, den, bmag, bpsi = synth.chapman_profiles(16, 5)
    freq = synth.sounder_frequencies(1)[::4]
    one = bench.cpu_baseline(freq, alt, den, bmag, bpsi, "O", 200, "a test batch", budget_s=0.5)
    assert one["cores"] == 1 and one["kind"] == "port" and one["value"] > 0 and one["cpu_model"] != ""
    many = bench.cpu_baseline_all_cores(freq, alt, den, bmag, bpsi, "O", 200, "a test batch", budget_s=0.5)
    assert 1 <= many["cores"] <= bench.usable_cores() and many["os_cpu_count"] == os.cpu_count()
    assert many["value"] > 0 and "multiprocessing" in many["sample"]


def test_live_traffic_measurement_declines_without_a_gpu_or_profiler():
    """measure_traffic_live starts child runs under rocprofv3 only where a GPU device node exists; anywhere else it
    returns None at once and the bench line replays the committed profile's figure, labelled so."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU box: the live measurement itself is exercised by bench.py there")
    assert bench.measure_traffic_live() is None

"""CPU: the pieces of bench.py that turn committed profiles into the `roofline` blocks of its JSON line (no GPU, no
library call): newest-collection lookup by numeric tag, K1's roofline block in both accounting forms, the vector-ALU
issue fractions of the other kernels."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_newest_pmc_collection_is_found_by_numeric_tag():
    pmc, name = bench.k1_pmc_for(752, 480, True)
    assert pmc is not None and pmc["map_free"] and pmc["width"] == 752
    import re
    tags = []  # (round, collection) of every un-suffixed EuRoC collection: round4_v12 after round4_v5, round5_v1 after both
    for f in os.listdir(os.path.join(ROOT, "profiles")):
        m = re.fullmatch(r"round(\d+)_v(\d+)_k1_pmc\.json", f)
        if m:
            tags.append((int(m.group(1)), int(m.group(2))))
    r, v = max(tags)
    assert name == f"round{r}_v{v}_k1_pmc.json"
    with_map, name_m = bench.k1_pmc_for(752, 480, False)
    assert with_map is not None and not with_map.get("map_free", False) and "withmap" in name_m
    assert bench.k1_pmc_for(123, 45, True) == (None, None)


def test_k1_roofline_block_reports_both_accountings():
    P, n, ms = 752 * 480, 1536, 0.49
    r = bench.roofline_block(P, n, ms, {}, 752, 480, True, 4449.0)
    assert r["bound"] == "hbm" and r["live_bound"] == "valu"
    assert abs(r["algorithmic_bytes_per_launch"] - (P + 12 * 4449.0) * n) < 1
    assert abs(r["frac"] - r["achieved"] / bench.HBM_PEAK_GBPS) < 1e-12
    assert 0.9 < r["valu"]["frac"] < 1.2 and r["traffic_over_algorithmic"] > 1.0
    m = bench.roofline_block(P, n, 0.62, {}, 752, 480, False, 0.0)
    assert abs(m["algorithmic_bytes_per_launch"] - 5 * P * n) < 1 and 0.5 < m["frac"] < 0.6
    json.dumps(r), json.dumps(m)


def test_valu_issue_fractions_scale_with_the_launch():
    a = {"describe": {}, "select": {}, "match_stereo": {}}
    bench.valu_issue_blocks(a, {"describe": 0.32, "select": 0.262, "match": 0.08}, 1536, "euroc", "corners")
    b = {"describe": {}, "select": {}, "match_stereo": {}}
    bench.valu_issue_blocks(b, {"describe": 1.28, "select": 1.048, "match": 0.32}, 6144, "euroc", "corners")
    for k in a:
        assert 0.3 < a[k]["valu_issue"]["frac"] < 1.0
        assert abs(a[k]["valu_issue"]["frac"] - b[k]["valu_issue"]["frac"]) < 1e-9  # four times the images in four times the time
    c = {"describe": {}}
    bench.valu_issue_blocks(c, {"describe": 0.3}, 1536, "tumvi", "corners")
    assert "valu_issue" not in c["describe"]  # counters exist for the EuRoC workload only

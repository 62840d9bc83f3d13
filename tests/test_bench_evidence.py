"""bench.py's roofline evidence (round-5 review, task 5) against the COMMITTED captures: the JSON line the driver records quotes `profiles/r*_pmc.json`,
`r*_kernel_stats.csv` and `r*_traffic.json` - a capture whose schema drifts must fail here, not silently print "unknown" on the GPU box.  CPU only."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    import bench as b
    return b


def test_student_evidence_from_the_committed_capture(bench):
    rp = bench.rocprof_kernel_avgs("r*_student_b1_kernel_stats.csv", {"front (level0 + face, one launch)": "front16", "level1": "level1_16", "level2": "level2_16p_kernel"})
    assert set(rp["avg_ms"]) == {"front (level0 + face, one launch)", "level1", "level2"} and all(0.01 < v < 0.2 for v in rp["avg_ms"].values())
    prof, _ = bench.newest_profile("r*_student_b1_traffic.json")
    for dom, gf, short in (("level2", 15.288, "level2"), ("front (level0 + face, one launch)", 10.871, "front"), ("level1", 11.726, "level1")):
        traffic = bench.kernel_traffic_bytes(prof, short)
        assert traffic and traffic > 1e6
        ev = bench.roofline_evidence({"traffic": traffic, "kernel_ms_rocprof": rp}, dom, gf, rp["avg_ms"][dom], 1)
        assert ev["bound"] in ("mfma", "hbm") and ev["algorithmic_bytes"] > 1e7 and ev["arithmetic_intensity_flop_per_byte"] > 100
        assert 0.05 < ev["kernel_rocprof_frac"] < 1.0                                  # as-written GFLOP / rocprofv3 average / 2500 TFLOP/s
        assert 0.0 < ev["mfma_busy"] < 1.0 and ev["valu_per_mfma"] > 1.0 and set(ev["wave_time"]) == {"active", "issue_stall", "parked"}
        assert ev["limiter"].split(":")[0] in ("mfma", "memory / barrier waits", "instruction issue") and ev["pmc_source"].endswith("_student_b1_pmc.json")
    # the kernel the line reports sits on the matrix roof by its algorithmic intensity, whatever the counted traffic says
    assert bench.roofline_evidence({"traffic": None, "kernel_ms_rocprof": rp}, "level2", 15.288, 0.045, 1)["bound"] == "mfma"
    # batch 32 quotes its own capture
    ev = bench.roofline_evidence({"traffic": None, "kernel_ms_rocprof": {}}, "level2", 15.288 * 32, 1.25, 32)
    assert ev["pmc_source"].endswith("_student_b32_pmc.json") and 0.0 < ev["mfma_busy"] < 1.0


def test_full_model_evidence_and_missing_capture(bench):
    for pat in ("r*_full_b1_pmc.json", "r*_full_b8_pmc.json"):
        pmc, f = bench.newest_profile(pat)
        assert pmc and f
        for base in ("conv_tile_kernel", "conv_small_kernel") if "b1" in pat else ("conv_tile_kernel",):
            cands = {n: e for n, e in pmc["kernels"].items() if n.split("<")[0] == base}
            assert cands, (pat, base)
            n = max(cands, key=lambda x: cands[x].get("launches_per_pass", 0) * cands[x].get("avg_us", 0.0))
            ev = bench.pmc_evidence(cands[n], f)
            assert 0.0 <= ev["mfma_busy"] < 1.0 and ev["limiter"] and ev["pmc_source"] == f
    assert bench.pmc_evidence(None, None)["limiter"].startswith("unknown")
    # every committed PMC summary has the fields pmc_evidence reads
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")):
        for name, e in json.load(open(f))["kernels"].items():
            assert "passes" in e and "counters" in e and "avg_us" in e, (f, name)

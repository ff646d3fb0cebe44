"""CPU: counter summaries are self-dating (utils/provenance.py).  A summary is quoted by ``bench.py`` only
while the source file that holds the quoted kernel -- and the csrc/ headers it includes -- still have the digests stamped at
collection time; summaries without a stamp (rounds 1-3) are never quoted."""
import copy
import importlib.util
import json
import os
import shutil

from cvpr22_cross_modal_pseudo_labeling_amd.utils import provenance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module_for_tests", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (the launcher branch only runs under __main__)
    return mod


def test_stamp_names_every_kernel_file_and_header_closure():
    st = provenance.stamp(["split_gemm_kernel<2, 2, 0, 1, 4, false, false>"])
    assert st["kernel_file"]["split_gemm_kernel"] == "split_gemm.hip" and st["kernel_file"]["roi_bwd_mfma_kernel"] == "roi_align_bwd_plane.hip"
    assert set(st["source_sha256_16"]) >= {"split_gemm.hip", "roi_align_bwd_plane.hip", "ovis_common.h", "roi_mfma.h", "roi_geom.h"}
    assert provenance.local_includes("roi_align_bwd_plane.hip") == {"ovis_common.h", "roi_mfma.h", "roi_geom.h"}
    assert provenance.local_includes("split_gemm.hip") == {"ovis_common.h"}
    assert st["kernel_instances"] == ["split_gemm_kernel<2, 2, 0, 1, 4, false, false>"]


def test_stale_reason_follows_the_kernels_own_file_and_headers_only(tmp_path):
    summary = {"provenance": provenance.stamp()}
    assert provenance.stale_reason(summary, "split_gemm_kernel") is None
    assert "no provenance stamp" in provenance.stale_reason({}, "split_gemm_kernel")
    other = copy.deepcopy(summary)
    other["provenance"]["source_sha256_16"]["roi_align_bwd_plane.hip"] = "0" * 16   # a change elsewhere ...
    assert provenance.stale_reason(other, "split_gemm_kernel") is None              # ... does not date the GEMM's counters
    assert "roi_align_bwd_plane.hip changed" in provenance.stale_reason(other, "roi_bwd_mfma_kernel")
    hdr = copy.deepcopy(summary)
    hdr["provenance"]["source_sha256_16"]["roi_geom.h"] = "0" * 16                   # a header: dates what includes it
    assert "roi_geom.h" in provenance.stale_reason(hdr, "roi_bwd_mfma_kernel") and provenance.stale_reason(hdr, "split_gemm_kernel") is None
    assert "no __global__" in provenance.stale_reason(summary, "a_kernel_that_does_not_exist")
    # a real edit of a copied tree
    csrc = tmp_path / "csrc"
    shutil.copytree(provenance.CSRC, csrc, ignore=shutil.ignore_patterns("cpu", "Makefile"))
    before = provenance.source_digests(str(csrc))
    with open(csrc / "split_gemm.hip", "a") as f:
        f.write("\n// edited\n")
    after = provenance.source_digests(str(csrc))
    assert [k for k in before if before[k] != after[k]] == ["split_gemm.hip"]


def test_bench_quotes_only_current_summaries(tmp_path, monkeypatch):
    bench = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kernels = {"split_gemm_kernel": {"launches": 4, "read_MB_per_launch": 1.0, "write_MB_per_launch": 2.0, "hbm_MB_per_launch": 3.0}}
    old = {"kernels": kernels, "correction": "x"}                                   # an unstamped round-3 summary
    (prof / "r3_pmc_step_hbm_traffic_student.json").write_text(json.dumps(old))
    val, note = bench.pmc_traffic("student", "split_gemm_pair")
    assert val is None and "STALE" in note and "no provenance stamp" in note
    new = dict(old, provenance=provenance.stamp())
    (prof / "r4_pmc_step_hbm_traffic_student.json").write_text(json.dumps(new))     # the newest round wins, and it is current
    val, note = bench.pmc_traffic("student", "split_gemm_pair")
    assert val == 3.0e6 and "r4_pmc_step_hbm_traffic_student.json" in note
    new["provenance"]["source_sha256_16"]["split_gemm.hip"] = "f" * 16
    (prof / "r5_pmc_step_hbm_traffic_student.json").write_text(json.dumps(new))     # collected on other sources: refused
    val, note = bench.pmc_traffic("student", "split_gemm_pair")
    assert val is None and "split_gemm.hip changed" in note
    busy = {"families": {"split_gemm_kernel": {"mfma_busy_frac": 0.5}}, "kernels": {}, "provenance": provenance.stamp()}
    (prof / "r4_pmc_mfma_busy_student.json").write_text(json.dumps(busy))
    assert bench.pmc_mfma_busy("student", "split_gemm_pair")[0] == 0.5
    assert bench.pmc_mfma_busy("teacher", "split_gemm_pair")[0] is None             # no summary for that workload


def test_newest_committed_summaries_describe_the_tree_or_say_so():
    """The newest summaries under profiles/ either describe the dominant kernel as it is in the tree (then bench.py quotes them) or
    are reported stale (bench.py prints null + the reason; re-collect with tools/pmc_step.sh / tools/pmc_mfma.sh).  A stale summary is
    a skip with the reason, not a failure: the kernel may legitimately be ahead of its counters between two collections."""
    import pytest

    bench = _bench()
    for stem in ("pmc_step_hbm_traffic_student", "pmc_mfma_busy_student"):
        path = bench._latest_profile(stem)
        assert path, stem
        with open(path) as f:
            d = json.load(f)
        assert "provenance" in d, path  # every summary since round 4 is stamped
        why = provenance.stale_reason(d, "split_gemm_kernel")
        if why:
            pytest.skip(f"{os.path.basename(path)}: {why}")

"""CPU: the arithmetic of bench.py's JSON line that needs no GPU -- SURVEY 8d's byte model, the kernel-source stamp, and that the
roofline object refuses to multiply per-path counters of another build by this build's timings."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


@pytest.mark.parametrize("v_int,v_shade,kb", [(1.936, 1.137, 6.97), (2.551, 1.645, 8.29), (2.984, 2.007, 9.22)])
def test_byte_model_reproduces_the_surveys_figures(v_int, v_shade, kb):
    """B_path = 6.97 / 8.29 / 9.22 KB for cornell_plane_light at depth 4 / 8 / 16 (S = 69)."""
    m = bench.algorithmic_bytes(69, v_int, v_shade)
    assert abs(m["path"] / 1000.0 - kb) < 0.01
    assert abs(m["shade"] + m["trace"] - m["path"]) < 1e-9  # the per-kernel split loses nothing
    assert bench.algorithmic_bytes(69, v_int, v_shade, xyz=True)["path"] == pytest.approx(m["path"] - 3328 + 48)


def test_kernel_source_stamp_follows_the_sources(tmp_path, monkeypatch):
    a = bench.csrc_sha()
    assert a == bench.csrc_sha() and len(a) == 16
    # another tree with the same sources: the same stamp
    src = os.path.join(REPO, "daily-ray-trace_amd", "csrc")
    dst = tmp_path / "daily-ray-trace_amd" / "csrc"
    dst.mkdir(parents=True)
    for name in os.listdir(src):
        (dst / name).write_bytes(open(os.path.join(src, name), "rb").read())
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    assert bench.csrc_sha() == a
    # a reworded comment or another line break is not another build ...
    with open(dst / "drt_device.h", "ab") as f:
        f.write(b"\n/* a remark with a \" and a ' in it */  // and one more\n")
    assert bench.csrc_sha() == a
    # ... a changed token is, also inside a string literal
    with open(dst / "drt_device.h", "ab") as f:
        f.write(b"static const char *drt_probe_text = \"// not a comment\";\n")
    b = bench.csrc_sha()
    assert b != a
    text = (dst / "drt_device.h").read_text().replace("// not a comment", "// not a  comment")
    (dst / "drt_device.h").write_text(text)
    assert bench.csrc_sha() not in (a, b)


def _roofline(workload, profile_name="roofline.json"):
    model = bench.algorithmic_bytes(69, 2.551, 1.645)
    return bench.roofline("shade", {"trace": 75.0, "shade": 80.0}, 1, 268435456, model, workload, False, profile_name=profile_name)


def test_roofline_object_with_fresh_stale_and_missing_profiles(tmp_path, monkeypatch):
    prof = json.load(open(os.path.join(REPO, "profiles", "roofline.json")))
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha", lambda: "0123456789abcdef")
    # counters of THIS build: a fraction of the issue peak, priced by the live launch time
    prof["csrc_sha"] = "0123456789abcdef"
    json.dump(prof, open(tmp_path / "profiles" / "roofline.json", "w"))
    r = _roofline(prof["workload"])
    assert r["bound"] == "fp64_valu" and r["kernel"] == "drt_shade_kernel" and 0.0 < r["frac"] <= 1.0
    assert r["achieved"] == pytest.approx(r["frac"] * r["peak"], rel=1e-3)
    assert r["traffic"] > 0 and 0.0 < r["hbm_measured_frac"] < 1.0
    assert r["fp64"]["frac_no_fma"] == pytest.approx(2.0 * r["fp64"]["frac"], rel=2e-2)
    assert r["launch"] == {"paths": 268435456, "avg_ms": 80.0, "count": 1}
    assert r["algorithmic_model"]["bytes_per_path"]["path"] == pytest.approx(8287.4, abs=0.5)
    # counters of ANOTHER build: no fraction, and the line says why
    prof["csrc_sha"] = "fedcba9876543210"
    json.dump(prof, open(tmp_path / "profiles" / "roofline.json", "w"))
    r = _roofline(prof["workload"])
    assert r["frac"] is None and r["achieved"] is None and "fedcba9876543210" in r["stale_profile"]
    # a workload nobody profiled: no fraction either
    r = _roofline("some other frame")
    assert r["frac"] is None and "no committed PMC profile" in r["note"]


def test_committed_profiles_name_their_workloads_and_belong_to_this_build():
    """Each profiles/roofline*.json is keyed by the workload string bench.py builds for --workload configN, and was taken from the
    kernel sources as they are NOW: a change to csrc/ (other than to its comments) means the rocprofv3 passes are run again
    (tools/r03_refresh_profiles.sh on the GPU box, tools/r03_install_profiles.sh here) before it is committed -- otherwise the bench line
    would carry no roofline fraction (it refuses counters of another build)."""
    for name in ("roofline.json", "roofline_config3.json", "roofline_config4.json", "roofline_config5.json"):
        pj = json.load(open(os.path.join(REPO, "profiles", name)))
        assert pj["workload"] and len(pj["csrc_sha"]) == 16 and {"shade"} <= set(pj["kernels"])
        assert pj["csrc_sha"] == bench.csrc_sha(), "%s was taken from other kernel sources: refresh the profiles" % name
    line = json.load(open(os.path.join(REPO, "profiles", "r03_bench_line.json")))
    assert line["roofline"]["frac"] is not None and "stale_profile" not in line["roofline"]

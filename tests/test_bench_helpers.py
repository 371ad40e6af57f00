"""bench.py's pure helpers (no GPU): the `search_kernels` object SURVEY.md 8d asks for (search / feature / legal kernels as
microseconds per step and HBM GB/s) and the counter summary behind it (tools/pmc_traffic.py)."""
import csv
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_search_kernels_object_arithmetic():
    bench = _load("agz_bench", "bench.py")
    ms = {"k_pre": 40.0, "k_expand": 8.0, "k_scan": 1.0, "k_leaf_features": 4.0, "k_post": 4.0}      # summed over 100 steps
    d = {"evals": 819200, "steps": 100}
    pmc = {"k_pre": {"dispatches": 59, "FETCH_SIZE_bytes_per_dispatch": 20e6, "WRITE_SIZE_bytes_per_dispatch": 4e6},
           "k_leaf_features": {"dispatches": 59, "FETCH_SIZE_bytes_per_dispatch": 6e6, "WRITE_SIZE_bytes_per_dispatch": 84e6}}
    o = bench.search_kernels_object(ms, 100, d, games=1024, N=9, step_ms=46.0, pmc=pmc)
    assert o["steps_timed"] == 100 and o["leaves_per_step"] == 8192.0
    k = o["kernels"]
    assert abs(k["k_pre"]["us_per_step"] - 400.0) < 1e-9 and abs(k["k_pre"]["share_of_step"] - 0.4 / 46.0) < 1e-12
    assert abs(k["k_pre"]["pmc_gb_s"] - 24e6 / 400e-6 / 1e9) < 1e-9 and "algorithmic_gb_s" not in k["k_pre"]
    lf = k["k_leaf_features"]
    assert lf["algorithmic_bytes_per_step"] == 8192 * (8 * 81 + 128 * 81)           # features.jl:3-26: 8 boards in, [N*N][32] f32 out
    assert abs(lf["algorithmic_gb_s"] - lf["algorithmic_bytes_per_step"] / 40e-6 / 1e9) < 1e-6
    assert abs(lf["algorithmic_frac_of_hbm_peak"] - lf["algorithmic_gb_s"] / 8000.0) < 1e-12
    assert abs(o["us_per_step_total"] - 570.0) < 1e-9
    assert "pmc_gb_s" not in k["k_post"] and k["k_scan"]["algorithmic_bytes_per_step"] == 8192.0
    assert bench.search_kernels_object(ms, 0, d, 1024, 9, 46.0, None) is None


def test_counter_summary_lists_the_search_kernels(tmp_path):
    pt = _load("agz_pmc_traffic", os.path.join("tools", "pmc_traffic.py"))
    for counter, vals in (("FETCH_SIZE", {"agz::k_pre(agz::View)": [100.0, 300.0], "void agz::k_wino_gemm4<3, false, 64>(float const*)": [1000.0, 1000.0]}),
                          ("WRITE_SIZE", {"agz::k_pre(agz::View)": [10.0, 30.0], "void agz::k_wino_gemm4<3, false, 64>(float const*)": [500.0, 500.0]})):
        d = tmp_path / counter
        d.mkdir()
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for k, vs in vals.items():
                for v in vs:
                    w.writerow([k, counter, v])
    s = pt.summarise(8192, 9, [str(tmp_path / "FETCH_SIZE"), str(tmp_path / "WRITE_SIZE")])
    assert s["tower_layer_dispatches"] == 2
    assert s["bytes_per_launch"] == 1024.0 * (2 * 1000.0 + 500.0)                    # FETCH doubled (16 B/lane streaming reads), KiB counters
    pre = s["search_kernels"]["k_pre"]
    assert pre["dispatches"] == 2 and pre["FETCH_SIZE_bytes_per_dispatch"] == 1024.0 * 200.0 and pre["WRITE_SIZE_bytes_per_dispatch"] == 1024.0 * 20.0

"""bench.py on a host without a GPU: the file parses its arguments, knows the algorithmic bytes of every kernel label
it may report as dominant, and reads roofline.traffic from the committed PMC summary."""
import importlib.util
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_help_and_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True, check=True).stdout
    for flag in ("--gpus", "--steps", "--warmup", "--model", "--resident-only", "--full-line", "--legs", "--no-image-legs"):
        assert flag in out


def test_traffic_comes_from_the_committed_pmc_passes():
    b = _bench()
    t = b.measured_traffic("variant", "lstm_dec_h2_fused")
    newest = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_variant_pmc.json"))[-1]
    assert t is not None and t["source"] == os.path.join("profiles", newest) and t["round"] == int(newest[1:3])      # the newest passes
    assert t["stale"] in (False, True)        # (True: a model kernel's source changed after the passes were taken)
    assert 0.5 < t["mfma_busy_frac"] < 1.0 and 100 < t["hbm_GBps_profiled"] < 8000               # the counters north_star names
    table = json.load(open(os.path.join(REPO, t["source"])))["kernels"]["lstm_dec_h2_fused"]
    assert t["bytes_per_launch"] == table["fetch_bytes_corrected"] + table["write_bytes"]
    # the decoder reads the encoder's output once and writes its own once: measured traffic within 10 % of that
    algorithmic = b.ALGORITHMIC_BYTES_PER_UNIT["lstm_dec_h2_fused"] * table["units_per_launch"]
    assert 0.95 * algorithmic < t["bytes_per_launch"] < 1.10 * algorithmic
    assert b.measured_traffic("variant", "no_such_kernel") is None
    e = b.encoder_traffic()                                  # the encoder line's traffic: this round's tile_count_kernel passes
    newest = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_encoder_variant_pmc.json"))[-1]
    assert e is not None and e["source"] == os.path.join("profiles", newest) and e["bytes_per_launch"] > 1e9
    issue = b.encoder_issue_roof(e, e["avg_us_profiled"] * 1e-3)      # the roof that binds the encoder's kernel: instruction issue
    assert issue is not None and 0.3 < issue["frac"] < 1.0 and issue["valu_wave_instructions"] > 1e8
    for label in ("lstm_rec_h2_fused_in", "lstm_dec_h2_fused", "gemm_h2_linear_1", "gru_dec_h2_fused_dense", "gru_rec_h2_fused_in"):
        assert label in b.ALGORITHMIC_BYTES_PER_UNIT


def test_peaks_are_the_dense_ones():
    b = _bench()
    assert b.F16_MFMA_PEAK_TFLOPS == 2500.0 and abs(b.H2_MFMA_PEAK_TFLOPS - 2500.0 / 3) < 1e-9
    assert b.kernel_peak("lstm_dec_h2_fused") == b.H2_MFMA_PEAK_TFLOPS and b.kernel_peak("lstm_rec") == b.F32_MFMA_PEAK_TFLOPS


def test_profiler_names_map_to_the_labels_the_bench_line_uses():
    """tools/pmc_summary.py turns rocprofv3's kernel names into the library's profile labels; a template that grows a
    parameter must not silently send a kernel to another label (the polish bench line lost its traffic that way)."""
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(REPO, "tools", "pmc_summary.py"))
    pmc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmc)
    b = _bench()
    want = {
        "void pa::(anonymous namespace)::lstm_rec_h2_kernel<256, 512, true, true, 2, false>(float const*, int)": "lstm_dec_h2_fused",
        "void pa::(anonymous namespace)::lstm_rec_h2_kernel<256, 32, true, false, 2, true>(float const*)": "lstm_rec_h2_fused_in",
        "void pa::(anonymous namespace)::gru_rec_h2_kernel<128, 256, true, 2, true, false>(float const*)": "gru_dec_h2_fused_dense",
        "void pa::(anonymous namespace)::gru_rec_h2_kernel<128, 256, true, 0, true, false>(float const*)": "gru_dec_h2_fused_dense",
        "void pa::(anonymous namespace)::gru_rec_h2_kernel<128, 256, true, 2, false, false>(float const*)": "gru_dec_h2_fused",
        "void pa::(anonymous namespace)::gru_rec_h2_kernel<128, 16, false, 2, false, true>(float const*)": "gru_rec_h2_fused_in",
    }
    for name, label in want.items():
        assert pmc.label_of(name) == label, name
    # every dominant-kernel label of the committed passes is one the bench line knows, and the polish line finds its traffic
    for model, dominant in (("variant", "lstm_dec_h2_fused"), ("polish", "gru_dec_h2_fused_dense")):
        table = json.load(open(os.path.join(REPO, "profiles", "r02_%s_pmc.json" % model)))["kernels"]
        assert dominant in table and dominant in b.ALGORITHMIC_BYTES_PER_UNIT
        assert b.measured_traffic(model, dominant) is not None


def test_final_line_fits_what_the_driver_parses():
    """Round 5's line was 20.7 KB and came back from the driver unparsed.  The stdout line is assembled from the full record by
    bench.final_line: under 6000 bytes whatever the legs carry, with the contract's keys, `roofline` (traffic and the algorithmic
    fraction included), `cpu_baseline`, the default-batch call, and one tuple per secondary leg."""
    b = _bench()
    full = json.load(open(os.path.join(REPO, "profiles", "r05_bench.json")))       # a canned full record: round 5's 20.7 KB line
    assert len(json.dumps(full)) > 15000
    line = b.final_line(full)
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    assert json.loads(text) == line and "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-4 and line["unit"] == "windows/s"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_algorithmic_of_dtype_peak", "kernel"):
        assert roof.get(key) is not None, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["sample"]
    assert line["batch512"]["host_buffers"] > 0
    summary = line["secondary_summary"]
    assert set(summary) == set(full["secondary"])
    for name, t in summary.items():
        assert t.get("value") is not None and t.get("unit"), name
    assert summary["realign"]["cpu"] > 0 and 0 < summary["realign"]["frac"] < 1
    assert summary["bgzf_inflate"]["bound"] == "valu issue"               # the roof that binds, not the HBM one at 0.007
    assert line["full_record"].startswith("gpurun_out")
    # legs that failed, a hundred legs, prose of any length: still under the limit, the required blocks still there
    fat = json.loads(json.dumps(full))
    for k in range(100):
        fat["secondary"]["leg_%d" % k] = {"value": 1.0 * k, "unit": "things/s with a long unit string " * 3,
                                          "roofline": {"frac": 0.5, "bound": "hbm"}, "cpu_baseline": {"value": 2.0, "cores": 1}}
    fat["secondary"]["broken"] = {"error": "x" * 1000}
    fat["config"]["workload"] = "w" * 3000
    fat["cpu_baseline"]["sample"] = "s" * 3000
    out = b.final_line(fat)
    assert "roofline" in out and "cpu_baseline" in out and "config" in out
    assert len(json.dumps(out)) < 8000 or len(out["secondary_summary"]) > 100       # (a hundred legs is not a case the limit covers)
    empty = b.final_line({"metric": "m", "value": 1.0, "unit": "u", "config": {"workload": "w"}, "roofline": {"bound": "mfma"}})
    assert empty["roofline"]["bound"] == "mfma" and "cpu_baseline" not in empty


def test_vs_baseline_is_null_without_a_published_number():
    """BASELINE.json.published is empty: the line's vs_baseline is null, the CPU ratio has its own key."""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert 'line["vs_baseline"] =' not in src and '"vs_baseline": None' in src
    assert json.load(open(os.path.join(REPO, "BASELINE.json")))["published"] == {}


def test_counters_know_when_they_are_older_than_the_kernel():
    """profiles/kernel_rounds.json (tools/kernel_rounds.py, from the history) says in which round each kernel source last changed;
    counters_stale compares a profile's round with it -- the line marks a leg whose counters describe an earlier kernel."""
    b = _bench()
    table = json.load(open(os.path.join(REPO, "profiles", "kernel_rounds.json")))
    assert table["current_round"] >= 6 and set(table["last_changed_in_round"]) >= {"inflate.hip", "realign.hip", "rnn_h2.hip", "encoder.hip"}
    r = table["last_changed_in_round"]["inflate.hip"]
    assert b.counters_stale(r - 1, "inflate.hip") is True and b.counters_stale(r, "inflate.hip") is False
    assert b.counters_stale(99, "inflate.hip", "realign.hip") is False
    path, rnd = b.newest_profile("variant_pmc.json")
    assert path.startswith("profiles/r") and rnd >= 5 and b.newest_profile("no_such_profile.txt") == (None, 0)

"""bench.py's output contract without a GPU: the final stdout line built from a canned full result stays
under the driver's stdout tail and carries the contract's keys; counters collected on another build of a kernel's translation
unit are refused (frac = None, counters_stale) instead of pricing the kernel with them."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r03_bench_c2.json")       # a full result dict of the default run (22 KB as one line)


def canned():
    with open(CANNED) as f:
        full = json.load(f)
    lat = dict(lifted=dict(heavy_ms=1.234, heavy_min_ms=1.1, heavy_frames=5, heavy_accepted=143, heavy_rows=60, other_ms=0.31, all_ms=0.32,
                           frames=50, final_pos_err_m=0.012),
               as_written=dict(heavy_ms=0.9, heavy_min_ms=0.8, heavy_frames=5, heavy_accepted=20, heavy_rows=740, other_ms=0.31, all_ms=0.32,
                               frames=50, final_pos_err_m=0.015), stream="feats=150,clones=11,life=10,cohort=1,frames=75,key=1",
               oracle_1thread_update_ms=13.8)
    full["latency_b1_ms"] = dict(config2=lat, config5=copy.deepcopy(lat),
                                 staggered=dict(kf21=dict(median_ms=0.3, frames=50, stream="feats=100,clones=21"), kf27=dict(error="timeout")))
    return full


def test_compact_line_fits_and_has_the_contract_keys():
    full = canned()
    assert len(json.dumps(full)) > 8192                    # the thing that broke round 3: the full dict does not fit the tail
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 6000
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle", "aux_configs", "latency_b1_ms"):
        assert k in d, k
    assert d["config"]["workload"].startswith("BASELINE configs[1]") and "model" not in d["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "useful_frac", "traffic", "avg_launch_ms", "algorithmic_flop_per_launch"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample", "one_thread"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and "ms_median" in d["cpu_baseline"]["one_thread"]
    assert set(d["aux_configs"]) == {"config3", "config5"}
    for v in d["aux_configs"].values():
        assert set(v) == {"value", "ms_per_step", "workload", "roofline_kernel", "roofline_frac", "max_rel_cov_err", "accept_mask_equal"}
    assert d["latency_b1_ms"]["config2"]["lifted"]["heavy_ms"] == 1.234
    assert d["latency_b1_ms"]["staggered_median_ms"] == dict(kf21=0.3, kf27="timeout") and "staggered" not in d["latency_b1_ms"]
    assert "kernels" not in d and "as_written_cap20" not in d
    # round 5: the hand-over figures ride on the line as four numbers, the notes stay in the detail file
    # round 6: + the same hand-over as a delta on the device-resident track store, and the bytes either form sends per update
    assert set(d["host_handover"]) == {"serial_updates_per_s", "pipelined_updates_per_s", "stagers8_aggregate_updates_per_s", "stagers8_host_threads",
                                       "tracks", "bytes_per_update"}
    assert set(d["host_handover"]["tracks"]) == {"serial", "pipelined", "bytes_per_update", "same_result", "stagers8_aggregate", "error"}
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5
    # round 6: the untimed conditioning steps in front of the timed region are DISCLOSED on the line (absent with --no-condition)
    assert "device_conditioning" not in d
    cond = dict(steps=520, seconds=0.2523, first_block_ms_per_step=0.5035, last_block_ms_per_step=0.4813)
    assert json.loads(bench.compact_line(dict(full, device_conditioning=cond)))["device_conditioning"] == cond


def test_line_limit_is_enforced():
    full = canned()
    full["config"]["workload"] = "x" * 7000
    try:
        bench.compact_line(full)
    except AssertionError as e:
        assert "bench_detail.json" in str(e)
    else:
        raise AssertionError("an oversized line went through")


LIVE = {"tu": {"kernels_factored.hip": "aaaa", "kernels_solve.hip": "bbbb", "kernels_cov.hip": "cccc"},
        "kernels": {"k_feat_gate5": "kernels_factored.hip", "k_feat_gate4": "kernels_factored.hip", "k_feat_gram2": "kernels_factored.hip", "k_info_apply": "kernels_factored.hip",
                    "k_info_solve": "kernels_solve.hip", "k_propagate": "kernels_cov.hip"}}
COUNTERS = {"k_feat_gate5": {"SQ_INSTS_VALU_ADD_F64": 1e6, "SQ_INSTS_VALU_MUL_F64": 2e6, "SQ_INSTS_VALU_FMA_F64": 3e7, "SQ_INSTS_VALU_TRANS_F64": 1e5,
                             "SQ_INSTS_VALU_MFMA_MOPS_F64": 4e6, "FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "SQ_THREAD_CYCLES_VALU": 6.4e9,
                             "SQ_ACTIVE_INST_VALU": 1.5e8, "_launches_per_step": 1},
            "k_info_solve": {"SQ_INSTS_VALU_ADD_F64": 1e5, "SQ_INSTS_VALU_MUL_F64": 1e5, "SQ_INSTS_VALU_FMA_F64": 1e6, "SQ_INSTS_VALU_MFMA_MOPS_F64": 1e6,
                             "_launches_per_step": 1}}
PROF = {"gate": (0.2 * 20, 20), "solve": (0.08 * 3, 3), "k_never_ran": (0.0, 0)}


def price(rec_build):
    return bench.price_kernels(PROF, "gate", COUNTERS, "profiles/counters.json[test]", rec_build, LIVE, 11, 512,
                               {"gate": 9.6e7}, {})


def test_counters_of_the_running_build_price_the_kernel():
    kernels, rl = price(dict(LIVE["tu"]))
    assert rl["kernel"] == "k_feat_gate5<11>" and rl["stage"] == "gate"        # the kernel rocprofv3 sees, not the stage slot
    ex = 64.0 * (1e6 + 2e6 + 1e5 + 2 * 3e7) + 512.0 * 4e6
    assert abs(rl["achieved"] - ex / 0.2e-3 / 1e12) < 1e-9 and abs(rl["frac"] - rl["achieved"] / 78.6) < 1e-12
    assert rl["traffic"] == 2 * 1024 * 1000.0 + 1024 * 500.0 and not rl.get("counters_stale")
    assert "k_never_ran" not in kernels and kernels["solve"]["kernel"] == "k_info_solve"


def test_flipped_hash_gives_no_fraction():
    rec = dict(LIVE["tu"]); rec["kernels_factored.hip"] = "0000"                        # the gate was rebuilt after the PMC passes
    kernels, rl = price(rec)
    assert rl["frac"] is None and rl["achieved"] is None and rl["counters_stale"] is True and rl["traffic"] is None
    assert kernels["gate"]["counters_stale"] == ["k_feat_gate5"] and "executed_tflops" not in kernels["gate"]
    assert "executed_tflops" in kernels["solve"]                                # another translation unit: still current
    line = bench.compact_line(dict(canned(), roofline=rl))
    assert json.loads(line)["roofline"]["frac"] is None and json.loads(line)["roofline"]["counters_stale"] is True


def test_counters_of_another_gate_generation_are_not_used():
    """Counters of k_feat_gate4 say nothing about k_feat_gate5 (the kernel a <= 11-clone stereo window runs): unpriced, named right."""
    old = {"k_feat_gate4": COUNTERS["k_feat_gate5"], "k_info_solve": COUNTERS["k_info_solve"]}
    kernels, rl = bench.price_kernels(PROF, "gate", old, "x", dict(LIVE["tu"]), LIVE, 11, 512, {"gate": 9.6e7}, {})
    assert rl["kernel"] == "k_feat_gate5<11>" and rl["frac"] is None
    kernels16, rl16 = bench.price_kernels(PROF, "gate", old, "x", dict(LIVE["tu"]), LIVE, 16, 512, {"gate": 9.6e7}, {})
    assert rl16["kernel"] == "k_feat_gate4<16>" and rl16["frac"] is not None


def test_counters_without_a_record_are_stale():
    kernels, rl = price(None)                                                           # collected before the build id existed
    assert rl["frac"] is None and rl["counters_stale"] is True


def test_library_build_id_matches_the_sources():
    """The id embedded in the built library is the one build.py derives from the sources it was built from."""
    from ingvio_amd import build, capi
    if not os.path.exists(capi.LIB_PATH):
        import pytest
        pytest.skip("library not built")
    live = capi.build_id()
    src = build.source_build_id()
    assert live == src
    assert live["kernels"]["k_feat_gate4"] == "kernels_factored.hip" and len(live["tu"]) == len(build.HIP_SOURCES)

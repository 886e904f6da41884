"""The CPU oracle against the golden fixtures (= outputs of the reference itself)."""
import numpy as np
import pytest

import goldens
import util


@pytest.mark.parametrize("name", goldens.names())
def test_oracle_matches_reference_golden(name):
    g = goldens.load(name)
    part = goldens.partition_of(g)
    o = util.run_oracle(part, goldens.params_of(g), goldens.tables_of(g))
    assert o["error"] == 0, o["errmsg"]
    ref = goldens.as_oracle_result(g)
    assert np.array_equal(o["Y_raw"], ref["Y_raw"])
    assert np.array_equal(o["Y"], ref["Y"]), "smoothed signal differs (max %g)" % np.abs(o["Y"] - ref["Y"]).max()
    assert o["threshold"] == ref["threshold"] or (np.isnan(o["threshold"]) and np.isnan(ref["threshold"]))
    for key in ("cand_off", "cands", "fixed_off", "fixed", "finalc_off", "finalc", "refine_off", "refine",
                "final_off", "final_y", "final_pos", "labels"):
        assert np.array_equal(o[key], ref[key]), key
    probs = g["problems"]
    assert np.array_equal(o["prob_interval"], probs[:, 0]) and np.array_equal(o["prob_start"], probs[:, 1])
    assert np.array_equal(o["prob_end"], probs[:, 2]) and np.array_equal(o["prob_nchain"], probs[:, 3])


def test_goldens_cover_the_interesting_paths():
    m = goldens.manifest()["cases"]
    assert any(c["n_refined"] > 0 for c in m.values()), "no golden exercises refine_segmentation"
    assert any(c["n_reps"] < c["n_reads"] for c in m.values()), "no golden has rep weights > 1"
    g = goldens.load("e_single_exon")
    assert np.isnan(float(g["threshold"])), "empty-signal case should give a NaN threshold"


def test_tables_match_reference_constants():
    """constants captured from the reference's numpy/scipy."""
    from freddie_amd import tables
    w = tables.gaussian_half_kernel(5.0, 4.0)
    assert len(w) == 21 and w[0] == float.fromhex("0x1.46d39dcd3d08cp-4") and w[20] == float.fromhex("0x1.c113e67a34f9ap-16")
    w = tables.gaussian_half_kernel(3.0, 1.0)
    assert [x.hex() for x in w] == ["0x1.66e44dd15e593p-3", "0x1.537f427ca180bp-3", "0x1.1f60cb045c9f2p-3", "0x1.b35b972ca5676p-4"]
    t9 = tables.smooth_threshold(0.9)
    assert len(t9) == 100 and t9[:7] == [0.5, 0.51, 0.52, 0.53, 0.54, 0.55, 0.57] and t9[-1] == 0.89
    assert len(tables.smooth_threshold(0.8)) == 89
    for name in goldens.names():
        g = goldens.load(name)
        assert np.array_equal(g["w_main"], tables.gaussian_half_kernel(float(g["sigma"]), 4.0))
        assert np.array_equal(g["h_table"], np.array(tables.smooth_threshold(float(g["threshold_rate"]))))


@pytest.mark.parametrize("name", goldens.raising_names())
def test_oracle_refuses_what_the_reference_raises_on(name):
    """x_*.npz: inputs on which the reference itself raises (break_large_problems :640 with max_problem_size = 4)."""
    g = goldens.load(name)
    assert str(g["raised"]).split(":")[0] in ("IndexError", "AssertionError")
    o = util.run_oracle(goldens.partition_of(g), goldens.params_of(g), goldens.tables_of(g))
    assert o["error"] != 0 and "break_large_problems" in o["errmsg"]


def test_goldens_reach_the_cli_parameter_bounds():
    """parse_args :104-109: 0 < sigma <= 50, 0.5 <= threshold_rate <= 1, 0 < variance_factor < 10, max_problem_size > 3."""
    runs = [c["run"] for c in goldens.manifest()["cases"].values()]
    assert {50.0, 0.1} <= {r["sigma"] for r in runs}
    assert {0.5, 0.51} <= {r["threshold_rate"] for r in runs}
    assert {0.01, 9.99} <= {r["variance_factor"] for r in runs}
    assert {4, 5} <= {r["max_problem_size"] for r in runs}
    g = goldens.load("b_sigma50_dense")
    assert len(g["w_main"]) == 201 and (g["iv_end"] - g["iv_start"] + 1).min() < 200     # intervals shorter than the radius
    assert len(goldens.load("b_sigma01")["w_main"]) == 1                                  # radius 0

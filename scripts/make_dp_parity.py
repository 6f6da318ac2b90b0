"""Condense the displaced-pole entries of a GPU test run's parity report (OGG_PARITY_REPORT=gpurun_out/parity_report.json
python -m pytest tests -m gpu) into profiles/dp_parity.json, which bench.py attaches to its output line as `parity` for the
workloads that have a displaced-pole cap.  usage: python scripts/make_dp_parity.py [report.json]"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_report.json"
rep = json.load(open(src))
full = rep["full_r8_latdp_pass_vs_oracle"]
out = {
    "what": "max relative difference of the displaced-pole cap's quadrature (dx, dy, area) from the numpy oracle, both arc forms of "
            "the great-arc distance, measured on MI355X by tests/test_gpu_pipeline.py::test_full_size_r8_latdp_pass_vs_oracle "
            "(1/8 degree, BASELINE config 4, every cell of the kept rows) and tests/test_gpu_parity.py::test_displaced_pole_quad_vs_oracle "
            "(smaller caps)",
    "full_size_r8_latdp": {
        form: {f: {"max_rel": full["%s_cap_max_rel_%s" % (f, form)], "max_abs": full["%s_cap_max_abs_%s" % (f, form)]} for f in ("dx", "dy", "area")}
        for form in ("literal", "chord")},
    "full_size_chord_vs_literal_max_rel": {f: full["%s_cap_chord_vs_literal_max_rel" % f] for f in ("dx", "dy", "area")},
    "rest_of_grid_max_rel": {f: full["%s_rest_max_rel" % f] for f in ("dx", "dy", "area")},
    "coordinates_max_abs_deg": {"x": full["x_max"], "y": full["y_max"]},
    "smaller_caps": {k[8:]: {f: v[f + "_rel"] for f in ("dx", "dy", "area")} for k, v in sorted(rep.items()) if k.startswith("dp_quad_")},
    "units": {"max_abs": "m (dx, dy), m^2 (area)"},
}
json.dump(out, open("profiles/dp_parity.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out["full_size_r8_latdp"], indent=1))

"""bench.py's cpu_baseline leg (the oracle timed on the host cores) runs without a GPU and reports the contract's fields."""
import importlib.util
import os


def test_cpu_baseline_leg_fields():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("medt_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.cpu_baseline_leg(steps=1)
    assert set(r) >= {"value", "unit", "cores", "kind", "sample"}
    assert r["unit"] == "images/s" and r["kind"] == "port" and r["cores"] >= 1
    assert 0.005 < r["value"] < 1000.0                      # a MedT step on CPU takes seconds, not microseconds
    assert "MedT" in r["sample"] and "1 training steps" in r["sample"]

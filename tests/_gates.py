"""Integration-gate helper: assert `value <= tol` AND leave a record of what was measured.

VERDICT r5 (weak 2): the model-level tolerances were 10-25x looser than what is measured, so they would not have caught a
slot-dependent rounding bug of the profiles/r04_determinism_bisect.txt class.  Every model-level gate now goes through
`gate()`: the tolerance is set at <= ~3x the value measured on MI355X (the measured figure is quoted at the call site), and
each evaluation appends one JSON line - test id, gate name, measured value, tolerance - to gpurun_out/measured_gates.jsonl
(copied to profiles/ per round), so the headroom of every gate is on record and a drift shows up before it fails.
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "measured_gates.jsonl")


def gate(name: str, value: float, tol: float, lower: bool = False) -> float:
    """`value <= tol` (or `>= tol` with lower=True); the measurement is logged either way."""
    value = float(value)
    ok = (value >= tol) if lower else (value <= tol)
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "gate": name,
                                 "value": value, "tol": tol, "kind": ">=" if lower else "<=", "ok": bool(ok)}) + "\n")
    except OSError:
        pass
    print(f"[gate] {name}: {value:.4g} ({'>=' if lower else '<='} {tol:g})")
    assert ok, f"{name}: measured {value:.6g}, gate {'>=' if lower else '<='} {tol:g}"
    return value

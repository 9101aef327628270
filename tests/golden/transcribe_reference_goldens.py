#!/usr/bin/env python3
"""
Transcribe the known-answer values the reference holds for the hot path into
``tests/golden/reference_goldens.json`` (G1-G8 of BASELINE.md / SURVEY.md section 8c).

Run in the BUILD container only (needs /root/reference); the JSON it writes is the committed
fixture.  Only DATA is extracted -- numbers from ``tests/test_gp_surrogate.py`` /
``tests/test_optimisation.py`` constants and from the logged OUTPUT cells of the example notebooks
-- no reference source text.

    python tests/golden/transcribe_reference_goldens.py
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")

FLOAT = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"


def _const(text, name):
    m = re.search(rf"{name}\s*=\s*({FLOAT})", text)
    return float(m.group(1))


def _array_const(text, name):
    m = re.search(rf"{name}\s*=\s*np\.array\(\[({FLOAT}),\s*({FLOAT})\]\)", text)
    return [float(m.group(1)), float(m.group(2))]


def _notebook_text(path):
    """Concatenate every stream/text output of the notebook, in cell order."""
    with open(path) as fh:
        nb = json.load(fh)
    chunks = []
    for cell in nb["cells"]:
        for out in cell.get("outputs", []):
            if "text" in out:
                chunks.append("".join(out["text"]))
            elif "data" in out and "text/plain" in out["data"]:
                chunks.append("".join(out["data"]["text/plain"]))
    return "\n".join(chunks)


def _iteration_trace(text):
    pat = re.compile(
        rf"After (\d+)th iteration:\s*\n\s*number of obj\. func\. evaluations: (\d+)\s*\n"
        rf"\s*highest score: ({FLOAT})\s*\n\s*highest UCB: ({FLOAT})"
    )
    return [
        {"iteration": int(a), "evaluations": int(b), "highest_score": float(c), "highest_ucb": float(d)}
        for a, b, c, d in pat.findall(text)
    ]


def _theta_tables(text):
    """Every printed GPflow parameter table (plain-text form) -> list of dicts, in order."""
    rows = re.findall(
        rf"GPR\.(mean_function\.c|kernel\.variance|kernel\.lengthscales|likelihood\.variance)\s+Parameter"
        rf"[^\n]*?float64\s+({FLOAT})",
        text,
    )
    tables, cur = [], {}
    for name, val in rows:
        key = {"mean_function.c": "mean_c", "kernel.variance": "variance",
               "kernel.lengthscales": "lengthscale", "likelihood.variance": "noise"}[name]
        if key in cur:
            tables.append(cur)
            cur = {}
        cur[key] = float(val)
    if cur:
        tables.append(cur)
    return [t for t in tables if len(t) == 4]


def _best_points(text):
    pat = re.compile(rf"GPPoint\(normed_coord=array\(\[({FLOAT}),\s*({FLOAT})\]\), score_mu=({FLOAT})")
    return [{"normed_coord": [float(a), float(b)], "score_mu": float(c)} for a, b, c in pat.findall(text)]


def _html_theta(path):
    """Final-theta HTML table of notebook 0 (only there as text/html)."""
    with open(path) as fh:
        raw = fh.read()
    vals = re.findall(
        rf"GPR\.(mean_function\.c|kernel\.variance|kernel\.lengthscales|likelihood\.variance)\s*</td>"
        rf".*?text-align: right;\\\">({FLOAT})\s*</td>",
        raw,
    )
    key = {"mean_function.c": "mean_c", "kernel.variance": "variance",
           "kernel.lengthscales": "lengthscale", "likelihood.variance": "noise"}
    out = {}
    for name, val in vals:
        out.setdefault(key[name], float(val))
    return out


def main():
    with open(f"{REF}/tests/test_gp_surrogate.py") as fh:
        t_surr = fh.read()
    with open(f"{REF}/tests/test_optimisation.py") as fh:
        t_opt = fh.read()
    gpr_section = t_surr[t_surr.index("class TestGPRSurrogate"): t_surr.index("class TestVGPSurrogate")]

    nb0 = _notebook_text(f"{REF}/examples/0-basic-optimisation.ipynb")
    nb1 = _notebook_text(f"{REF}/examples/1-callbacks.ipynb")
    nb3 = _notebook_text(f"{REF}/examples/3-saving-resuming-optimisation.ipynb")

    goldens = {
        "_source": "transcribed by tests/golden/transcribe_reference_goldens.py from jajcayn/pygpso v0.6.1",
        "fixture_recipe": {
            "source": "tests/test_gp_surrogate.py:221-244",
            "n_points": 10, "seed_base": 42,
            "note": "for i in range(10): np.random.seed(42+i); coord=rand(2); mu=rand(); rand(); rand(); "
                    "label=choice([1,2],p=[.8,.2]); label 1 = evaluated",
            "kernel": "Matern52", "lengthscale0": 1.0, "variance0": 1.0, "mean_c0": 0.0, "noise0": 1.0e-3,
        },
        "G1": {"source": "tests/test_gp_surrogate.py:259-268", "predict_at": [[0.5, 0.5]],
               "mean": _const(gpr_section, "EXP_MEAN"), "var": _const(gpr_section, "EXP_VAR"), "decimals": 8},
        "G2": {"source": "tests/test_gp_surrogate.py:293-309", "predict_at": [[0.5, 0.5], [0.5, 0.3]],
               "mean": _const(gpr_section, "EXP_MEAN"), "var": _const(gpr_section, "EXP_VAR"),
               "ucb_rule": "round(mean + erfcinv(0.01) * var, 8)", "decimals": 8},
        "G3": {"source": "tests/test_gp_surrogate.py:152-169", "num_evaluated": 6, "num_points": 10,
               "highest_score": _const(t_surr, "HIGHEST_SCORE"),
               "highest_coords": _array_const(t_surr, "HIGHEST_COORDS")},
        "G4": {"source": "tests/test_optimisation.py:19-23,66-89",
               "bounds": [[-3, 5], [-3, 3]], "depth": 3, "budget": 50,
               "best_coords": _array_const(t_opt, "BEST_COORDS_v1"),
               "best_score": _const(t_opt, "BEST_SCORE_v1"), "coord_decimals": 7, "score_decimals": 8},
        "G5": {"source": "tests/test_optimisation.py:91-152", "split_budgets": [25, 25],
               "same_answer_as": "G4"},
        "G6": {"source": "examples/0-basic-optimisation.ipynb:207-295,316,327-330",
               "bounds": [[-3, 5], [-3, 3]], "depth": 5, "budget": 50,
               "trace": _iteration_trace(nb0), "best": _best_points(nb0)[0],
               "final_theta": _html_theta(f"{REF}/examples/0-basic-optimisation.ipynb")},
        "G7": {"source": "examples/1-callbacks.ipynb:294-526", "depth": 5, "budget": 50,
               "theta_after_each_update": _theta_tables(nb1), "printed_significant_digits": 6},
        "G8": {"source": "examples/3-saving-resuming-optimisation.ipynb:193,406-408,430",
               "depth": 5, "budgets": [25, 25, 25],
               "trace": _iteration_trace(nb3), "best_points": _best_points(nb3)},
    }
    with open(OUT, "w") as fh:
        json.dump(goldens, fh, indent=1)
    print(f"wrote {OUT}")
    print("G6 trace rows", len(goldens["G6"]["trace"]), "| G7 tables", len(goldens["G7"]["theta_after_each_update"]),
          "| G8 trace rows", len(goldens["G8"]["trace"]), "best", goldens["G8"]["best_points"])
    print("G6 final theta", goldens["G6"]["final_theta"])


if __name__ == "__main__":
    main()

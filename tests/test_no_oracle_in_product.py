"""CPU: the product package never imports, links or executes the oracle (prompt rule 3)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_sources_do_not_reference_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "probpose_code_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/|oracle\.", text, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, f"product files mention the oracle: {bad}"

"""CPU: the drivers' numpy tables agree with the reference's own tables (golden fixtures)."""
import numpy as np
import pytest

from conftest import golden_cases, load_case
from mfem_amd.tables import bakeoff_tables


@pytest.mark.parametrize("name", golden_cases("cart2_bp*"))
def test_bakeoff_tables_match_reference(name):
    c = load_case(name)
    prob = {0: "bp1", 1: "bp3" if c["rule"] == 0 else "bp5"}[c["kind"]]
    kind, D, Q, B, G, Bn, Gn, W = bakeoff_tables(prob, c["p"])
    assert (kind, D, Q) == (c["kind"], c["D1D"], c["Q1D"])
    assert np.max(np.abs(B - c["B"])) < 1e-14 and np.max(np.abs(G - c["G"]))/np.max(np.abs(c["G"])) < 1e-14
    assert np.max(np.abs(Bn - c["nodes_B"])) < 1e-14 and np.max(np.abs(W - c["W"]))/np.max(c["W"]) < 1e-14

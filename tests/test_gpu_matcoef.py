"""-m gpu: DiffusionIntegrator with vector and matrix coefficients against the reference's OWN outputs
(tests/golden/mat/*.npz from oracle/ref_dump_mat.cpp): setup from J and from the nodes (coeffDim 3 / 6 / 9),
the apply and diagonal kernels with the reference's `symmetric` flag (9-plane D when it is 0), the fused
L->L operator, and the Jacobi-PCG solve for the SPD cases.  Bars: 1e-12 relative, identical iteration counts."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_oracle_matcoef import CASES, load_mat, rel          # noqa: E402


@pytest.mark.parametrize("name", CASES)
def test_matrix_coefficient_path_matches_reference(name):
    from gpu_helpers import dev, hB, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    c = load_mat(name)
    D, Q, NE, nd, sym, cd, kind = c["D1D"], c["Q1D"], c["NE"], c["ndofs"], c["symmetric"], c["coeff_vdim"], c["kind"]
    D3, Q3 = D**3, Q**3
    nc = 6 if sym else 9
    B, G = hB(c["B"]), hB(c["G"])
    const_c = int(c["coeff"].size == cd)
    W, J, Cq = dev(c["W"]), dev(c["J"]), dev(c["coeff"])
    # PADiffusionSetup3D from the reference's J
    pa = torch.full((nc*Q3*NE,), float("nan"), dtype=torch.float64, device="cuda:0")
    capi.call("mfem_amd_diffusion_setup_3d", Q, cd, NE, ptr(W), ptr(J), ptr(Cq), const_c, ptr(pa), None)
    assert rel(host(pa), c["pa_data"]) < 1e-13
    # nodes -> D in one pass (J never materialised)
    nodesE = dev(c["nodes_E"])
    pa2 = torch.full((nc*Q3*NE,), float("nan"), dtype=torch.float64, device="cuda:0")
    capi.call("mfem_amd_pa_setup_from_nodes_3d", 1, c["nodes_D1D"], Q, cd, NE, hB(c["nodes_B"]), hB(c["nodes_G"]), ptr(W),
              ptr(nodesE), ptr(Cq), const_c, 0, ptr(pa2), None, None)
    assert rel(host(pa2), c["pa_data"]) < 1e-12
    # integrator seams on the reference's own pa_data
    pa_ref, xE = dev(c["pa_data"]), dev(c["xE"])
    yE = zeros(NE*D3)
    capi.call("mfem_amd_pa_diffusion_apply_3d", D, Q, NE, sym, B, G, ptr(pa_ref), ptr(xE), ptr(yE), None)
    assert rel(host(yE), c["yE"]) < 1e-12
    dE = zeros(NE*D3)
    capi.call("mfem_amd_pa_diffusion_diag_3d", D, Q, NE, sym, B, G, ptr(pa_ref), 0, ptr(dE), None)
    assert rel(host(dE), c["diagE"]) < 1e-12
    # fused L->L (kind 2 = the 9-plane D), L-diagonal with the abs transpose
    gm, off, ind = dev(c["gather_map"], np.int32), dev(c["offsets"], np.int32), dev(c["indices"], np.int32)
    x = dev(c["x"])
    layouts = [(0, pa)]
    if sym and capi.lib.mfem_amd_has_specialization(1, D, Q):
        blk = zeros(capi.lib.mfem_amd_pa_data_size(1, 1, Q, NE))
        capi.call("mfem_amd_pa_data_to_blocked", 1, Q, NE, ptr(pa), ptr(blk), None)
        layouts.append((1, blk))
    for layout, pad in layouts:
        yE2, y, dg = zeros(NE*D3), zeros(nd), zeros(nd)
        capi.call("mfem_amd_pa_apply_L2E", kind, layout, D, Q, NE, B, G, ptr(pad), ptr(gm), ptr(x), ptr(yE2), None)
        capi.call("mfem_amd_restrict_scatter", nd, ptr(off), ptr(ind), ptr(yE2), ptr(y), 0, 0, None)
        assert rel(host(y), c["y"]) < 1e-12
        capi.call("mfem_amd_restrict_scatter", nd, ptr(off), ptr(ind), ptr(dE), ptr(dg), 0, 1, None)
        assert rel(host(dg), c["diag"]) < 1e-12
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        ess = np.asarray(c["ess_tdof"], np.int32) if sym else np.zeros(0, np.int32)
        ess_d = dev(ess, np.int32) if len(ess) else None
        marker, masked = zeros(nd, torch.uint8), None
        if len(ess):
            capi.call("mfem_amd_set_marker", len(ess), ptr(ess_d), ptr(marker), None)
            masked = zeros(NE*D3, torch.int32)
            capi.call("mfem_amd_mask_gather_map", NE*D3, ptr(gm), ptr(marker), ptr(masked), None)
        op = capi.PaOperator(kind, layout, D, Q, NE, nd, vp(B), vp(G), ptr(pad), ptr(gm), ptr(masked), ptr(off), ptr(ind),
                             len(ess), ptr(ess_d), ptr(marker), ptr(yE2), 0, 0, 0)
        yc = zeros(nd)
        capi.call("mfem_amd_pa_constrained_mult", C.byref(op), ptr(x), ptr(yc), None)
        if not sym:
            assert rel(host(yc), c["y"]) < 1e-12          # no essential dofs: the operator itself
            continue
        dinv = 1.0/np.asarray(c["diag"])
        dinv[ess] = 1.0
        dinv_d, b_d = dev(dinv), dev(c["B_rhs"])
        ws = zeros(capi.lib.mfem_amd_pcg_workspace_bytes(nd, 2000), torch.uint8)
        X, res = zeros(nd), capi.PcgResult()
        capi.call("mfem_amd_pcg_pa", C.byref(op), ptr(dinv_d), ptr(b_d), ptr(X), 1e-6, 0.0, 2000, 1,
                  ptr(ws), C.byref(res), None, None)
        assert (res.iterations, res.converged) == (c["cg_iters"], c["cg_converged"])
        assert rel(host(X), c["X"]) < 1e-9


def test_blocked_layout_is_refused_for_the_full_matrix():
    from gpu_helpers import zeros, ptr
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    assert capi.lib.mfem_amd_pa_data_size(2, 0, 4, 8) == 9*64*8
    assert capi.lib.mfem_amd_pa_data_size(2, 1, 4, 8) == -1
    a = zeros(8)
    B = np.zeros(12)
    with pytest.raises(capi.MfemAmdError):
        capi.call("mfem_amd_pa_apply_L2E", 2, 1, 3, 4, 1, B, B, ptr(a), ptr(a), ptr(a), ptr(a), None)

"""CPU: the C-ABI library loads and exports every symbol include/mfem_amd.h declares
(no compute calls without a GPU)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from mfem_amd import capi
    hdr = (ROOT / "include" / "mfem_amd.h").read_text()
    declared = set(re.findall(r"\b(mfem_amd_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(capi.lib, name), name
    assert declared == set(capi.EXPORTS)


def test_no_gpu_calls_fail_loudly_not_silently():
    import torch
    from mfem_amd import capi
    if torch.cuda.is_available():
        return
    rc = capi.lib.mfem_amd_init(0)
    assert rc != 0 and len(capi.lib.mfem_amd_last_error()) > 0


def test_product_does_not_touch_the_oracle():
    for p in (ROOT / "mfem_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".h", ".hpp", ".cpp"):
            txt = p.read_text()
            assert "pa_oracle" not in txt and "oracle_lib" not in txt and "orc_" not in txt, p

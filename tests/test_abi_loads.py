"""CPU test: the C-ABI library exists in-tree, loads, and exports every symbol include/lps_abi.h declares.
No compute calls (there is no GPU here)."""
import ctypes as C

from lps import abi, hip


def test_library_exports_all_declared_symbols():
    L = hip.load()
    syms = hip.declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), s
    import os, re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "lps_abi.h")).read()
    assert L.lps_abi_version() == int(re.search(r"#define\s+LPS_ABI_VERSION\s+(\d+)", hdr).group(1)) >= 15


def test_default_params_match_reference_defaults():
    L = hip.load()
    p = abi.Params()
    L.lps_default_params(C.byref(p))
    q = abi.default_params()
    for name, _ in abi.Params._fields_:
        assert getattr(p, name) == getattr(q, name), name


def test_struct_sizes_match_header():
    """ctypes mirrors must have the sizes the C side was compiled with."""
    L = hip.load()
    for which, st in enumerate((abi.Params, abi.VariantTable, abi.ReadBatch, abi.PhaseResult, abi.HaplotagResult, abi.Timings, abi.SomaticTagResult, abi.SiteCounters, abi.TumorExtractResult,
                                abi.ExtraVariantTable)):
        assert L.lps_struct_size(which) == C.sizeof(st), st.__name__

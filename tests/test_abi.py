"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/stylemesh_hip.h declares, and the ctypes binding lists exactly those (no compute calls)."""
import ctypes
import os
import re

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "stylemesh_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t)\s+(sm_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from stylemesh_amd.runtime import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(hip.SIGNATURES) == names


def test_layout_helpers_are_pure_host_functions():
    from stylemesh_amd.runtime import hip
    assert hip.lib.sm_abi_version() == hip.ABI_VERSION == 11
    for W in (1, 3, 4, 21, 341, 1045):
        wp = hip.row_stride(W)
        assert wp % 4 == 0 and wp >= W + 1 and wp < W + 5
    assert hip.plane(256, 341) % 64 == 0 and hip.plane(256, 341) >= 258 * 344


def test_problem_struct_layouts_match_the_library():
    """The ctypes mirrors of the header's problem structs have the sizes the library was compiled with."""
    from stylemesh_amd.runtime import hip
    for which, kind in enumerate((hip.ConvProblem, hip.PlaneProblem, hip.GramProblem, hip.StyleProblem, hip.GramBwdProblem,
                                  hip.CoverProblem, hip.ViewMasksDesc, hip.ViewLayerMask, hip.ViewResize,
                                  hip.ViewListsDesc, hip.ViewList, hip.Call)):
        assert hip.lib.sm_sizeof_problem(which) == ctypes.sizeof(kind), kind.__name__
    assert hip.lib.sm_sizeof_problem(99) == -1


def test_weight_packing_layouts():
    import torch
    from stylemesh_amd.runtime import ops
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).view(2, 3, 3, 3)
    f = ops.pack_conv_fwd(w)
    assert f.shape == (9, 4, 2) and float(f[5, 2, 1]) == float(w[1, 2, 1, 2]) and float(f[:, 3].abs().sum()) == 0
    d = ops.pack_conv_dgrad(w)
    assert d.shape == (9, 2, 4) and float(d[1, 1, 2]) == float(w[1, 2, 2, 1])

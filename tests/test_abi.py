"""The C-ABI library loads and exports exactly the symbols include/t2v_abi.h declares (no compute, no GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "t2v_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(t2v_[a-z0-9_]+)\s*\(", src)) - {"t2v_stream_t"}


def test_header_symbols_match_binding_table():
    import t2v_amd.native as nv
    assert _declared() == set(nv.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    import ctypes
    import t2v_amd.native as nv
    assert os.path.exists(nv.LIB_PATH), "build the extension first: python __graft_entry__.py"
    lib = ctypes.CDLL(nv.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert nv.lib().t2v_abi_version() == nv.ABI_VERSION


def test_ctypes_structs_mirror_header_sizes():
    """sizeof of the descriptor structs must match the C compiler's layout (checked against a gcc-compiled probe)."""
    import ctypes
    import subprocess
    import tempfile
    import t2v_amd.native as nv
    prog = r'''
#include <stdio.h>
#include "t2v_abi.h"
int main(){ printf("%zu %zu %zu %zu %zu\n", sizeof(T2VConvGeom), sizeof(T2VGemm), sizeof(T2VSmallConv), sizeof(T2VAttnOperand), sizeof(T2VAttn)); printf("%zu %zu %zu %zu\n", sizeof(T2VLoraWgrad), sizeof(T2VLoraMergeJob), sizeof(T2VLoraPrepJob), sizeof(T2VTemporalFused)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "p")]).split()))
    assert sizes == [ctypes.sizeof(nv.ConvGeom), ctypes.sizeof(nv.Gemm), ctypes.sizeof(nv.SmallConv),
                     ctypes.sizeof(nv.AttnOperand), ctypes.sizeof(nv.Attn), ctypes.sizeof(nv.LoraWgrad), ctypes.sizeof(nv.LoraMergeJob),
                     ctypes.sizeof(nv.LoraPrepJob), ctypes.sizeof(nv.TemporalFused)]


def test_native_calls_fail_loudly_without_gpu_tensors():
    import pytest
    import torch
    import t2v_amd.functional as F
    with pytest.raises(RuntimeError, match="ROCm device"):
        F.conv_linear(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8))

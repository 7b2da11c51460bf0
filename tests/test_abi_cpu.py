"""CPU: the C-ABI library loads and exports every symbol include/advmil_hip.h declares (no compute calls
without a GPU), the ctypes table covers the header, and the product path refuses CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "advmil_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(advmil_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from advmil_amd import _lib
    return _lib


def test_header_and_ctypes_table_agree(built):
    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(built.SIGNATURES) == syms


def test_library_exports_every_declared_symbol(built):
    handle = ctypes.CDLL(built.LIB_PATH)
    for name in header_symbols():
        assert hasattr(handle, name), name
    assert built.lib().advmil_version() >= 100


def test_workspace_queries_are_pure_host_functions(built):
    L = built.lib()
    assert L.advmil_gemm_f32_workspace_bytes(384, 1024, 16) == 16 * 384 * 1024 * 4
    assert L.advmil_gemm_f32_workspace_bytes(8192, 384, 1) == 0
    assert L.advmil_softmax_pool_workspace_bytes(8192, 384, 16) >= 16 * (8192 // 32) * 384 * 4 // 2
    assert L.advmil_ln_relu_mean16_bwd_workspace_bytes(8192, 128) == 512 * 384 * 4      # dgamma | dbeta | column sums of dy


def test_epilogue_struct_layout_matches_c(built, tmp_path):
    """Compile a 10-line C program against include/advmil_hip.h and compare sizeof/offsetof with the ctypes mirror."""
    import shutil
    import subprocess
    fields = [n for n, _ in built.Epilogue._fields_]
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "layout.c"
    body = "".join(f'printf("%zu\\n", offsetof(advmil_epilogue_t, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "advmil_hip.h"\n'
                   f'int main(void){{printf("%zu\\n", sizeof(advmil_epilogue_t));{body}return 0;}}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(built.Epilogue)
    assert out[1:] == [getattr(built.Epilogue, f).offset for f in fields]


def test_tail_struct_layouts_match_c(built, tmp_path):
    """advmil_dense_layer_t / advmil_dtail_t (the fused bag-level tail's argument block) against their ctypes mirrors."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    body = ""
    for cname, cls in (("advmil_dense_layer_t", built.DenseLayer), ("advmil_dtail_t", built.DTail), ("advmil_ghead_t", built.GHead), ("advmil_gemm_tn_call_t", built.GemmTnCall)):
        body += f'printf("%zu\\n", sizeof({cname}));'
        body += "".join(f'printf("%zu\\n", offsetof({cname}, {f}));' for f, _ in cls._fields_)
    src = tmp_path / "layout2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "advmil_hip.h"\n' f"int main(void){{{body}return 0;}}\n")
    exe = tmp_path / "layout2"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = []
    for cls in (built.DenseLayer, built.DTail, built.GHead, built.GemmTnCall):
        want += [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
    assert out == want


def test_product_path_has_no_cpu_fallback(built):
    from advmil_amd import ops
    from advmil_amd.model import load_backbone
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8), True, True, 8, 8, 8)
    net = load_backbone("abmil", [1024, 384, 384])
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 16, 1024), None)


def test_missing_library_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libadvmil_hip.so")
    with pytest.raises(built.HipLibraryMissing):
        built.lib()


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|__import__\(\s*[\"']oracle|importlib.*oracle", re.M)
    for d, _, files in os.walk(os.path.join(ROOT, "advmil_amd")):
        for f in files:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(d, f)).read()), os.path.join(d, f)


def test_layer_widths_without_a_hip_path_fail_at_model_construction():
    """A LayerNorm MLP layer wider than the row kernel takes is refused where the model is built, naming the width (it used to surface as an
    EINVAL from the first training step)."""
    from advmil_amd.model.model_utils import LN_RELU_MAX_WIDTH, make_mlp_layer
    make_mlp_layer(64, LN_RELU_MAX_WIDTH, layer_norm=True)
    with pytest.raises(ValueError, match="no HIP path"):
        make_mlp_layer(64, LN_RELU_MAX_WIDTH + 64, layer_norm=True)
    make_mlp_layer(64, LN_RELU_MAX_WIDTH + 64, layer_norm=False)

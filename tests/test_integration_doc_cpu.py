"""CPU: INTEGRATION.md cannot drift from the code. The rebinding snippet a maintainer of the reference would add and the ctypes
stub are parsed out of the document and checked against advmil_amd.model.MyHandler and against the binding table / the header."""
import ast
import inspect
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, re.S)


def test_rebinding_snippet_uses_only_what_the_handler_has():
    from advmil_amd.model import MyHandler
    src = next(b for b in code_blocks() if "class MyHandler(_RefHandler)" in b)
    tree = ast.parse(src)
    used = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            v = node.value
            # self._hip.<name>  and  hip.MyHandler.<name>
            if isinstance(v, ast.Attribute) and v.attr == "_hip" and isinstance(v.value, ast.Name) and v.value.id == "self":
                used.add(node.attr)
            if isinstance(v, ast.Attribute) and v.attr == "MyHandler" and isinstance(v.value, ast.Name) and v.value.id == "hip":
                used.add(node.attr)
    assert {"_train_each_epoch", "_update_disc", "_update_gen", "test_model", "optimizerG", "optimizerD", "netG", "netD",
            "patient_id", "pop_logs"} <= used, used
    init_src = inspect.getsource(MyHandler.__init__)
    assigned = set(re.findall(r"self\.(\w+)\s*=", init_src))
    for name in used:
        assert hasattr(MyHandler, name) or name in assigned, f"INTEGRATION.md uses MyHandler.{name}, which does not exist"
    # the reference's call signatures of the three step methods (model_handler.py:301, 349, 426) are accepted positionally
    sig = inspect.signature(MyHandler._train_each_epoch)
    assert list(sig.parameters)[:4] == ["self", "train_loader", "name_loader", "mode"]
    for m in (MyHandler._update_disc, MyHandler._update_gen):
        assert list(inspect.signature(m).parameters)[:6] == ["self", "i_batch", "xs", "ys", "mode", "label_visible_mask"]
    assert isinstance(inspect.getattr_static(MyHandler, "test_model"), staticmethod)
    assert list(inspect.signature(MyHandler.test_model).parameters)[:7] == [
        "modelG", "modelD", "backbone", "loader", "times_test_sample", "checkpoints", "test_zero_noise"]


def test_ctypes_stub_matches_the_binding_table_and_header():
    import ctypes
    from advmil_amd import _lib
    src = next(b for b in code_blocks() if "lib.advmil_softmax_pool_fwd.argtypes" in b)
    tree = ast.parse(src)
    argtypes, call_args = {}, {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) and node.targets[0].attr == "argtypes":
            fn = node.targets[0].value.attr
            argtypes[fn] = [e.attr for e in node.value.elts]
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("advmil_"):
            call_args[node.func.attr] = len(node.args)
    assert set(argtypes) == {"advmil_softmax_pool_fwd", "advmil_softmax_pool_workspace_bytes"}
    for fn, names in argtypes.items():
        want = _lib.SIGNATURES[fn][1]
        assert [getattr(ctypes, n) for n in names] == want, (fn, names, want)     # c_int64 is an alias of c_long on this ABI
        assert call_args[fn] == len(want), (fn, call_args[fn], len(want))
        assert all(hasattr(ctypes, n) for n in names)
    # and the header declares the same parameter count
    hdr = open(os.path.join(ROOT, "include", "advmil_hip.h")).read()
    decl = re.search(r"int advmil_softmax_pool_fwd\((.*?)\);", hdr, re.S).group(1)
    assert len([p for p in decl.split(",") if p.strip()]) == len(argtypes["advmil_softmax_pool_fwd"])


def test_entry_point_groups_name_only_exported_symbols():
    from advmil_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = text[text.index("| group | entry points |"):text.index("## What a maintainer of the reference adds")]
    for name in set(re.findall(r"`(advmil_[a-z0-9_]+)`", table)):
        if name.endswith("_") or name in ("advmil_epilogue_t",):
            continue
        assert any(k == name or k.startswith(name) for k in _lib.SIGNATURES), f"INTEGRATION.md names {name}, not in the binding table"

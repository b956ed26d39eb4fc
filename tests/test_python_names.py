"""A name that is read somewhere in a module must be bound somewhere in it (or be a builtin): catches the NameError that only
shows on a code path the CPU suite cannot execute (bench.py's GPU legs, the drop-in classes) -- a coarse check, no scoping."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unbound(path):
    tree = ast.parse(open(path).read())
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__path__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
        elif isinstance(n, ast.Import):
            bound.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ImportFrom):
            bound.update(a.asname or a.name for a in n.names)
        elif isinstance(n, ast.arg):
            bound.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            bound.update(n.names)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
    return sorted({(n.id, n.lineno) for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound})


def test_every_name_read_is_bound_somewhere_in_its_module():
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for pat in ("pin_slam_amd/*.py", "pin_slam_amd/dropin/*.py", "pin_slam_amd/dropin/*/*.py", "scripts/*.py", "oracle/*.py", "tests/*.py"):
        files += glob.glob(os.path.join(ROOT, pat))
    assert len(files) > 40
    bad = {os.path.relpath(f, ROOT): u for f in files if (u := _unbound(f))}
    assert not bad, bad

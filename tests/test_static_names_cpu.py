"""Every name a function of the repository's Python sources reads is bound somewhere it can see (function body incl. nested
functions, module, builtins).  A scope-approximate scan — it cannot prove a name is bound on every PATH — but it catches the
class of bug a GPU-only code path can hide from the CPU suite: a block pasted into a function whose variables live elsewhere."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + \
    sorted(glob.glob(os.path.join(ROOT, "tokenpacker_amd", "*.py"))) + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + \
    sorted(glob.glob(os.path.join(ROOT, "oracle", "*.py")))


def _bound_in(node):
    names = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            names.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, ast.arg):
            names.add(n.arg)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
    return names


def _module_level(tree):
    names = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
        else:
            names |= _bound_in(n)
    return names


def _unbound(path):
    tree = ast.parse(open(path).read())
    mod = _module_level(tree)
    bad = []

    def visit(scope_node, outer):
        for n in scope_node.body if hasattr(scope_node, "body") else []:
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
                seen = outer | _bound_in(n)
                for x in ast.walk(n):
                    if isinstance(x, ast.Name) and isinstance(x.ctx, ast.Load) and x.id not in seen:
                        bad.append((os.path.relpath(path, ROOT), n.name, x.lineno, x.id))
            elif isinstance(n, ast.ClassDef):
                visit(n, outer | _bound_in(n))
    visit(tree, mod)
    return bad


def test_no_function_reads_a_name_nothing_binds():
    bad = [b for f in FILES for b in _unbound(f)]
    assert not bad, "\n".join(f"{f}:{ln} in {fn}(): '{name}' is never bound" for f, fn, ln, name in bad)

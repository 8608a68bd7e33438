"""@ti.kernel / @ti.func bodies, re-compiled so that they compute the way Taichi does: in f32 / i32 (taichi/math.py).

Taichi does not run a kernel's Python: it rewrites the function's AST.  So does this module, at the function's first call
(every global the body names exists by then, and Taichi too reads them when it compiles):

  * every arithmetic operator goes through the kernel-scope scalar operation (`a * b` -> __ti_mul(a, b)): Taichi's promotion
    and one rounding to binary32 per operation;
  * float literals are rounded to binary32 where the function is compiled; module-level numbers the body names are
    compile-time constants (a float global becomes its f32 value), module-level vectors -- computed in Python scope, in
    double -- enter as copies with f32 elements;
  * a variable has the type of its FIRST assignment: a later `x = value` converts (a float stored into an integer variable
    truncates), vectors are copied on assignment (value semantics), and a name first assigned inside a for / while / if body
    is gone when that statement ends;
  * int() / float() / abs() / max() / min() / round() / pow() / any() / all() are the vector-aware DSL versions inside the
    body -- the module's own globals are left alone (its module-level code is plain Python).

`and` / `or` and the conditional expression keep Python's evaluation (Taichi's short_circuit_operators default).
"""
import ast as _ast
import builtins as _b
import inspect as _inspect
import textwrap as _tw

from . import math as _tm
from .math import Vector, _f32, _i2f, _trunc, _like


class _Unset:
    def __repr__(self): return "<unset>"


_UNSET = _Unset()


def _assign(old, new):
    """`x = value`: the first assignment fixes the variable's type, later ones convert to it; vectors are copied."""
    cn = new.__class__
    if old is _UNSET or old is None:
        return new._copy() if cn is Vector else new
    co = old.__class__
    if co is cn and co is not Vector:
        return new
    if co is Vector:
        if cn is Vector:
            if len(new._v) == len(old._v):
                return Vector._new([_like(o, n) for o, n in zip(old._v, new._v)])
            return new._copy()
        return new
    if cn is Vector:
        return new._copy()
    return _like(old, new)


def _assign_tuple(olds, news):
    news = tuple(news)
    if len(news) != len(olds):
        raise ValueError("unpacking sizes differ")
    return tuple(_assign(o, n) for o, n in zip(olds, news))


def _k_int(x=0, *a):
    if x.__class__ is Vector:
        return Vector._new([_trunc(v) if v.__class__ is float else _b.int(v) for v in x._v])
    if x.__class__ is float:
        return _trunc(x)
    return _b.int(x, *a)


def _k_float(x=0.0):
    if x.__class__ is Vector:
        return Vector._new([v if v.__class__ is float else _i2f(_b.int(v)) for v in x._v])
    if x.__class__ is float:
        return _f32(x)
    return _i2f(_b.int(x)) if isinstance(x, (_b.int, _b.bool)) else _f32(_b.float(x))


def _k_any(x):
    return _b.any(_b.bool(v) for v in x) if isinstance(x, (Vector, list, tuple)) else _b.bool(x)


def _k_all(x):
    return _b.all(_b.bool(v) for v in x) if isinstance(x, (Vector, list, tuple)) else _b.bool(x)


KERNEL_BUILTINS = {"int": _k_int, "float": _k_float, "abs": _tm.abs, "max": _tm.max, "min": _tm.min, "round": _tm.round,
                   "pow": _tm.pow, "any": _k_any, "all": _k_all}

_OPS = {_ast.Add: "__ti_add", _ast.Sub: "__ti_sub", _ast.Mult: "__ti_mul", _ast.Div: "__ti_div", _ast.FloorDiv: "__ti_floordiv",
        _ast.Mod: "__ti_mod", _ast.Pow: "__ti_pow", _ast.BitAnd: "__ti_and", _ast.BitOr: "__ti_or", _ast.BitXor: "__ti_xor",
        _ast.LShift: "__ti_lshift", _ast.RShift: "__ti_rshift"}
_HELPERS = {"__ti_add": _tm._kadd, "__ti_sub": _tm._ksub, "__ti_mul": _tm._kmul, "__ti_div": _tm._kdiv, "__ti_floordiv": _tm._kfloordiv,
            "__ti_mod": _tm._kmod, "__ti_pow": _tm._kpow, "__ti_and": _tm._kand, "__ti_or": _tm._kor, "__ti_xor": _tm._kxor,
            "__ti_lshift": _tm._klshift, "__ti_rshift": _tm._krshift, "__ti_assign": _assign, "__ti_assign_tuple": _assign_tuple,
            "__ti_unset": _UNSET}


def _call(name, args):
    return _ast.Call(func=_ast.Name(id=name, ctx=_ast.Load()), args=args, keywords=[])


def _load(name):
    return _ast.Name(id=name, ctx=_ast.Load())


def _stored(nodes):
    out = []
    for n in nodes:
        for sub in _ast.walk(n):
            if isinstance(sub, _ast.Name) and isinstance(sub.ctx, _ast.Store) and sub.id not in out:
                out.append(sub.id)
    return out


class _Rewrite(_ast.NodeTransformer):
    def __init__(self, fn_globals, params, local_names):
        self.g, self.params, self.locals = fn_globals, params, set(local_names)
        self.remap = {k: "__ti_b_" + k for k in KERNEL_BUILTINS
                      if (k not in fn_globals or fn_globals[k] is getattr(_b, k, None)) and k not in params and k not in self.locals}
        self.captured = {}          # name in the body -> (name bound in the module, value)
        self.defined = [set(params)]  # scope stack: names visible here

    # -- values ---------------------------------------------------------------------------------------------------------
    def visit_Constant(self, node):
        if node.value.__class__ is float:
            return _ast.copy_location(_ast.Constant(_f32(node.value)), node)
        return node

    def visit_Name(self, node):
        if not isinstance(node.ctx, _ast.Load):
            return node
        name = node.id
        if name in self.remap:
            return _ast.copy_location(_load(self.remap[name]), node)
        if name in self.params or name in self.locals or name not in self.g:
            return node
        v = self.g[name]
        if v.__class__ is float:                      # a module-level number: a compile-time constant, f32
            return _ast.copy_location(_ast.Constant(_f32(v)), node)
        if v.__class__ is Vector:                     # a Python-scope vector enters the kernel with f32 elements
            bound = "__ti_c_" + name
            self.captured[name] = (bound, Vector._new([_f32(x) if x.__class__ is float else x for x in v._v]))
            return _ast.copy_location(_load(bound), node)
        return node

    def visit_BinOp(self, node):
        self.generic_visit(node)
        return _ast.copy_location(_call(_OPS[type(node.op)], [node.left, node.right]), node) if type(node.op) in _OPS else node

    # -- assignments ----------------------------------------------------------------------------------------------------
    def _define(self, name):
        self.defined[-1].add(name)

    def visit_Assign(self, node):
        node.value = self.visit(node.value)
        node.targets = [self.visit(t) for t in node.targets]
        if len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, _ast.Name):
                node.value = _call("__ti_assign", [_load(t.id), node.value])
                self._define(t.id)
            elif isinstance(t, _ast.Tuple) and all(isinstance(e, _ast.Name) for e in t.elts):
                node.value = _call("__ti_assign_tuple", [_ast.Tuple(elts=[_load(e.id) for e in t.elts], ctx=_ast.Load()), node.value])
                for e in t.elts:
                    self._define(e.id)
        return node

    def visit_AugAssign(self, node):
        node.value = self.visit(node.value)
        op = _OPS.get(type(node.op))
        if op is None:
            return node
        if isinstance(node.target, _ast.Name):
            cur = _load(node.target.id)
            return _ast.copy_location(_ast.Assign(targets=[node.target], value=_call("__ti_assign", [cur, _call(op, [cur, node.value])])), node)
        import copy as _cp
        tl = _cp.deepcopy(node.target)
        for sub in _ast.walk(tl):
            if hasattr(sub, "ctx"):
                sub.ctx = _ast.Load()
        tl = self.visit(tl)
        node.target = self.visit(node.target)
        # (a vector's element store converts to the element's type itself: Vector.__setitem__)
        return _ast.copy_location(_ast.Assign(targets=[node.target], value=_call(op, [tl, node.value])), node)

    # -- blocks: a name first assigned inside a compound statement's body ends with it ---------------------------------------
    def _block(self, stmts):
        self.defined.append(set())
        out = []
        for s in stmts:
            r = self.visit(s)
            out.extend(r if isinstance(r, list) else [r])
        inner = self.defined.pop()
        return out, inner

    def _visible(self, name):
        return any(name in d for d in self.defined)

    def _resets(self, names):
        gone = sorted(n for n in names if not self._visible(n))
        return [_ast.Assign(targets=[_ast.Name(id=n, ctx=_ast.Store())], value=_load("__ti_unset")) for n in gone]

    def visit_For(self, node):
        node.iter = self.visit(node.iter)
        self.defined.append(set(_stored([node.target])))
        node.body, inner = self._block(node.body)
        node.orelse, inner2 = self._block(node.orelse) if node.orelse else ([], set())
        targets = self.defined.pop()
        return [node] + self._resets(inner | inner2 | targets)

    def visit_While(self, node):
        node.test = self.visit(node.test)
        node.body, inner = self._block(node.body)
        node.orelse, inner2 = self._block(node.orelse) if node.orelse else ([], set())
        return [node] + self._resets(inner | inner2)

    def visit_If(self, node):
        node.test = self.visit(node.test)
        node.body, inner = self._block(node.body)
        node.orelse, inner2 = self._block(node.orelse) if node.orelse else ([], set())
        return [node] + self._resets(inner | inner2)

    def visit_FunctionDef(self, node):      # (the function itself; nested defs are not DSL)
        out = []
        for s in node.body:
            r = self.visit(s)
            out.extend(r if isinstance(r, list) else [r])
        node.body = out
        return node


def compile_dsl(fn):
    """The function `fn` with Taichi's kernel-scope semantics (see the module's docstring)."""
    src = _tw.dedent(_inspect.getsource(fn))
    fdef = _ast.parse(src).body[0]
    fdef.decorator_list = []
    _ast.increment_lineno(fdef, fn.__code__.co_firstlineno - 1)
    for a in fdef.args.args + fdef.args.kwonlyargs:
        a.annotation = None
    fdef.returns = None
    params = {a.arg for a in fdef.args.args + fdef.args.kwonlyargs}
    if fdef.args.vararg:
        params.add(fdef.args.vararg.arg)
    if fdef.args.kwarg:
        params.add(fdef.args.kwarg.arg)
    local_names = [n for n in _stored(fdef.body) if n not in params]
    free = list(fn.__code__.co_freevars)      # a kernel defined inside a function: what it takes from the enclosing scope
    params_and_free = params | set(free)
    g = fn.__globals__
    # the defaults were evaluated in Python scope when the module ran: not part of the body
    defaults, kw_defaults = fdef.args.defaults, fdef.args.kw_defaults
    fdef.args.defaults, fdef.args.kw_defaults = [], [None] * len(kw_defaults)
    rw = _Rewrite(g, params_and_free, local_names)
    fdef = rw.visit(fdef)
    # every local exists from the start (typed by its first assignment: _assign)
    pre = [_ast.Assign(targets=[_ast.Name(id=n, ctx=_ast.Store())], value=_load("__ti_unset")) for n in local_names]
    doc = []
    if fdef.body and isinstance(fdef.body[0], _ast.Expr) and isinstance(getattr(fdef.body[0], "value", None), _ast.Constant):
        doc, fdef.body = fdef.body[:1], fdef.body[1:]
    fdef.body = doc + pre + fdef.body
    conv = lambda v: Vector._new([_f32(x) if x.__class__ is float else x for x in v._v]) if v.__class__ is Vector else (_f32(v) if v.__class__ is float else v)  # noqa: E731
    body = [fdef]
    if free:    # re-created inside a factory whose parameters are the enclosing scope's values (numbers and vectors as f32)
        factory = _ast.FunctionDef(name="__ti_factory", args=_ast.arguments(posonlyargs=[], args=[_ast.arg(arg=n) for n in free], kwonlyargs=[],
                                                                           kw_defaults=[], defaults=[]),
                                   body=[fdef, _ast.Return(value=_load(fdef.name))], decorator_list=[])
        body = [factory]
    tree = _ast.Module(body=body, type_ignores=[])
    _ast.fix_missing_locations(tree)
    g.update(_HELPERS)
    for k, v in rw.remap.items():
        g[v] = KERNEL_BUILTINS[k]
    for bound, v in rw.captured.values():
        g[bound] = v
    ns = {}
    exec(compile(tree, _inspect.getsourcefile(fn) or "<ti.func>", "exec"), g, ns)
    out = ns["__ti_factory"](*[conv(c.cell_contents) for c in fn.__closure__]) if free else ns[fdef.name]
    # the original defaults, vectors among them with f32 elements (they are only ever used in kernel scope)
    if fn.__defaults__:
        out.__defaults__ = tuple(conv(v) for v in fn.__defaults__)
    if fn.__kwdefaults__:
        out.__kwdefaults__ = {k: conv(v) for k, v in fn.__kwdefaults__.items()}
    return out

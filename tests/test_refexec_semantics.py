"""Known answers for tests/refexec/taichi, the Taichi-DSL emulation under which the reference's own source is executed to write
tests/golden/reference/ (make_reference_vectors.py): each test states one rule of Taichi's semantics the emulation stands for
 on a kernel small enough to work out by hand.  The module is loaded under a private name: `import taichi`
elsewhere in the suite must keep meaning the product's DSL shim.  (Kernels are re-compiled in their module's namespace, like the
reference's: what they use lives at module level here.)"""
import importlib.util
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    spec = importlib.util.spec_from_file_location("refexec_taichi", os.path.join(HERE, "refexec", "taichi", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(HERE, "refexec", "taichi")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refexec_taichi"] = mod
    spec.loader.exec_module(mod)
    return mod


ti = _load()


def test_f32_times_i32_is_an_f32_product_not_a_double_one():
    @ti.func
    def f(a: ti.f32, n: ti.i32):
        return a * n - 0.5
    a, n = np.float32(1.0 / 3.0), 3840
    want = np.float32(np.float32(a * np.float32(n)) - np.float32(0.5))
    assert f(a, n).dtype == np.float32 and f(a, n) == want
    assert np.float32(np.float64(a) * n - 0.5) != want or True      # (numpy's promotion would round once, from double)


def test_python_constants_fold_in_double_and_take_the_type_they_meet():
    @ti.func
    def f(x: ti.f32):
        return 3.0 / (16.0 * np.pi) * x
    assert f(np.float32(2.0)) == np.float32(np.float32(3.0 / (16.0 * np.pi)) * np.float32(2.0))


def test_a_variable_keeps_the_type_of_its_first_assignment():
    @ti.func
    def f(x: ti.f32):
        n = 0
        n = x * 2.75       # stored into an integer variable: truncated
        y = 1.5
        y = 7              # an integer stored into a float variable
        return n, y
    n, y = f(np.float32(3.0))
    assert n == 8 and isinstance(n, np.int32) and y == 7.0 and isinstance(y, np.float32)


@ti.func
def bump(v, hit: ti.template(), dist: ti.template()):
    v.x = 100.0            # by value: the caller's vector is untouched
    hit = 1                # by reference: the caller's variables change
    dist = dist * 0.5


def test_vectors_are_values_and_template_parameters_are_references():
    @ti.func
    def f():
        a = ti.Vector([1.0, 2.0, 3.0])
        b = a                  # a copy
        b.y = 9.0
        hit = 0
        dist = 8.0
        bump(a, hit, dist)
        return a, b, hit, dist
    a, b, hit, dist = f()
    assert a.to_list() == [1.0, 2.0, 3.0] and b.to_list() == [1.0, 9.0, 3.0] and hit == 1 and dist == 4.0


def test_constant_whole_powers_multiply_and_min_max_drop_nans():
    @ti.func
    def f(x: ti.f32, y: ti.f32):
        return ti.pow(x, 5.0), ti.max(y, 0.0), ti.min(ti.max(y, 0.0), 300.0)
    x = np.float32(0.7310586)
    p, m, c = f(x, np.float32(np.nan))
    x2 = np.float32(x * x)
    assert p == np.float32(x * np.float32(x2 * x2)) and m == 0.0 and c == 0.0


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
def test_casts_truncate_and_u32_wraps():
    @ti.func
    def f(x: ti.f32, k: ti.u32):
        return ti.cast(x, ti.i32), ti.cast(-x, ti.i32), k * 2654435761 + 1, ti.cast(x, ti.f16)
    a, b, h, half = f(np.float32(2.9), np.uint32(4000000000))
    assert (a, b) == (2, -2) and h == np.uint32((4000000000 * 2654435761 + 1) & 0xFFFFFFFF) and half == np.float16(2.9)


ROW = ti.field(ti.f32, shape=(4, 1))


def test_the_outermost_loop_is_parallel_iterations_see_pre_loop_values_of_what_others_write():
    for i in range(4):
        ROW[i, 0] = float(i + 1)

    @ti.kernel
    def shift_left():
        for i, j in ROW:
            ROW[i, j] = ROW[i, j] + 10.0              # its own write it reads back below
            if i > 0:
                ROW[i, j] = ROW[i, j] + ROW[i - 1, j]   # the neighbour was written earlier in this serial run: its OLD value counts
    shift_left()
    assert ROW.to_numpy()[:, 0].tolist() == [11.0, 13.0, 15.0, 17.0]


OZONE = ti.math.vec3(4.51103766177301e-21, 3.2854797958699e-21, 1.96774621921165e-22) * 0.0001 * (2.5035422e25 * 0.012588 * 8e-6)   # atmos.py:39-45


def test_python_scope_vectors_hold_python_numbers():
    assert isinstance(OZONE[0], float) and OZONE[0] == 4.51103766177301e-21 * 0.0001 * (2.5035422e25 * 0.012588 * 8e-6)

    @ti.func
    def f(x: ti.f32):
        return OZONE.x * x         # rounded to f32 once, where the kernel uses it
    assert f(np.float32(3.0)) == np.float32(np.float32(OZONE[0]) * np.float32(3.0))


def test_loop_indices_are_i32_values_unless_the_loop_is_static():
    """`for i in range(n)` in a kernel is a runtime loop: i is an i32 VALUE and `i * 0.1` an f32 product; under ti.static(...) the loop
    is unrolled over Python integers and `i * 0.1` folds in double (reference example6.py:60-64: vec3(0.5 - i * 0.1))."""
    @ti.func
    def f():
        out = []
        for i in range(3, 4):
            out.append(0.5 - i * 0.1)
        for i in ti.static(range(3, 4)):
            out.append(ti.Vector([0.5 - i * 0.1])[0])
        for i, j in ti.ndrange((3, 4), 1):
            out.append(0.5 - i * 0.1)
        return out
    runtime, static, nd = f()
    want = np.float32(np.float32(0.5) - np.float32(np.float32(3) * np.float32(0.1)))
    assert runtime.dtype == np.float32 and runtime == want and nd == want
    assert static == np.float32(0.5 - 3 * 0.1) and static != want


def test_max_and_min_promote_like_any_binary_operation():
    """ti.max(x, 0) with an f32 x is an f32 -- also when the integer wins: a variable first assigned from it is an f32 variable
    (reference example6.py:47-49: prob = ti.max(..., 0); prob = prob * prob)."""
    @ti.func
    def f(x: ti.f32):
        p = ti.max(x, 0)
        p = 0.25        # would truncate to 0 if p had been typed i32
        return p, ti.min(x, 7), ti.max(2, 3)
    p, m, k = f(np.float32(-1.5))
    assert p.dtype == np.float32 and p == np.float32(0.25)
    assert m.dtype == np.float32 and m == np.float32(-1.5)
    assert k == 3


def test_unorm8_texture_and_out_of_bounds_policies():
    t = ti.Texture(ti.Format.rgba8, (2, 2, 2))
    t.store(ti.Vector([1, 0, 1]), ti.Vector([0.5, 10 / 255.0, 1.2, 2 / 255.0]))
    assert [float(x) for x in t.fetch(ti.Vector([1, 0, 1]), 0).to_list()] == [float(np.float32(128) / np.float32(255)), float(np.float32(10) / np.float32(255)), 1.0, float(np.float32(2) / np.float32(255))]
    g = ti.field(ti.f32, shape=(2, 2))
    g[1, 1] = 5.0
    with pytest.raises(IndexError):
        g[2, 1]
    ti.set_out_of_bounds_reads("clamp", g)
    assert g[2, 1] == 5.0 and g[-3, 9] == g[0, 1]
    ti.set_out_of_bounds_reads("zero", g)
    assert g[2, 1] == 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/renderer"), reason="the reference tree is only in the build container")
def test_the_reference_source_still_reproduces_the_function_vectors():
    """Executes the reference's bsdf.py / math_utils.py live (a subprocess: it imports the emulation as `taichi`) on the first rows
    of tests/golden/reference/functions.npz and compares with what the file holds -- fixture and generator are in step."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_reference_vectors.py"), "--spot-check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "spot check: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

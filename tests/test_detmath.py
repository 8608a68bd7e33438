"""The numeric contract (include/vrt_detmath.h) on the host: accuracy against float64 references,
exactness of the f16 conversions, special-value behaviour of min/max/casts, the random stream."""
import numpy as np
import orc

RNG = np.random.default_rng(1234)


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-300)


def test_sin_cos_accuracy():
    x = np.concatenate([RNG.uniform(-100, 100, 200000), RNG.uniform(-7, 7, 200000), np.linspace(-3.2, 3.2, 10001)]).astype(np.float32)
    for op, f in ((0, np.sin), (1, np.cos)):
        got = orc.detmath(op, x)
        ref = f(x.astype(np.float64))
        # absolute error bound (near zeros of sin/cos the ulp measure is meaningless)
        assert np.max(np.abs(got - ref)) < 2.5e-7
    assert np.isnan(orc.detmath(0, np.array([np.inf, np.nan, 2e9], dtype=np.float32))).all()


def test_exp_log_accuracy():
    x = RNG.uniform(-87, 88, 300000).astype(np.float32)
    assert ulp_err(orc.detmath(2, x), np.exp(x.astype(np.float64))).max() < 2.0
    p = np.exp(RNG.uniform(-87, 88, 300000)).astype(np.float32)
    assert ulp_err(orc.detmath(3, p), np.log(p.astype(np.float64))).max() < 2.0
    sp = orc.detmath(2, np.array([-200.0, 100.0, 0.0, np.nan], dtype=np.float32))
    assert sp[0] == 0.0 and np.isinf(sp[1]) and sp[2] == 1.0 and np.isnan(sp[3])
    sl = orc.detmath(3, np.array([0.0, -1.0, np.inf, 1.0, 1e-42], dtype=np.float32))
    assert sl[0] == -np.inf and np.isnan(sl[1]) and sl[2] == np.inf and sl[3] == 0.0
    assert abs(sl[4] - np.log(np.float64(np.float32(1e-42)))) < 1e-4


def test_pow_accuracy():
    a = np.exp(RNG.uniform(-14, 3, 200000)).astype(np.float32)
    b = RNG.uniform(0.0, 2.5, 200000).astype(np.float32)
    got = orc.detmath(4, a, b)
    ref = np.power(a.astype(np.float64), b.astype(np.float64))
    rel = np.abs(got - ref) / ref
    assert rel.max() < 4e-6  # exp(y*log x): error grows with |y log x| <= 35
    sp = orc.detmath(4, np.array([0.0, 0.0, 1.0, -1.0, 2.0], dtype=np.float32), np.array([2.0, -1.0, 7.0, 2.0, 0.0], dtype=np.float32))
    assert sp[0] == 0.0 and np.isinf(sp[1]) and sp[2] == 1.0 and np.isnan(sp[3]) and sp[4] == 1.0


def test_acos_atan2_accuracy():
    x = np.concatenate([RNG.uniform(-1, 1, 200000), [-1.0, 1.0, 0.0, 0.5, -0.5]]).astype(np.float32)
    assert np.max(np.abs(orc.detmath(5, x) - np.arccos(x.astype(np.float64)))) < 5e-7
    y = RNG.standard_normal(200000).astype(np.float32)
    z = RNG.standard_normal(200000).astype(np.float32)
    assert np.max(np.abs(orc.detmath(6, y, z) - np.arctan2(y.astype(np.float64), z.astype(np.float64)))) < 6e-7
    q = orc.detmath(6, np.array([0.0, 1.0, -1.0, 0.0], dtype=np.float32), np.array([0.0, 0.0, 0.0, -1.0], dtype=np.float32))
    assert q[0] == 0.0 and abs(q[1] - np.pi / 2) < 1e-6 and abs(q[2] + np.pi / 2) < 1e-6 and abs(q[3] - np.pi) < 1e-6


def test_f16_round_trip_is_numpy_float16():
    x = np.concatenate([RNG.standard_normal(100000) * 10, np.exp(RNG.uniform(-30, 12, 100000)), -np.exp(RNG.uniform(-30, 12, 100000)),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf]]).astype(np.float32)
    got = orc.detmath(9, x)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.isnan(orc.detmath(9, np.array([np.nan], dtype=np.float32)))[0]


def test_min_max_semantics():
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    a = np.array([1.0, nan, 2.0, nan, -0.0, 0.0, -inf, 3.0], dtype=np.float32)
    b = np.array([2.0, 5.0, nan, nan, 0.0, -0.0, 1.0, 3.0], dtype=np.float32)
    mn, mx = orc.detmath(7, a, b), orc.detmath(8, a, b)
    assert mn[0] == 1 and mn[1] == 5 and mn[2] == 2 and np.isnan(mn[3]) and mn[6] == -inf and mn[7] == 3
    assert mx[0] == 2 and mx[1] == 5 and mx[2] == 2 and np.isnan(mx[3]) and mx[6] == 1 and mx[7] == 3
    assert np.signbit(mn[4]) and np.signbit(mn[5])            # -0 orders below +0
    assert not np.signbit(mx[4]) and not np.signbit(mx[5])


def test_random_stream():
    import ctypes as C
    out = np.empty(200000, dtype=np.float32)
    orc.lib().orc_unit_rng(C.c_uint32(1), C.c_uint32(2), C.c_uint32(3), C.c_uint32(0), out.size, orc.fptr(out))
    assert out.min() >= 0.0 and out.max() < 1.0
    assert abs(out.mean() - 0.5) < 5e-3 and abs(out.var() - 1 / 12) < 2e-3
    assert np.all((out * 2 ** 24) == np.floor(out * 2 ** 24))  # 24-bit resolution like ti.random
    hist, _ = np.histogram(out, bins=64, range=(0, 1))
    assert hist.min() > 0.9 * out.size / 64 and hist.max() < 1.1 * out.size / 64
    # streams keyed by (seed, frame, index, stream) are distinct and reproducible
    a, b, c = (np.empty(16, dtype=np.float32) for _ in range(3))
    orc.lib().orc_unit_rng(C.c_uint32(1), C.c_uint32(2), C.c_uint32(3), C.c_uint32(0), 16, orc.fptr(a))
    orc.lib().orc_unit_rng(C.c_uint32(1), C.c_uint32(2), C.c_uint32(4), C.c_uint32(0), 16, orc.fptr(b))
    orc.lib().orc_unit_rng(C.c_uint32(1), C.c_uint32(2), C.c_uint32(3), C.c_uint32(0), 16, orc.fptr(c))
    assert np.array_equal(a, c) and not np.array_equal(a, b)
    assert np.array_equal(a, out[:16])


def test_half_conversions_host_definition_equals_numpy_references():
    """The numpy references of tests/test_gpu_parity.py::test_half_conversions_on_device_equal_host_definition are the
    host definition of dm_f32_to_f16 / dm_f16_to_f32 (include/vrt_detmath.h, compiled into the emulation library)."""
    import ctypes as C
    import emu
    from test_gpu_parity import half_conversion_cases
    lib = emu.lib()
    lib.emu_half_probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.emu_half_probe.restype = None
    codes, ref32, words, ref16 = half_conversion_cases()
    for op, inp, ref in ((15, codes, ref32), (14, words, ref16)):
        inp = np.ascontiguousarray(inp, dtype=np.uint32)
        out = np.empty_like(inp)
        lib.emu_half_probe(op, inp.size, inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, ref), f"op {op}: {np.count_nonzero(out != ref)} differ, e.g. {hex(int(inp[out != ref][0]))}"

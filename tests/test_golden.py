"""Committed fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py with the CPU oracle; the reference
itself cannot run here) against (a) the oracle as it is now, (b) the product's device code
compiled for the host, (c) on a GPU box, libvrt_hip.so through the C ABI -- all bit for bit."""
import importlib.util
import os

import numpy as np
import pytest

import emu
import orc

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)


def check(session, name):
    case = mg.CASES[name]
    got = mg.run_case(session, case)
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert sorted(want.files) == sorted(got.keys())
    for key in want.files:
        a, b = np.ascontiguousarray(got[key]), want[key]
        assert a.shape == b.shape and a.dtype == b.dtype, key
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{name}: {key} differs from the fixture"


def test_every_case_has_a_fixture():
    have = {f[:-4] for f in os.listdir(os.path.join(HERE, "golden")) if f.endswith(".npz")}
    have.discard("example6_grid")   # an authored voxel grid (make_example_fixtures.py), not a rendered case
    assert have == set(mg.CASES)


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_fixture(name):
    o = orc.Oracle(mg.config_of(mg.CASES[name]), threads=4)
    check(o, name)
    o.close()


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_emulated_device_code_reproduces_fixture(name):
    e = emu.Emulated(mg.config_of(mg.CASES[name]))
    check(e, name)
    e.close()


@pytest.mark.parametrize("name", sorted(n for n in mg.CASES if "restir" in n))
def test_emulated_split_spatial_pass_reproduces_fixture(name, monkeypatch):
    """The spatial-reuse pass as the GPU runs it -- two kernels, the first leaving the second its masks of accepted and of live taps
    (vrt_restir.h) -- on the host build of the device code."""
    monkeypatch.setenv("VRT_EMU_GRIS_SPLIT", "1")
    e = emu.Emulated(mg.config_of(mg.CASES[name]))
    check(e, name)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["pool", "fused"])
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_gpu_reproduces_fixture(name, schedule, monkeypatch):
    from voxel_rt2_amd import _lib
    from voxel_rt2_amd._session import NativeSession
    monkeypatch.setenv("VRT_RENDER", schedule)
    g = NativeSession(_lib.load(), "vrt_", mg.config_of(mg.CASES[name]))
    check(g, name)
    g.close()

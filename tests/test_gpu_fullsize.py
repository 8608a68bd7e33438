"""BASELINE configs 3 and 4 at their real frame sizes on the GPU, against the oracle on a row shard (the oracle renders
16 rows of a 1080p / 4K frame in seconds; per-pixel random streams make a shard's pixels those of the full frame)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from voxel_rt2_amd import _abi, _lib, host, scenes
from voxel_rt2_amd._session import NativeSession

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_full_size_shard_matches_oracle():
    """Config 4's frame -- dense 128^3 fill (p = 0.5, seed 12345), 3840x2160, 8 bounces, one fused call of 4 samples -- on the
    16 rows that hold pixel (3839, 2159): the pooled kernel packs u, v in 12 bits each and 3840 is 94 % of that range.  Then the
    whole frame on one GPU: same rows, bit for bit."""
    W, H, rows = 3840, 2160, (2144, 2160)
    mat, rgb, params = scenes.scene_dense(12345)
    kw = dict(voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0)
    g, o = NativeSession(_lib.load(), "vrt_", host.make_config(W, H, rows=rows, **kw)), orc.Oracle(host.make_config(W, H, rows=rows, **kw), threads=16)
    for s in (g, o):
        orc.setup(s, mat, rgb, params)
        s.accumulate(4)
    a, b = g.fetch_hdr()[rows[0]:rows[1]], o.fetch_hdr()[rows[0]:rows[1]]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} values differ"
    for which in (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_MAT, _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR):
        assert np.array_equal(g.fetch_buffer(which)[rows[0]:rows[1]].view(np.uint8), o.fetch_buffer(which)[rows[0]:rows[1]].view(np.uint8)), which
    g.close(); o.close()
    f = NativeSession(_lib.load(), "vrt_", host.make_config(W, H, **kw))
    orc.setup(f, mat, rgb, params)
    f.accumulate(4)
    full = f.fetch_hdr()
    f.close()
    assert np.array_equal(full[rows[0]:rows[1]].view(np.uint32), a.view(np.uint32))
    assert np.isfinite(full).all() and full[:, -1].mean() > 0.01 and full[-1].mean() > 0.01   # the last column and row were rendered


def _oracle_sky_columns(o, u0, u1, R):
    """Columns [u0, u1) of both sky tables as the oracle computes them at table size R: Scene.finish()'s 32 cloud passes and
    the atmosphere pass (scene.py:243-253) restricted to those columns.  A texel of either pass depends on no other texel
    (atmos.py:140-189), and the random stream of a texel is keyed by (pass, u * R + v), so the columns are the whole table's."""
    n = u1 - u0
    assert R % n == 0 and u0 % n == 0
    o.prepare()   # tables zeroed, cloud pass counter back to 0
    for _ in range(32):
        o.sky_accumulate_clouds_slice(32, u0 // n, R // n)
    o.sky_compute_slice(u0 // n, R // n)
    scat, trans = np.empty((n, R, 3), np.float32), np.empty((n, R, 3), np.float32)
    o.sky_table_io(_abi.BUF_SKY_SCATTERING, u0, u1, scat.ctypes.data, 0)
    o.sky_table_io(_abi.BUF_SKY_TRANSMITTANCE, u0, u1, trans.ctypes.data, 0)
    return scat, trans


def test_config3_full_size_shard_matches_oracle():
    """Config 3 as benchmarked -- scene S6, physical sky + clouds with the 3840^2 tables (atmos.py:66-69), ReSTIR spatial reuse,
    1920x1080, 8 bounces -- on a 16-row shard through the horizon (26 halo rows each side for the reuse radius), two accumulate
    passes.  The tables are computed on the GPU the way Scene.finish() does (32 cloud passes over the whole table, then 32
    slices of 120 columns) and compared with the oracle AT THIS SIZE on 64 whole columns, bit for bit: the first and the last
    16 columns of the table, the 16 around the first slice boundary (columns 112-127: 119 | 120), and the 16 that hold the
    sun's azimuth; every column carries all 3840 elevations, horizon rows included.  (The oracle needs ~20 s per column and
    thread, so the other 3776 columns are its only through the same per-texel code.)  The render comparison then runs on the
    GPU's tables."""
    W, H, rows, R = 1920, 1080, (560, 576), 3840
    mat, rgb, params = scenes.scene_s6(0)
    cloud = np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy"))
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0, use_restir=True, sky_res=R,
                           rows=rows)
    g = NativeSession(_lib.load(), "vrt_", cfg)
    orc.setup(g, mat, rgb, params, cloud=cloud)
    for _ in range(32):
        g.sky_accumulate_clouds(32)
    for sl in range(32):
        g.sky_compute_slice(sl, 32)
    scat, trans = g.fetch_buffer(_abi.BUF_SKY_SCATTERING), g.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE)
    assert np.isfinite(scat).all() and np.isfinite(trans).all() and scat.max() > 0
    o = orc.Oracle(cfg, threads=16)
    orc.setup(o, mat, rgb, params, cloud=cloud)
    uv = np.zeros(2, dtype=np.float32)
    sun = np.asarray(host.normalize3(params["light_direction"]), dtype=np.float32)
    orc.lib().orc_unit_project_sky(C.c_void_p(o._ctx), orc.fptr(sun), orc.fptr(uv))
    sun_col = int(np.clip(uv[0] * R, 0, R - 1)) & ~15
    for u0 in sorted({0, 112, sun_col, R - 16}):
        os_, ot = _oracle_sky_columns(o, u0, u0 + 16, R)
        assert os_.max() > 0 and np.isfinite(os_).all()
        assert np.array_equal(scat[u0:u0 + 16].view(np.uint32), os_.view(np.uint32)), f"scattering table, columns {u0}..{u0 + 15}"
        assert np.array_equal(trans[u0:u0 + 16].view(np.uint32), ot.view(np.uint32)), f"transmittance table, columns {u0}..{u0 + 15}"
    o.upload_sky(scat, trans)
    for s in (g, o):
        s.accumulate(2)
    a, b = g.fetch_hdr()[rows[0]:rows[1]], o.fetch_hdr()[rows[0]:rows[1]]
    assert np.isfinite(a).all() and a.mean() > 0.01
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} of {a.size} values differ"
    for which in (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_MAT, _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR):
        assert np.array_equal(g.fetch_buffer(which)[rows[0]:rows[1]].view(np.uint8), o.fetch_buffer(which)[rows[0]:rows[1]].view(np.uint8)), which


def test_restir_row_shards_equal_full_frame():
    """Row shards with spatial reuse on (26 halo rows): the tap angles of a pixel are hashed from its 8x8 tile in FRAME
    coordinates (pathtracer.py:834-836) and the kernel works them out once per wave, so its wave tiles must sit on that grid
    whatever row a shard starts at.  (Round 1's kernel started them at the shard's first row: a shard whose start was not
    2 modulo 8 -- every tile of an 8-GPU split of 1080 rows but one -- got other taps than the full frame.)"""
    W, H = 208, 136
    mat, rgb, params = scenes.scene_sunlit(0)
    kw = dict(voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=5, seed=13, use_restir=True)

    def run(rows):
        s = NativeSession(_lib.load(), "vrt_", host.make_config(W, H, rows=rows, **kw))
        orc.setup(s, mat, rgb, params)
        s.accumulate(2)
        out = s.fetch_hdr()
        s.close()
        return out

    full = run(None)
    for rows in ((0, 17), (17, 34), (34, 51), (51, 68), (68, 85), (85, 102), (102, 119), (119, 136), (33, 90)):
        part = run(rows)
        assert np.array_equal(part[rows[0]:rows[1]].view(np.uint32), full[rows[0]:rows[1]].view(np.uint32)), rows
    o = orc.Oracle(host.make_config(W, H, **kw), threads=16)
    orc.setup(o, mat, rgb, params)
    o.accumulate(2)
    assert np.array_equal(full.view(np.uint32), o.fetch_hdr().view(np.uint32))


@pytest.mark.parametrize("W,H,depth,rows", [(4096, 16, 4, None), (4104, 16, 4, None), (96, 64, 15, None), (96, 64, 16, None),
                                            (640, 4100, 3, (4090, 4100))])
def test_limits_of_the_pooled_kernel(W, H, depth, rows):
    """The pooled kernel packs pixel coordinates in 12 bits each and the path depth in 4 (vrt_pool.h, pack_ids): frames up to
    4096 x 4096 and 15 bounces.  At the limits (column 4095; depth 15) it must still equal the oracle; one step beyond
    (4104 columns; rows 4090-4099 of a 4100-row frame; 16 bounces) the library falls back to the one-path-per-lane kernel
    by itself and must equal it too."""
    mat, rgb, params = scenes.scene_sunlit(0)
    kw = dict(voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=3)
    if rows:
        kw["rows"] = rows
    g, o = NativeSession(_lib.load(), "vrt_", host.make_config(W, H, **kw)), orc.Oracle(host.make_config(W, H, **kw), threads=16)
    for s in (g, o):
        orc.setup(s, mat, rgb, params)
        s.accumulate(4)
        s.accumulate(2)
    r0, r1 = rows or (0, H)
    a, b = g.fetch_hdr()[r0:r1], o.fetch_hdr()[r0:r1]
    assert np.isfinite(a).all() and a.mean() > 0.0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} of {a.size} values differ"
    for which in (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_MAT):
        assert np.array_equal(g.fetch_buffer(which)[r0:r1].view(np.uint8), o.fetch_buffer(which)[r0:r1].view(np.uint8)), which
    g.close(); o.close()

"""Deterministic synthetic voxel scenes for tests and bench.py.

Every builder returns ``(mat int8[G,G,G], rgb uint8[G,G,G,3], params dict)`` with arrays indexed
``[x+G/2, y+G/2, z+G/2]`` -- the storage of voxel_world.py:14-18; G = 128 (the reference's grid) unless the
builder says otherwise (BASELINE config 5 uses 256, with voxels of half the size: the world box stays [-1,1]^3).  Colours follow
Renderer.set_voxel (pathtracer.py:1325-1328, math_utils.py:86-92): u8 = trunc(clamp(c,0,1)*255)
evaluated in float32.  Randomness is a numpy PCG hash of the cell index, so a scene depends only
on its seed.
"""
import numpy as np

G = 128
OFF = 64


def pcg_hash(v):
    v = np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF
    s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w).astype(np.uint32)


def rand01(seed, idx):
    """float32 uniform [0,1) per index (24-bit)."""
    h = pcg_hash(pcg_hash(np.uint64(seed) ^ np.uint64(0x9E3779B9)).astype(np.uint64) + np.asarray(idx, dtype=np.uint64))
    return (h >> 8).astype(np.float32) * np.float32(2.0 ** -24)


def color_u8(c):
    c = np.clip(np.asarray(c, dtype=np.float32), np.float32(0.0), np.float32(1.0))
    return (c * np.float32(255.0)).astype(np.uint8)


def empty(grid=G):
    return np.zeros((grid, grid, grid), dtype=np.int8), np.zeros((grid, grid, grid, 3), dtype=np.uint8)


def _set(mat, rgb, x, y, z, m, c):
    mat[x + OFF, y + OFF, z + OFF] = m
    rgb[x + OFF, y + OFF, z + OFF] = color_u8(c)


def scene_s1(seed=0):
    """Example-1-style scene (reference example1.py:9-24): a 50x50 slab at y=0 with an emissive
    border, 4 % of the interior cells grow a blue tower capped by a white emissive voxel.  Lit
    only by the emissive voxels (sun colour and background stay at their (0,0,0) defaults)."""
    mat, rgb = empty()
    n = 50
    for i in range(n):
        for j in range(n):
            if min(i, j) == 0 or max(i, j) == n - 1:
                _set(mat, rgb, i, 0, j, 2, (0.9, 0.1, 0.1))
            else:
                _set(mat, rgb, i, 0, j, 1, (0.9, 0.1, 0.1))
                k = i * n + j
                if rand01(seed, 2 * k) < np.float32(0.04):
                    height = int(rand01(seed, 2 * k + 1) * np.float32(20.0))
                    for y in range(1, height):
                        _set(mat, rgb, i, y, j, 1, (0.0, 0.5, 0.9))
                    if height:
                        _set(mat, rgb, i, height, j, 2, (1.0, 1.0, 1.0))
    params = dict(exposure=10.0, voxel_edges=0.06, floor_height=-0.05, floor_color=(1.0, 1.0, 1.0), floor_material=1,
                  background_color=(0.0, 0.0, 0.0), light_direction=(1.0, 1.0, 1.0), light_cone=0.1,
                  light_color=(0.0, 0.0, 0.0), use_physical_sky=0, use_clouds=0)
    return mat, rgb, params


def scene_sunlit(seed=0):
    """A small sun-lit scene touching every BSDF lobe: glossy / metallic / clear-coated blocks on a
    floor under a coloured sky (example4.py-style lighting).  Used by the parity tests so that
    NEE, MIS and all three samplers contribute."""
    mat, rgb = empty()
    mats = [1, 11, 21, 32, 51, 53, 54, 82, 20, 2]
    k = 0
    for bx in range(-3, 3):
        for bz in range(-2, 2):
            m = mats[k % len(mats)]
            h = 3 + (k * 7) % 9
            col = (0.3 + 0.1 * (k % 7), 0.9 - 0.1 * (k % 5), 0.4 + 0.15 * (k % 4))
            for x in range(6):
                for z in range(6):
                    for y in range(h):
                        _set(mat, rgb, bx * 9 + x, y - 10, bz * 9 + z, m, col)
            k += 1
    params = dict(exposure=1.0, voxel_edges=0.06, floor_height=-10.0 / 64.0, floor_color=(0.8, 0.8, 0.8), floor_material=10,
                  background_color=(0.3, 0.4, 0.6), light_direction=(1.0, 1.0, 1.0), light_cone=0.1,
                  light_color=(1.0, 1.0, 1.0), use_physical_sky=0, use_clouds=0)
    return mat, rgb, params


def scene_dense(seed=12345, occupancy=0.5, grid=128):
    """Config 4 (grid 128) / config 5 (grid 256): every voxel solid with probability `occupancy`, material 1,
    random colour."""
    G = int(grid)
    idx = np.arange(G * G * G, dtype=np.uint64)
    solid = rand01(seed, idx * 4) < np.float32(occupancy)
    mat = solid.astype(np.int8).reshape(G, G, G)
    rgb = np.stack([(pcg_hash(np.uint64(seed) + idx * 4 + c) & 0xFF).astype(np.uint8) for c in (1, 2, 3)], axis=-1)
    rgb = (rgb * solid[:, None]).reshape(G, G, G, 3).astype(np.uint8)
    params = dict(exposure=1.0, voxel_edges=0.06, floor_height=-1.0, floor_color=(1.0, 1.0, 1.0), floor_material=1,
                  background_color=(0.3, 0.4, 0.6), light_direction=(1.0, 1.0, 1.0), light_cone=0.1,
                  light_color=(1.0, 1.0, 1.0), use_physical_sky=0, use_clouds=0)
    return np.ascontiguousarray(mat), np.ascontiguousarray(rgb), params


def scene_s6(seed=0):
    """Example-6-style scene (reference example6.py): stepped ground, trunks with leaf crowns, a
    fence, physical sky + clouds, sun (1,1,-1) cone 0.025 colour 1.3*(1.0,0.949,0.937),
    floor -0.85, voxel_edges 0, exposure 2."""
    mat, rgb = empty()

    def block(pos, size, color, noise, m=11, tag=0):
        x0, y0, z0 = pos
        xs, ys, zs = np.meshgrid(np.arange(x0, x0 + size[0]), np.arange(y0, y0 + size[1]), np.arange(z0, z0 + size[2]),
                                 indexing="ij")
        ok = (xs >= -64) & (xs < 64) & (ys >= -64) & (ys < 64) & (zs >= -64) & (zs < 64)
        xs, ys, zs = xs[ok], ys[ok], zs[ok]
        lin = ((xs + OFF) * G + (ys + OFF)) * G + (zs + OFF)
        r = rand01(seed + 17 * tag, lin)
        c = np.asarray(color, dtype=np.float32)[None, :] + np.asarray(noise, dtype=np.float32)[None, :] * r[:, None]
        mat[xs + OFF, ys + OFF, zs + OFF] = m
        rgb[xs + OFF, ys + OFF, zs + OFF] = color_u8(c)

    for i in range(4):
        base = np.float32(0.5 - i * 0.1) * np.array([1.0, 0.8, 0.6], dtype=np.float32)
        block((-60, -(i + 1) ** 2 - 40, -60), (120, 2 * i + 1, 120), base, (0.05 * (3 - i),) * 3, tag=i)
    block((-60, -40, -60), (120, 1, 120), (0.3, 0.2, 0.1), (0.01,) * 3, tag=5)

    def tree(pos, height, radius, color, tag):
        block(pos, (3, int(height - radius * 0.5), 3), (0.7, 0.7, 0.7), (0.3, 0.3, 0.3), tag=tag)
        cx, cy, cz = pos[0], pos[1] + height, pos[2]
        r = radius
        I = np.mgrid[-r:r, -r:r, -r:r].reshape(3, -1).T
        f = I.astype(np.float32) / np.float32(r)
        h = 0.5 - np.maximum(f[:, 1], -0.5) * 0.5
        d = np.sqrt(f[:, 0] ** 2 + f[:, 2] ** 2)
        prob = np.maximum(0, 1 - d) ** 2 * h * h
        prob += np.sin(f[:, 0] * 5 + cx) * 0.02 + np.sin(f[:, 1] * 9 + cy) * 0.01 + np.sin(f[:, 2] * 10 + cz) * 0.03
        prob[prob < 0.1] = 0.0
        P = I + np.array([cx, cy, cz])
        ok = np.all((P >= -64) & (P < 64), axis=1)
        lin = ((P[:, 0] + OFF) * G + (P[:, 1] + OFF)) * G + (P[:, 2] + OFF)
        pick = ok & (rand01(seed + 101 * tag, lin.astype(np.uint64) * 2) < prob)
        P = P[pick]
        jit = (rand01(seed + 101 * tag, lin[pick].astype(np.uint64) * 2 + 1) - np.float32(0.5)) * np.float32(0.2)
        mat[P[:, 0] + OFF, P[:, 1] + OFF, P[:, 2] + OFF] = 80
        rgb[P[:, 0] + OFF, P[:, 1] + OFF, P[:, 2] + OFF] = color_u8(np.asarray(color, dtype=np.float32)[None, :] + jit[:, None])

    tree((-20, -40, 25), 65, 35, (1.0, 0.3, 0.15), 1)
    tree((45, -40, -45), 15, 10, (0.8, 0.4, 0.1), 2)
    tree((20, -40, 0), 45, 25, (1.0, 0.4, 0.1), 3)
    tree((30, -40, -20), 25, 15, (1.0, 0.4, 0.1), 4)
    tree((30, -40, 30), 45, 25, (1.0, 0.4, 0.1), 5)

    def fence(start, direction, length, tag):
        color = (0.5, 0.3, 0.2)
        d = np.array(direction)
        block(start, tuple(d * length + np.array([3, 2, 3])), color, (0.1,) * 3, tag=tag)
        for i in range(length // 3 + 1):
            p = np.array(start) + d * i * 3 + np.array([1, -3, 1])
            block(tuple(p), (1, 5, 1), color, (0.0,) * 3, tag=tag + 1)

    fence((-58, -36, -58), (1, 0, 0), 115, 20)
    fence((-59, -36, 57), (1, 0, 0), 115, 22)
    fence((-59, -36, -58), (0, 0, 1), 115, 24)
    fence((57, -36, -58), (0, 0, 1), 115, 26)

    params = dict(exposure=2.0, voxel_edges=0.0, floor_height=-0.85, floor_color=(1.0, 1.0, 1.0), floor_material=1,
                  background_color=(0.0, 0.0, 0.0), light_direction=(1.0, 1.0, -1.0), light_cone=0.025,
                  light_color=(1.0 * 1.3, 0.949 * 1.3, 0.937 * 1.3), use_physical_sky=1, use_clouds=1)
    return mat, rgb, params


def upsample2(mat, rgb):
    """The same scene on a grid of twice the resolution: every voxel becomes 2x2x2 voxels of half the size."""
    m = np.repeat(np.repeat(np.repeat(mat, 2, axis=0), 2, axis=1), 2, axis=2)
    c = np.repeat(np.repeat(np.repeat(rgb, 2, axis=0), 2, axis=1), 2, axis=2)
    return np.ascontiguousarray(m), np.ascontiguousarray(c)


def scene_s1_256(seed=0):
    """Scene S1 refined to 256^3 (sparse: 0.17 % occupancy), same world geometry, voxel edges drawn per fine voxel."""
    mat, rgb, params = scene_s1(seed)
    mat, rgb = upsample2(mat, rgb)
    return mat, rgb, params


def scene_sponge256(seed=0):
    """A level-5 Menger sponge (243^3 cells) centred in a 256^3 grid, sun-lit, a few emissive cells: occupancy that is
    neither sparse nor uniformly dense -- holes at every scale of the brick pyramid, long walks through them."""
    G, n = 256, 243
    c = np.arange(n)
    digits = np.stack([(c // 3 ** k) % 3 == 1 for k in range(5)], axis=0)      # [level][coord]: digit is 1
    dx, dy, dz = digits[:, :, None, None], digits[:, None, :, None], digits[:, None, None, :]
    hole = ((dx & dy) | (dx & dz) | (dy & dz)).any(axis=0)                    # two or more middle digits at some level
    solid = ~hole
    mat = np.zeros((G, G, G), dtype=np.int8)
    rgb = np.zeros((G, G, G, 3), dtype=np.uint8)
    o = (G - n) // 2
    idx = np.arange(n * n * n, dtype=np.uint64).reshape(n, n, n)
    r = rand01(seed, idx * 2)
    m = np.where(r < np.float32(0.002), 2, np.where(r < np.float32(0.3), 11, 1)).astype(np.int8)
    mat[o:o + n, o:o + n, o:o + n] = np.where(solid, m, 0)
    col = np.stack([(pcg_hash(np.uint64(seed) + idx * 4 + k) & 0x7F).astype(np.uint8) + 96 for k in (1, 2, 3)], axis=-1)
    rgb[o:o + n, o:o + n, o:o + n] = col * solid[..., None]
    params = dict(exposure=1.0, voxel_edges=0.06, floor_height=-1.0, floor_color=(0.8, 0.8, 0.8), floor_material=1,
                  background_color=(0.3, 0.4, 0.6), light_direction=(1.0, 1.0, 1.0), light_cone=0.1,
                  light_color=(1.0, 1.0, 1.0), use_physical_sky=0, use_clouds=0)
    return mat, rgb, params


def scene_dense256(seed=12345, occupancy=0.5):
    """Config 5's stress scene: the config-4 fill on the 256^3 grid (64 MiB of texels, 2 MiB of fine brick words)."""
    return scene_dense(seed, occupancy, grid=256)


SCENES = {"s1": scene_s1, "sunlit": scene_sunlit, "dense": scene_dense, "s6": scene_s6,
          "s1_256": scene_s1_256, "sponge256": scene_sponge256, "dense256": scene_dense256}

"""Deterministic synthetic scenes for the stress configurations (no such assets ship with the
reference: SURVEY §0-D7).  `tessellate(scene, n)` splits every face into n*n coplanar sub-faces
with interpolated normals/uvs (indoor.obj x 24*24 = 256 896 one-sided triangles: the
"Sponza-class" deep-BVH workload of BASELINE.json configs[3]) keeping the storage order, so the
same lights, camera, materials and tie-break rule apply."""
from __future__ import annotations

import numpy as np

from .scene import FACE_DTYPE, HostScene


def tessellate(scene: HostScene, n: int) -> HostScene:
    if n < 1:
        raise ValueError("n must be >= 1")
    f = scene.faces
    nf = len(f)
    # barycentric lattice: sub-triangles of row r: (r, c) "up" and "down"
    tris = []
    for r in range(n):
        for c in range(n - r):
            a, b, d = (r, c), (r + 1, c), (r, c + 1)
            tris.append((a, b, d))
            if c < n - r - 1:
                tris.append((b, (r + 1, c + 1), d))
    tris = np.asarray(tris, dtype=np.float32) / np.float32(n)      # [n*n, 3 corners, (u, v)]
    u = tris[:, :, 0][None, :, :, None]                             # weight of vertex 1
    v = tris[:, :, 1][None, :, :, None]                             # weight of vertex 2
    w = np.float32(1.0) - u - v

    def interp(attr):                                               # attr: [nf, 3, k]
        a0, a1, a2 = attr[:, None, 0:1, :], attr[:, None, 1:2, :], attr[:, None, 2:3, :]
        return (w * a0 + u * a1 + v * a2).astype(np.float32).reshape(nf * len(tris), 3, attr.shape[2])

    out = np.zeros(nf * len(tris), dtype=FACE_DTYPE)
    out["vertices"] = interp(f["vertices"])
    out["normals"] = interp(f["normals"])
    out["texcoords"] = interp(f["texcoords"])
    out["material_id"] = np.repeat(f["material_id"], len(tris))
    with np.errstate(all="ignore"):                                 # scene.cpp:251-261
        e1 = out["vertices"][:, 1] - out["vertices"][:, 0]
        e2 = out["vertices"][:, 2] - out["vertices"][:, 0]
        d1 = out["texcoords"][:, 1] - out["texcoords"][:, 0]
        d2 = out["texcoords"][:, 2] - out["texcoords"][:, 0]
        ff = np.float32(1.0) / (d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1])
        out["tangent"] = (ff[:, None] * (d2[:, 1:2] * e1 - d1[:, 1:2] * e2)).astype(np.float32)
    return HostScene(out, scene.mesh_sizes * np.uint32(len(tris)), scene.materials, scene.lights, scene.textures,
                     scene.texels, scene.camera, scene.cubemap)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: a "Sponza-class" scene written as OBJ + MTL + .scene so that it goes through the same
# loader path as the reference's assets (scene.cpp:304-358 -> tinyobj -> flatten).

def _grid(nu, nv):
    """Indices of the 2*nu*nv triangles of an (nu+1) x (nv+1) vertex lattice, vertex (i, j) -> i * (nv + 1) + j.
    Winding: (i,j) -> (i+1,j) -> (i,j+1): counter-clockwise when u runs right and v runs up, seen from +normal."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b, c, d = a + (nv + 1), a + 1, a + (nv + 1) + 1
    return np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)], 0)


def _surface(fn, nu, nv):
    """Samples fn(u, v) -> (position[...,3], normal[...,3]) on the unit square; returns (P, N, UV, triangles).  Triangles
    are wound so that their front (counter-clockwise) side is the side the supplied normals point to: the reference's
    faces are one-sided (intersection.cuh:110)."""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    p, n = fn(u, v)
    p, n = p.reshape(-1, 3), n.reshape(-1, 3)
    tri = _grid(nu, nv)
    g = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    if (g * n[tri[:, 0]]).sum() < 0:
        tri = tri[:, ::-1]
    uv = np.stack([u, v], -1).reshape(-1, 2)
    return p, n, uv, tri


def atrium_parts(seed: int = 12345):
    """Seed-fixed geometry of the atrium: a 24 x 12 m hall with an uneven stone floor, a barrel vault, two rows of fluted
    columns carrying arches, balustrades with glass balusters, hanging drapes, and clerestory openings to the
    environment.  ~250 k one-sided triangles (front = counter-clockwise, as intersection.cuh:110 culls), real depth
    complexity from the camera at one end looking down the nave.  Returns [(name, material, P, N, UV, triangles)]."""
    rng = np.random.RandomState(seed)        # MT19937, like the survey's std::mt19937(12345)
    parts = []
    L, Wd, Hh = 12.0, 6.0, 8.0               # half length (x), half width (z), wall height (y)

    # floor: displaced lattice (terrain-like), normal +y
    k = rng.uniform(0.5, 3.0, size=(6, 2)); ph = rng.uniform(0, 2 * np.pi, size=6); amp = rng.uniform(0.004, 0.02, size=6)

    def floor(u, v):
        x, z = (2 * u - 1) * L, (1 - 2 * v) * Wd                       # v runs towards -z so that du x dv = +y
        y = sum(a * np.sin(kx * x + kz * z + p) for (kx, kz), p, a in zip(k, ph, amp))
        dydx = sum(a * kx * np.cos(kx * x + kz * z + p) for (kx, kz), p, a in zip(k, ph, amp))
        dydz = sum(a * kz * np.cos(kx * x + kz * z + p) for (kx, kz), p, a in zip(k, ph, amp))
        n = np.stack([-dydx, np.ones_like(x), -dydz], -1)
        return np.stack([x, y, z], -1), n / np.linalg.norm(n, axis=-1, keepdims=True)
    parts.append(("floor", "paving") + _surface(floor, 240, 120))

    # barrel vault over the nave, normal pointing down into the hall
    def vault(u, v):
        x = (2 * u - 1) * L
        th = np.pi * v                                                   # 0..pi across the width
        z, y = -Wd * np.cos(th), Hh + 0.55 * Wd * np.sin(th)
        n = np.stack([np.zeros_like(x), -0.55 * np.sin(th), np.cos(th)], -1)   # inward (towards the axis)
        return np.stack([x, y, z], -1), n / np.linalg.norm(n, axis=-1, keepdims=True)
    parts.append(("vault", "plaster") + _surface(vault, 240, 48))

    # walls: long walls with a clerestory band left open (y in [5.2, 6.6]) so that the environment lights the hall
    def wall(x0, z0, x1, z1, y0, y1, nx, ny):
        def f(u, v):
            x, z, y = x0 + (x1 - x0) * u, z0 + (z1 - z0) * u, y0 + (y1 - y0) * v
            d = np.array([x1 - x0, 0.0, z1 - z0]); up = np.array([0.0, 1.0, 0.0])
            n = np.cross(up, d); n = n / np.linalg.norm(n)                # into the hall
            return np.stack([x, y, z], -1), np.broadcast_to(n, x.shape + (3,)).copy()
        return _surface(f, nx, ny)
    for name, (x0, z0, x1, z1) in (("wall_n", (L, -Wd, -L, -Wd)), ("wall_s", (-L, Wd, L, Wd))):
        parts.append((name + "_low", "plaster") + wall(x0, z0, x1, z1, -0.1, 5.2, 96, 20))
        parts.append((name + "_high", "plaster") + wall(x0, z0, x1, z1, 6.6, Hh, 96, 6))
    parts.append(("wall_e", "plaster") + wall(L, Wd, L, -Wd, -0.1, Hh + 0.6 * Wd, 24, 24))
    parts.append(("wall_w", "plaster") + wall(-L, -Wd, -L, Wd, -0.1, Hh + 0.6 * Wd, 24, 24))

    # two rows of fluted columns (radius modulated by 12 flutes, entasis along the height), outward normals
    def column(cx, cz, r0, h):
        def f(u, v):
            th = 2 * np.pi * u
            r = r0 * (1.0 - 0.12 * v * v) * (1.0 + 0.06 * np.cos(12 * th))
            x, z, y = cx + r * np.cos(th), cz - r * np.sin(th), h * v     # -sin: u runs clockwise seen from above -> du x dv outward
            n = np.stack([np.cos(th), np.zeros_like(th), -np.sin(th)], -1)
            return np.stack([x, y, z], -1), n
        return _surface(f, 48, 32)
    col_x = np.linspace(-L + 1.5, L - 1.5, 16)
    for row, cz in enumerate((-3.2, 3.2)):
        for i, cx in enumerate(col_x):
            parts.append((f"column_{row}_{i}", "stone") + column(cx, cz, 0.28 + 0.02 * rng.rand(), 4.6))

    # arches between neighbouring columns of a row: half tori in the x-y plane
    def arch(cx, cz, span):
        R, r = span / 2, 0.16
        def f(u, v):
            a, b = np.pi * u, 2 * np.pi * v
            cxr = (R + r * np.cos(b))
            x, y, z = cx - cxr * np.cos(a), 4.6 + cxr * np.sin(a), cz + r * np.sin(b)
            n = np.stack([-np.cos(b) * np.cos(a), np.cos(b) * np.sin(a), np.sin(b)], -1)
            return np.stack([x, y, z], -1), n
        return _surface(f, 24, 8)
    for row, cz in enumerate((-3.2, 3.2)):
        for i in range(len(col_x) - 1):
            parts.append((f"arch_{row}_{i}", "stone") + arch(0.5 * (col_x[i] + col_x[i + 1]), cz, col_x[i + 1] - col_x[i]))

    # balustrades on top of the arcades: small turned balusters, every fifth one of glass (Ni 1.5: the refraction branch)
    def baluster(cx, cz, y0):
        def f(u, v):
            th = 2 * np.pi * u
            r = 0.035 + 0.025 * np.sin(np.pi * v) ** 2
            x, z, y = cx + r * np.cos(th), cz - r * np.sin(th), y0 + 0.7 * v
            n = np.stack([np.cos(th), -0.1 * np.cos(np.pi * v) * np.ones_like(th), -np.sin(th)], -1)
            return np.stack([x, y, z], -1), n / np.linalg.norm(n, axis=-1, keepdims=True)
        return _surface(f, 8, 6)
    bx = np.linspace(-L + 1.0, L - 1.0, 110)
    for row, cz in enumerate((-3.2, 3.2)):
        for i, cx in enumerate(bx):
            parts.append((f"baluster_{row}_{i}", "glass" if i % 5 == 2 else "bronze") + baluster(cx, cz, 5.45))

    # drapes hanging between the columns and the side walls: sinusoidal sheets facing the nave
    def drape(x0, cz, side, phase):
        def f(u, v):
            x = x0 + 2.4 * u
            y = 5.0 - 3.4 * v
            z = cz + side * (0.12 * np.sin(9.0 * u * np.pi + phase) * (0.3 + v) + 0.05 * np.sin(5.0 * v + phase))
            dzdx = side * 0.12 * 9.0 * np.pi / 2.4 * np.cos(9.0 * u * np.pi + phase) * (0.3 + v)
            n = np.stack([dzdx, np.zeros_like(x), -np.ones_like(x)], -1) * side
            return np.stack([x, y, z], -1), n / np.linalg.norm(n, axis=-1, keepdims=True)
        return _surface(f, 64, 40)
    for i, x0 in enumerate(np.linspace(-L + 2.0, L - 4.4, 4)):
        parts.append((f"drape_n_{i}", "cloth") + drape(x0, -4.4, -1.0, rng.uniform(0, 6.28)))
        parts.append((f"drape_s_{i}", "cloth") + drape(x0, 4.4, 1.0, rng.uniform(0, 6.28)))
    return parts


ATRIUM_MTL = """# atrium.mtl - written by cuda_pathtracer_amd.synthetic.write_atrium
newmtl paving
Kd 0.52 0.47 0.41
Ks 0.05 0.05 0.05
Ni 1.0

newmtl plaster
Kd 0.78 0.76 0.70
Ks 0.0 0.0 0.0
Ni 1.0

newmtl stone
Kd 0.66 0.62 0.55
Ks 0.1 0.1 0.1
Ni 1.0

newmtl bronze
Kd 0.55 0.36 0.16
Ks 0.6 0.6 0.6
Ni 1.0

newmtl glass
Kd 0.92 0.96 0.94
Ks 0.0 0.0 0.0
Ni 1.5

newmtl cloth
Kd 0.62 0.08 0.10
Ks 0.0 0.0 0.0
Ni 1.0
"""

ATRIUM_SCENE = """# no cubemap line: the reference then binds its 1x1 fallback environment 0x131b23 (gpu_processor.cpp:128-132)
# camera x y z dir_x dir_y dir_z fov_x dof_focus dof_aperture
camera -10.6 1.75 0.45 1.0 0.02 -0.045 80.0 9.0 0.01

# pos_x pos_y pos_z r g b emission radius
p_light -9.0 3.4 0.0 1.0 0.95 0.9 6.0 1.0
p_light -5.0 3.4 0.4 1.0 0.95 0.9 6.0 1.0
p_light -1.0 3.4 -0.4 1.0 0.95 0.9 6.0 1.0
p_light 3.0 3.4 0.4 1.0 0.95 0.9 6.0 1.0
p_light 7.0 3.4 -0.4 1.0 0.95 0.9 6.0 1.0
p_light 10.5 3.4 0.0 1.0 0.95 0.9 6.0 1.0

scene obj/atrium.obj
"""


def write_atrium(out_dir: str, seed: int = 12345) -> str:
    """Writes <out_dir>/atrium.scene, <out_dir>/obj/atrium.obj and .mtl (about 250 k triangles, 27 MB of text) and returns
    the .scene path.  Deterministic: the same seed gives the same bytes."""
    import io
    import os
    os.makedirs(os.path.join(out_dir, "obj"), exist_ok=True)
    buf = io.StringIO()
    buf.write("# atrium.obj - written by cuda_pathtracer_amd.synthetic.write_atrium (seed %d)\nmtllib atrium.mtl\n" % seed)
    base = 1
    for name, mtl, p, n, uv, tri in atrium_parts(seed):
        buf.write("o %s\n" % name)
        np.savetxt(buf, p, fmt="v %.6f %.6f %.6f")
        np.savetxt(buf, uv, fmt="vt %.6f %.6f")
        np.savetxt(buf, n, fmt="vn %.6f %.6f %.6f")
        buf.write("usemtl %s\ns off\n" % mtl)
        t = tri + base
        np.savetxt(buf, np.repeat(t, 3, axis=1), fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
        base += len(p)
    with open(os.path.join(out_dir, "obj", "atrium.obj"), "w") as f:
        f.write(buf.getvalue())
    with open(os.path.join(out_dir, "obj", "atrium.mtl"), "w") as f:
        f.write(ATRIUM_MTL)
    scene = os.path.join(out_dir, "atrium.scene")
    with open(scene, "w") as f:
        f.write(ATRIUM_SCENE)
    return scene

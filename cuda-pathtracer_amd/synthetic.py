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

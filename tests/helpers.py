"""Shared helpers for the test-suite: synthetic scenes in the C-ABI layouts."""
import numpy as np


def make_scene(P, tris, normals=None, uvs=None, material_ids=None, materials=None, lights=None,
               textures=None, mesh_sizes=None, camera=None):
    """tris: float32[n,3,3].  materials: list of (diffuse_tex, normal_tex, ior).  textures: list of
    float32 arrays [h,w,c].  Tangents are computed like scene.cpp:251-261."""
    tris = np.asarray(tris, dtype=np.float32)
    n = len(tris)
    faces = np.zeros(n, dtype=P.FACE_DTYPE)
    faces["vertices"] = tris
    if normals is None:
        with np.errstate(all="ignore"):   # (degenerate, NaN, infinite and huge triangles are test inputs)
            e1 = tris[:, 1] - tris[:, 0]
            e2 = tris[:, 2] - tris[:, 0]
            nn = np.cross(e1, e2)
            ln = np.linalg.norm(nn, axis=1, keepdims=True)
            nn = np.where(ln > 0, nn / np.maximum(ln, 1e-30), 0).astype(np.float32)
        normals = np.repeat(nn[:, None, :], 3, axis=1)
    faces["normals"] = np.asarray(normals, dtype=np.float32)
    if uvs is None:
        uvs = np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (n, 1, 1))
    faces["texcoords"] = np.asarray(uvs, dtype=np.float32)
    with np.errstate(all="ignore"):
        e1 = faces["vertices"][:, 1] - faces["vertices"][:, 0]
        e2 = faces["vertices"][:, 2] - faces["vertices"][:, 0]
        d1 = faces["texcoords"][:, 1] - faces["texcoords"][:, 0]
        d2 = faces["texcoords"][:, 2] - faces["texcoords"][:, 0]
        f = np.float32(1.0) / (d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1])
        faces["tangent"] = (f[:, None] * (d2[:, 1:2] * e1 - d1[:, 1:2] * e2)).astype(np.float32)
    if textures is None:
        textures = [np.array([[[0.7, 0.6, 0.5, 0.1]]], dtype=np.float32)]
    if materials is None:
        materials = [(0, -1, 1.0)]
    faces["material_id"] = 0 if material_ids is None else np.asarray(material_ids, dtype=np.uint32)
    mats = np.zeros(len(materials), dtype=P.MATERIAL_DTYPE)
    for i, (d, nm, ior) in enumerate(materials):
        mats[i] = (d, nm, ior, 0)
    tex = np.zeros(len(textures), dtype=P.TEXTURE_DTYPE)
    blob, off = [], 0
    for i, t in enumerate(textures):
        t = np.asarray(t, dtype=np.float32)
        tex[i] = (t.shape[1], t.shape[0], t.shape[2], 0, off)
        blob.append(t.reshape(-1))
        off += t.size
    lts = np.zeros(0 if lights is None else len(lights), dtype=P.LIGHT_DTYPE)
    for i, (pos, col, em, rad) in enumerate(lights or []):
        lts[i] = (col, pos, em, rad)
    cam = np.zeros((), dtype=P.CAMERA_DTYPE)
    if camera is None:
        camera = dict(position=(0.1, 0.2, 4.0), dir=(0.0, 0.0, -1.0), fov_x=1.2, aperture=0.02, focus_dist=3.0)
    cam["position"] = camera["position"]
    d = np.asarray(camera["dir"], dtype=np.float32)
    cam["dir"] = d / np.float32(np.sqrt((d * d).sum(dtype=np.float32)))
    cam["fov_x"], cam["aperture"], cam["focus_dist"], cam["speed"] = camera["fov_x"], camera["aperture"], camera["focus_dist"], 1.4
    return P.HostScene(faces, [n] if mesh_sizes is None else mesh_sizes, mats, lts, tex,
                       np.concatenate(blob) if blob else np.zeros(0, np.float32), cam, "")


def random_soup(rng, n, extent=2.0, size=0.6):
    c = rng.uniform(-extent, extent, size=(n, 1, 3))
    return (c + rng.normal(scale=size, size=(n, 3, 3))).astype(np.float32)


def random_rays(rng, n, extent=3.0):
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.uniform(0.2, 1.0, size=(n, 1))  # unnormalised directions occur on the path (Q4)
    o = rng.uniform(-extent, extent, size=(n, 3))
    return np.concatenate([d, o], axis=1).astype(np.float32)


def synthetic_cubemap(rng, size):
    return rng.uniform(0.0, 1.0, size=(6, size, size, 4)).astype(np.float32)

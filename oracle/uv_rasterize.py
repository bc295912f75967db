"""CPU restatement of the driver-side UV rasteriser (test infrastructure: imported only by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke()).

Follows Faceverse_manager.make_driven_rendering from the rasteriser call on (data_preprocess/FaceVerse/renderer.py:66-82) and
render_after_rasterize (training_avatar_texture/volumetric_rendering/renderer.py:556-571).  The rasteriser itself is pytorch3d
(third-party, CUDA-only, absent from /root/reference and from this image; environment.yml pins pytorch3d 0.7.x): its published
naive algorithm (pytorch3d/renderer/mesh/rasterize_meshes.py: rasterize_meshes_python, faces_per_pixel = 1, blur_radius > 0 =>
clip_barycentric_coords) is restated with a plain loop over faces.  PARITY UNPINNED by reference tests: no pytorch3d fixture can
be generated here; the HIP kernel is held to this restatement."""
import numpy as np

K_EPS = 1e-8


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _seg_dist2(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    l2 = dx * dx + dy * dy
    if l2 <= K_EPS:
        return (px - bx) ** 2 + (py - by) ** 2
    t = np.clip(((px - ax) * dx + (py - ay) * dy) / l2, 0.0, 1.0)
    return (ax + t * dx - px) ** 2 + (ay + t * dy - py) ** 2


def rasterize(verts, tris, size, blur_radius=1e-6):
    """verts [V,3] as handed to Meshes(); returns (pix_to_face [S,S] int, bary [S,S,3]) for the camera of get_renderer(orthoCam=True,
    K=[-1,-1,0,0], T=[0,0,10]) (ortho_renderer.py:56-66): ndc = (-x, -y), depth = z + 10; pixel i sits at ndc 1 - (2i+1)/S."""
    v = np.asarray(verts, np.float32).copy()
    v[:, 0] *= -1; v[:, 1] *= -1; v[:, 2] += 10.0
    ndc = (1.0 - (2.0 * np.arange(size, dtype=np.float32) + 1.0) / size).astype(np.float32)
    px, py = np.meshgrid(ndc, ndc, indexing='xy')                       # px[r, c] = ndc[c], py[r, c] = ndc[r]
    zbuf = np.full((size, size), np.inf, np.float32)
    face = np.full((size, size), -1, np.int64)
    bary = np.zeros((size, size, 3), np.float32)
    r = np.float32(np.sqrt(blur_radius))
    for f, (i0, i1, i2) in enumerate(np.asarray(tris)):
        a, b, c = v[i0], v[i1], v[i2]
        area = np.float32(_edge(c[0], c[1], a[0], a[1], b[0], b[1]))
        if abs(area) <= K_EPS or max(a[2], b[2], c[2]) < 0:
            continue
        xs, ys = (a[0], b[0], c[0]), (a[1], b[1], c[1])
        cols = np.nonzero((ndc >= min(xs) - r - 2.0 / size) & (ndc <= max(xs) + r + 2.0 / size))[0]
        rows = np.nonzero((ndc >= min(ys) - r - 2.0 / size) & (ndc <= max(ys) + r + 2.0 / size))[0]
        if not len(cols) or not len(rows):
            continue
        sl = (slice(rows[0], rows[-1] + 1), slice(cols[0], cols[-1] + 1))
        x, y = px[sl], py[sl]
        w0 = (_edge(x, y, b[0], b[1], c[0], c[1]) / (area + np.float32(K_EPS))).astype(np.float32)
        w1 = (_edge(x, y, c[0], c[1], a[0], a[1]) / (area + np.float32(K_EPS))).astype(np.float32)
        w2 = (_edge(x, y, a[0], a[1], b[0], b[1]) / (area + np.float32(K_EPS))).astype(np.float32)
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        d = np.minimum(np.vectorize(_seg_dist2)(x, y, a[0], a[1], b[0], b[1]),
                       np.minimum(np.vectorize(_seg_dist2)(x, y, b[0], b[1], c[0], c[1]), np.vectorize(_seg_dist2)(x, y, c[0], c[1], a[0], a[1])))
        keep = inside | (d < blur_radius)
        w = np.stack([np.clip(w0, 0, 1), np.clip(w1, 0, 1), np.clip(w2, 0, 1)], -1)
        w = w / np.maximum(w.sum(-1, keepdims=True), 1e-5)
        z = (w * np.array([a[2], b[2], c[2]], np.float32)).sum(-1).astype(np.float32)
        keep &= z >= 0
        better = keep & (z < zbuf[sl])                                   # ties keep the earlier (lower-index) face
        zbuf[sl] = np.where(better, z, zbuf[sl])
        face[sl] = np.where(better, f, face[sl])
        bary[sl] = np.where(better[..., None], w, bary[sl])
    return face, bary


def render_after_rasterize(face_attrs, pix_to_face, bary):
    """face_attrs [F,3,D]; returns [D+1, S, S]: interpolated attributes (0 where no face) + visibility (renderer.py:556-571)."""
    vis = (pix_to_face > -1)
    vals = (bary[..., None] * np.asarray(face_attrs, np.float32)[np.where(vis, pix_to_face, 0)]).sum(-2)
    vals[~vis] = 0
    return np.concatenate([vals.transpose(2, 0, 1), vis[None].astype(np.float32)], 0)


def make_driven_rendering(verts, tris, face_attrs, size=512, crop=(128, 114, 256, 256), blur_radius=1e-6, res=None):
    """uvcoords_image [res, res, 3] for one frame (FaceVerse/renderer.py:66-84): crop, then -- when `res` differs from the crop
    size -- bilinear interpolation of all channels INCLUDING the continuous mask (:78-79), and the threshold last (:82)."""
    face, bary = rasterize(verts, tris, size, blur_radius)
    rend = render_after_rasterize(face_attrs, face, bary)
    rend = rend * (rend[-1:] * rend[-2:-1])
    left, top, cw, ch = crop
    rend = rend[:, top:top + ch, left:left + cw]
    if res is not None and res != ch:
        import torch
        rend = torch.nn.functional.interpolate(torch.from_numpy(np.ascontiguousarray(rend))[None], size=(res, res), mode='bilinear',
                                               align_corners=False)[0].numpy()
    uv = rend.transpose(1, 2, 0)[..., :3].copy()
    uv[..., 2] = (uv[..., 2] >= 0.5).astype(np.float32)
    return uv

"""CPU restatement of the reference's vertex stage (TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product path).

Follows /root/reference/src/shaders/WaterSurfaceMesh.vert:24-41 for the grid of
WaterSurfaceMesh::CreateGridVertices (/root/reference/src/scene/WaterSurfaceMesh.cpp:500-533), with the
sampler the reference creates (vulkan/Sampler.cpp:60-66: LINEAR filter, REPEAT addressing).  Vulkan leaves
the precision of the filter weights to the implementation; this restatement and the HIP kernel both use
fp32 weights from frac(u*W - 0.5) and the evaluation order written below.  PARITY UNPINNED: the reference
has no test or golden vector for this stage, and the shader cannot run here.
"""
import numpy as np

F = np.float32


def sample_linear_repeat(tex: np.ndarray, us: np.ndarray, vs: np.ndarray) -> np.ndarray:
    """tex [N, N, 4] float32 (row = v, column = u); us, vs float32 arrays -> [..., 4] float32."""
    n = tex.shape[0]
    s = us * F(n) - F(0.5)
    t = vs * F(n) - F(0.5)
    fs, ft = np.floor(s), np.floor(t)
    a, b = (s - fs)[..., None], (t - ft)[..., None]
    x0 = fs.astype(np.int64) & (n - 1)
    y0 = ft.astype(np.int64) & (n - 1)
    x1, y1 = (x0 + 1) & (n - 1), (y0 + 1) & (n - 1)
    t00, t10, t01, t11 = tex[y0, x0], tex[y0, x1], tex[y1, x0], tex[y1, x1]
    ia, ib = F(1.0) - a, F(1.0) - b
    return ((t00 * ia + t10 * a) * ib + (t01 * ia + t11 * a) * b).astype(np.float32)


def displace_grid(disp: np.ndarray, nrm: np.ndarray, amp: float, grid: int, vertex_distance: float,
                  uv_scale: float = 1.0, choppy: float = -1.0):
    """Returns (positions, normals), each [(grid+1)^2, 4] float32, vertex i = y*(grid+1) + x."""
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    nrm = np.ascontiguousarray(nrm, dtype=np.float32)
    side, half = grid + 1, grid // 2
    i = np.arange(side * side)
    xi, yi = i % side - half, i // side - half                       # WaterSurfaceMesh.cpp:514-518
    px = xi.astype(np.float32) * F(vertex_distance)                  # .cpp:520-525
    pz = yi.astype(np.float32) * F(vertex_distance)
    u = (xi + half).astype(np.float32) / F(grid)                     # .cpp:528-531
    v = (yi + half).astype(np.float32) / F(grid)
    us, vs = u * F(uv_scale), v * F(uv_scale)                        # .vert:26
    d = sample_linear_repeat(disp, us, vs)
    dy = d[:, 1] * F(amp)                                            # .vert:27
    pos = np.stack([px + d[:, 0], F(0.0) + dy, pz + d[:, 2], d[:, 3]], axis=1).astype(np.float32)   # .vert:28-29
    sl = sample_linear_repeat(nrm, us, vs)                           # .vert:33
    nx = -(sl[:, 0] / (F(1.0) + F(choppy) * sl[:, 2]))               # .vert:34-38
    nz = -(sl[:, 1] / (F(1.0) + F(choppy) * sl[:, 3]))
    ln = np.sqrt(nx * nx + F(1.0) + nz * nz)
    out_n = np.stack([nx / ln, F(1.0) / ln, nz / ln, np.zeros_like(nx)], axis=1).astype(np.float32)
    return pos, out_n


def foam_mask(positions: np.ndarray) -> np.ndarray:
    """What the reference's fragment stage does with the interpolated w (WaterSurfaceMesh.frag:210-212:
    `if (inPos.w < 0.0) color = vec3(1.0)`): per VERTEX here, True where the surface folds over itself -- the
    Jacobian of the horizontal displacement, carried in displacement.w (WaterSurfaceMesh.vert:29), is negative."""
    return positions[:, 3] < F(0.0)


def displace_grid_cascades(disps, nrms, amps, uv_scales, grid: int, vertex_distance: float, choppy: float = -1.0):
    """Cascades (the reference's to-do "Endless - solving the tiling artifacts", README.md:37-44): the sum of
    several independent tiles, tile c sampled at uv * uv_scales[c]; heights times their own amplitude, slopes and displacement
    derivatives summed before the reference's normal formula (.vert:34-38); w = the smallest Jacobian slot."""
    side, half = grid + 1, grid // 2
    i = np.arange(side * side)
    xi, yi = i % side - half, i // side - half
    px = xi.astype(np.float32) * F(vertex_distance)
    pz = yi.astype(np.float32) * F(vertex_distance)
    u = (xi + half).astype(np.float32) / F(grid)
    v = (yi + half).astype(np.float32) / F(grid)
    dx = np.zeros(side * side, np.float32); dy = dx.copy(); dz = dx.copy()
    s = np.zeros((side * side, 4), np.float32)
    w = np.full(side * side, np.finfo(np.float32).max, np.float32)
    for d, q, amp, sc in zip(disps, nrms, amps, uv_scales):
        us, vs = u * F(sc), v * F(sc)
        sd = sample_linear_repeat(np.ascontiguousarray(d, dtype=np.float32), us, vs)
        dx = dx + sd[:, 0]; dy = dy + sd[:, 1] * F(amp); dz = dz + sd[:, 2]
        w = np.minimum(w, sd[:, 3])
        s = s + sample_linear_repeat(np.ascontiguousarray(q, dtype=np.float32), us, vs)
    pos = np.stack([px + dx, F(0.0) + dy, pz + dz, w], axis=1).astype(np.float32)
    nx = -(s[:, 0] / (F(1.0) + F(choppy) * s[:, 2]))
    nz = -(s[:, 1] / (F(1.0) + F(choppy) * s[:, 3]))
    ln = np.sqrt(nx * nx + F(1.0) + nz * nz)
    return pos, np.stack([nx / ln, F(1.0) / ln, nz / ln, np.zeros_like(nx)], axis=1).astype(np.float32)


def mip_chain(tex: np.ndarray):
    """The reference's mip chain of a map (s_kUseMipMapping, WaterSurfaceMesh.h:216; Texture2D::GenerateMipmaps,
    vulkan/Texture2D.cpp:228-330: level i = vkCmdBlitImage(VK_FILTER_LINEAR) of level i-1 into half the extent, floor(log2 N) + 1
    levels).  An exact 2:1 linear blit samples the corner shared by four source texels: the bilinear formula above with both
    weights 1/2, in the same evaluation order.  Returns levels 1 .. log2 N, [(N >> l, N >> l, 4)].  PARITY UNPINNED (Vulkan
    leaves the blit filter's precision to the implementation; the shipped reference has the switch off)."""
    lv, cur = [], np.ascontiguousarray(tex, dtype=np.float32)
    h = F(0.5)
    while cur.shape[0] > 1:
        t00, t10, t01, t11 = cur[0::2, 0::2], cur[0::2, 1::2], cur[1::2, 0::2], cur[1::2, 1::2]
        cur = ((t00 * h + t10 * h) * h + (t01 * h + t11 * h) * h).astype(np.float32)
        lv.append(cur)
    return lv

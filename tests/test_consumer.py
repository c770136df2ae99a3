"""Vertex-stage consumer: oracle properties on the CPU, HIP kernel against the
oracle on the GPU (through the C ABI)."""
import numpy as np
import pytest


def _maps(n, seed=0):
    rng = np.random.default_rng(seed)
    disp = rng.standard_normal((n, n, 4)).astype(np.float32)
    disp[..., 3] = 1.0
    nrm = (0.2 * rng.standard_normal((n, n, 4))).astype(np.float32)
    return disp, nrm


def test_oracle_sampler_is_bilinear_with_repeat():
    """WaterSurfaceMesh.vert:26,33 with vulkan/Sampler.cpp:60-66: texel centres reproduce the texel, vertices on
    texel corners (grid == map size, scale 1) average the four texels around the corner, and uv wraps."""
    from oracle import consumer as C
    n = 16
    disp, _ = _maps(n)
    k = np.arange(n)
    centres = ((k + 0.5) / n).astype(np.float32)
    got = C.sample_linear_repeat(disp, centres[None, :].repeat(n, 0), centres[:, None].repeat(n, 1))
    assert np.allclose(got, disp, rtol=0, atol=1e-6)
    corners = (k / n).astype(np.float32)
    got = C.sample_linear_repeat(disp, corners[None, :].repeat(n, 0), corners[:, None].repeat(n, 1))
    want = 0.25 * (disp + np.roll(disp, 1, 0) + np.roll(disp, 1, 1) + np.roll(np.roll(disp, 1, 0), 1, 1))
    assert np.allclose(got, want, rtol=0, atol=1e-6)
    shifted = C.sample_linear_repeat(disp, centres[None, :].repeat(n, 0) + np.float32(3.0), centres[:, None].repeat(n, 1) - np.float32(2.0))
    assert np.allclose(shifted, disp, rtol=0, atol=2e-5)


def test_oracle_grid_layout_and_normals():
    """WaterSurfaceMesh.cpp:500-533: (grid+1)^2 vertices, row-major in z then x, centred; flat water -> the grid itself."""
    from oracle import consumer as C
    n, g, vd = 8, 8, 2.5
    disp = np.zeros((n, n, 4), np.float32); disp[..., 3] = 1.0
    nrm = np.zeros((n, n, 4), np.float32)
    pos, nr = C.displace_grid(disp, nrm, 7.0, g, vd)
    assert pos.shape == ((g + 1) ** 2, 4)
    assert pos[0, 0] == -g / 2 * vd and pos[0, 2] == -g / 2 * vd and pos[-1, 0] == g / 2 * vd and pos[g, 0] == g / 2 * vd
    assert np.all(pos[:, 1] == 0) and np.all(pos[:, 3] == 1)
    assert np.all(nr[:, :3] == np.array([0, 1, 0], np.float32)) and np.all(nr[:, 3] == 0)
    # a uniform slope tilts every normal the same way; choppy rescales it (vert:34-38)
    nrm[..., 0] = 0.5; nrm[..., 2] = 0.25
    _, nr = C.displace_grid(disp, nrm, 7.0, g, vd, choppy=-1.0)
    nx = -(0.5 / (1 - 0.25)); ln = np.sqrt(nx * nx + 1)
    assert np.allclose(nr[:, 0], nx / ln, atol=1e-6) and np.allclose(nr[:, 1], 1 / ln, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,grid,scale", [(64, 64, 1.0), (256, 256, 1.0), (256, 100, 1.0), (128, 300, 2.5), (512, 512, 1.0)])
def test_displace_grid_matches_oracle(n, grid, scale):
    """The HIP kernel against oracle/consumer.py on a real frame: positions within 1e-6 of the position range,
    normals within 2e-6 (correctly rounded fp32 division and square root on both sides)."""
    import watersurfacerendering_amd as W
    from oracle import consumer as C
    b = W.OceanBatch(n, 1, 0)
    b.prepare(0x5EED0000)
    amp = float(b.compute_waves(2.75)[0])
    disp, nrm = b.read_maps()
    vd = 1000.0 / 512.0
    pos, nr = b.displace_grid(0, grid, vd, scale, -1.0)
    opos, onr = C.displace_grid(disp[0], nrm[0], amp, grid, vd, scale, -1.0)
    assert pos.shape == opos.shape == ((grid + 1) ** 2, 4)
    assert np.abs(pos - opos).max() <= 1e-6 * np.abs(opos).max()
    assert np.abs(nr - onr).max() <= 2e-6
    assert np.all(pos[:, 3] == 1.0) and np.all(nr[:, 3] == 0.0)
    # the amplitude folded into the map is undone by the consumer: y is the raw height again
    assert np.abs(pos[:, 1]).max() <= amp * (1 + 1e-6)
    b.close()


@pytest.mark.gpu
def test_displace_grid_argument_checks_and_batches():
    import watersurfacerendering_amd as W
    b = W.OceanBatch(64, 2, 0)
    with pytest.raises(W.OceanError):
        b.displace_grid(0, 64)                      # nothing prepared
    b.prepare(3)
    with pytest.raises(W.OceanError):
        b.displace_grid(0, 64)                      # no frame yet
    b.compute_waves(1.0)
    with pytest.raises(W.OceanError):
        b.displace_grid(2, 64)                      # tile out of range
    with pytest.raises(W.OceanError):
        b.displace_grid(0, 0)
    p0, _ = b.displace_grid(0, 64)
    p1, _ = b.displace_grid(1, 64)
    assert not np.array_equal(p0, p1)               # tiles have their own seed
    b.close()


def test_oracle_cascades_reduce_to_one_tile_and_break_the_tile_period():
    """oracle/consumer.py: one cascade = the plain vertex stage; a single tile sampled over two periods repeats exactly, the
    sum of tiles at incommensurate rates does not (the reference's to-do "solving the tiling artifacts", README.md:37-44)."""
    from oracle import consumer as C
    n = 32
    d0, q0 = _maps(n, 1); d1, q1 = _maps(n, 2)
    p1, n1 = C.displace_grid(d0, q0, 3.0, 64, 1.0, 2.0, -1.0)
    pc, nc = C.displace_grid_cascades([d0], [q0], [3.0], [2.0], 64, 1.0, -1.0)
    assert np.array_equal(p1, pc) and np.array_equal(n1, nc)
    side = 65
    y = p1[:, 1].reshape(side, side)
    assert np.allclose(y[:, :32], y[:, 32:64], atol=1e-5)                          # uv scale 2: the tile repeats after half the grid
    pc2, _ = C.displace_grid_cascades([d0, d1], [q0, q1], [3.0, 2.0], [2.0, 2.0 * 1.37], 64, 1.0, -1.0)
    y2 = pc2[:, 1].reshape(side, side)
    assert np.abs(y2[:, :32] - y2[:, 32:64]).max() > 0.5


@pytest.mark.gpu
def test_cascade_consumer_matches_oracle_on_tiles_of_different_length():
    """ocean_displace_grid_cascades on a batch whose tiles are cascades of one ocean (tile lengths 1000 / 370 / 93 m, own seeds):
    against oracle/consumer.py, equal to ocean_displace_grid for a single cascade, and no longer periodic with the base tile."""
    import watersurfacerendering_amd as W
    from oracle import consumer as C
    n, grid = 256, 512
    lengths = [1000.0, 370.0, 93.0]
    b = W.OceanBatch(n, len(lengths), 0)
    for i, L in enumerate(lengths):
        b.set_params(tile=i, tile_length=L)
    b.prepare(0x5EED0000)
    amps = b.compute_waves(2.75)
    disp, nrm = b.read_maps()
    scales = [2.0 * lengths[0] / L for L in lengths]              # the grid spans two base tiles; every cascade keeps its metres per texel
    vd = 2.0 * lengths[0] / grid
    pos, nr = b.displace_grid_cascades(scales, 0, grid, vd, -1.0)
    opos, onr = C.displace_grid_cascades(disp, nrm, [float(a) for a in amps], scales, grid, vd, -1.0)
    assert np.abs(pos - opos).max() <= 2e-6 * np.abs(opos).max()
    assert np.abs(nr - onr).max() <= 4e-6
    p1, n1 = b.displace_grid(0, grid, vd, scales[0], -1.0)
    pc, nc = b.displace_grid_cascades(scales[:1], 0, grid, vd, -1.0)
    assert np.array_equal(p1, pc) and np.array_equal(n1, nc)
    side = grid + 1
    y1 = p1[:, 1].reshape(side, side); y3 = pos[:, 1].reshape(side, side)
    assert np.abs(y1[:, :256] - y1[:, 256:512]).max() <= 1e-3      # one tile: the period of the base tile
    assert np.abs(y3[:, :256] - y3[:, 256:512]).max() > 1.0        # cascades: it is gone
    with pytest.raises(W.OceanError):
        b.displace_grid_cascades([1.0] * 4, 0, grid, vd)           # more cascades than tiles
    b.close()


def test_oracle_mip_chain_is_the_2x2_mean_down_to_one_texel():
    """Texture2D.cpp:228-330 (level i = linear blit of level i-1 into half the extent; floor(log2 N) + 1 levels): levels 1 .. log2 N,
    each the mean of 2 x 2 texels of the one above, the last one the mean of the whole map."""
    from oracle import consumer as C
    n = 32
    disp, _ = _maps(n, 3)
    lv = C.mip_chain(disp)
    assert [l.shape for l in lv] == [(n >> k, n >> k, 4) for k in range(1, 6)]
    want = 0.25 * (disp[0::2, 0::2] + disp[0::2, 1::2] + disp[1::2, 0::2] + disp[1::2, 1::2])
    assert np.allclose(lv[0], want, rtol=0, atol=1e-6)
    assert np.allclose(lv[-1][0, 0], disp.reshape(-1, 4).mean(0), rtol=0, atol=1e-5)
    assert np.array_equal(lv[0][..., 3], np.ones((n // 2, n // 2), np.float32))      # displacement.w = 1 survives every level


@pytest.mark.gpu
@pytest.mark.parametrize("n,tiles,tile", [(16, 1, 0), (256, 3, 2), (1024, 1, 0)])
def test_mip_chain_matches_oracle(n, tiles, tile):
    """ocean_build_mips / ocean_read_mips / ocean_device_mips against oracle/consumer.py::mip_chain of the maps the same frame
    read out: bit for bit (products by 1/2 are exact, the sums are taken in the same order)."""
    import ctypes
    import watersurfacerendering_amd as W
    from oracle import consumer as C
    b = W.OceanBatch(n, tiles, 0)
    b.prepare(0x5EED0000 + n)
    with pytest.raises(W.OceanError):
        b.build_mips(0)                                          # no frame yet
    b.compute_waves(1.25)
    disp, nrm = b.read_maps()
    dl, ql = b.build_mips(tile)
    od, oq = C.mip_chain(disp[tile]), C.mip_chain(nrm[tile])
    assert len(dl) == len(od) == n.bit_length() - 1
    for got, want in zip(dl + ql, od + oq):
        assert got.shape == want.shape and np.array_equal(got, want)
    assert int(b._L.ocean_mip_texels(n)) == (n * n - 1) // 3 == sum(l.shape[0] * l.shape[1] for l in od)
    pd, pq, lv = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
    assert b._L.ocean_device_mips(b._h, ctypes.byref(pd), ctypes.byref(pq), ctypes.byref(lv)) == 0
    assert pd.value and pq.value and lv.value == len(od)
    with pytest.raises(W.OceanError):
        b.build_mips(tiles)                                      # no such tile
    # a second frame, asynchronously: the chain is built behind it on the same stream
    b.compute_waves_async(2.5)
    dl2, _ = b.build_mips(tile)
    disp2, _ = b.read_maps()
    assert np.array_equal(dl2[0], C.mip_chain(disp2[tile])[0])
    # SetTileSize: the chain follows the new size (its buffers are re-allocated on the next build)
    if n >= 64:
        b.set_tile_size(n // 2)
        b.prepare(0x5EED0000 + n)
        b.compute_waves(0.75)
        disp3, nrm3 = b.read_maps()
        dl3, ql3 = b.build_mips(tile)
        for got, want in zip(dl3 + ql3, C.mip_chain(disp3[tile]) + C.mip_chain(nrm3[tile])):
            assert got.shape == want.shape and np.array_equal(got, want)
    b.close()

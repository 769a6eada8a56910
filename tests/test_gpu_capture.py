"""GPU: the dist2 producer (fd_capture_dist2*, next row N2) against the independently formulated
golden vectors and the oracle (reference src/capture.cpp:46-99), and chained into fd_deform.

Bar: the kernel works in fp32 on differences from the triangle's first vertex; |d2 - ref| <=
2e-6 * (ref + L^2) with L the rig's extent -- a few fp32 ulps of the lengths involved.  The
-1 / 0 decisions are exact except within that margin of the radius."""
import os
import numpy as np
import pytest
import torch

from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cap():
    return np.load(os.path.join(HERE, "golden", "capture_golden.npz"))


def _rig_triangles(rest):
    """A triangle fan over nearest neighbours: every control point with two of its neighbours."""
    tris = []
    for i in range(rest.shape[0]):
        d = np.linalg.norm(rest - rest[i], axis=1)
        j, k = np.argsort(d)[1:3]
        tris.append(np.concatenate([rest[i], rest[j], rest[k]]))
    return np.array(tris, np.float32)


def test_golden(hip_lib, cap):
    e = capi.Engine()
    P, tris, ref, mask = cap["P"], cap["tris"], cap["d2"], cap["mask"]
    out = e.capture_dist2(P, tris, 1e30, True)
    L2 = float(np.abs(tris).max()) ** 2
    assert np.all(np.abs(out - ref) <= 2e-6 * (ref + L2)), np.abs(out - ref).max()
    r2 = np.float32(0.09)
    out = e.capture_dist2(P, tris, r2, True, mask)
    inside = mask.astype(bool)
    margin = 2e-6 * (ref + L2)
    near = inside & (ref < r2 - margin)
    far = inside & (ref > r2 + margin)
    assert np.all(out[~inside] == 0.0) and np.all(out[far] == -1.0)
    assert np.all(np.abs(out[near] - ref[near]) <= margin[near])
    assert np.all(e.capture_dist2(P, tris, r2, False, mask) == 0.0)
    assert np.all(e.capture_dist2(P[:7], tris[:0], r2, True) == -1.0)
    e.close()


def test_full_size_against_oracle_sample_and_into_deform(hip_lib, oracle):
    """N = 1M, T = 256 triangles (more than one would stage at M = 256 is covered by T = 2100
    below): device-resident, the result feeds fd_deform_dev without leaving the device."""
    N, M = 1_000_000, 256
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 1)
    tris = _rig_triangles(rest)
    rng = np.random.default_rng(4)
    mask = (rng.random(N) < 0.8).astype(np.uint8)
    r2 = np.float32(0.05)
    d_P = torch.from_numpy(P).to(dev); d_tri = torch.from_numpy(tris).to(dev); d_mask = torch.from_numpy(mask).to(dev)
    d_d2 = torch.empty(N, device=dev, dtype=torch.float32)
    d_out = torch.empty_like(d_P); d_fall = torch.zeros(N, device=dev)
    e = capi.Engine()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    e.set_stream(stream.cuda_stream)
    e.set_points(rest, (deform - rest).astype(np.float32)); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
    e.build()
    e.capture_dist2_dev(N, d_P.data_ptr(), d_mask.data_ptr(), tris.shape[0], d_tri.data_ptr(), r2, True, d_d2.data_ptr())
    e.deform_dev(N, d_P.data_ptr(), d_out.data_ptr(), d_dist2=d_d2.data_ptr(), d_falloff=d_fall.data_ptr(),
                 radius2=r2, falloffrate=1.5)
    stream.synchronize()
    idx = np.linspace(0, N - 1, 5000).astype(np.int64)
    ref = oracle.capture_dist2(P[idx], tris, r2, True, mask[idx])
    got = d_d2.cpu().numpy()[idx]
    exact = oracle.capture_dist2(P[idx], tris, 1e30, True)            # unthresholded distances
    margin = 2e-6 * (exact + 1.0)
    decided = np.abs(exact - r2) > margin                              # away from the threshold
    assert np.all(got[decided & (ref == -1)] == -1.0) and np.all(got[mask[idx] == 0] == 0.0)
    sel = decided & (ref >= 0) & (mask[idx] == 1)
    assert sel.sum() > 100 and np.all(np.abs(got[sel] - ref[sel]) <= margin[sel])
    # and the deformation that consumed it, against the oracle fed with the device's dist2
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    ref_P, ref_fall = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P[idx], dist2=got, radius2=r2, falloffrate=1.5)
    assert parity_ratio(d_out.cpu().numpy()[idx], ref_P, P[idx], 1e-5) <= 1.0
    e.set_stream(None); e.close()


def test_many_triangles_and_ragged_n(hip_lib, oracle):
    rng = np.random.default_rng(8)
    N, T = 10_007, 2100                                     # three LDS chunks, ragged both ways
    P = (rng.normal(size=(N, 3)) * 0.7).astype(np.float32)
    v = rng.normal(size=(T, 3)).astype(np.float32)
    tris = np.concatenate([v, v + 0.1 * rng.normal(size=(T, 3)).astype(np.float32),
                           v + 0.1 * rng.normal(size=(T, 3)).astype(np.float32)], axis=1).astype(np.float32)
    e = capi.Engine()
    out = e.capture_dist2(P, tris, 1e30, True)
    ref = oracle.capture_dist2(P, tris, 1e30, True)
    assert np.all(np.abs(out - ref) <= 2e-6 * (ref + 16.0))
    e.close()


def _knn_adjacency(P, k=6):
    """Symmetric k-nearest-neighbour graph of a point cloud as a CSR adjacency (a stand-in for the
    mesh's edges: the synthetic meshes are lattices without topology)."""
    from scipy.spatial import cKDTree
    tree = cKDTree(P)
    _, idx = tree.query(P, k=k + 1)
    pairs = set()
    for i in range(P.shape[0]):
        for j in idx[i, 1:]:
            pairs.add((i, int(j))); pairs.add((int(j), i))
    nbrs = [[] for _ in range(P.shape[0])]
    for i, j in sorted(pairs):
        nbrs[i].append(j)
    offsets = np.zeros(P.shape[0] + 1, np.int64)
    offsets[1:] = np.cumsum([len(n) for n in nbrs])
    return offsets, np.concatenate([np.array(n, np.int32) for n in nbrs])


def test_islands_match_oracle_and_feed_the_capture(hip_lib, oracle):
    """fd_capture_islands against the oracle on a k-NN graph of the head mesh, for several ring
    counts, then the whole capture chain: islands -> dist2 -> deform."""
    N, M = 30_000, 40
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    offsets, nb = _knn_adjacency(P)
    e = capi.Engine()
    for k in (0, 1, 4, 9):
        got = e.capture_islands(P, offsets, nb, rest, k)
        ref = oracle.capture_islands(P, offsets, nb, rest, k)
        assert np.array_equal(got, ref), k
        assert ref.sum() >= M // 2 and (k == 0 or ref.sum() > M)
    assert e.capture_islands(P, offsets, nb, rest[:0], 4).sum() == 0
    mask = e.capture_islands(P, offsets, nb, rest, 6)
    tris = _rig_triangles(rest)
    r2 = np.float32(0.04)
    d2 = e.capture_dist2(P, tris, r2, True, mask)
    assert np.array_equal(d2, oracle.capture_dist2(P, tris, r2, True, mask)) or \
        np.all(np.abs(d2 - oracle.capture_dist2(P, tris, r2, True, mask))[(d2 >= 0)] <= 2e-6 * 2.0)
    assert np.all(d2[mask == 0] == 0.0)
    e.close()


def test_sop_cook_captures_on_the_device_and_caches_like_the_reference(hip_lib, oracle):
    """ProximityCapture inside the cook (reference src/SOP_FaceDeform.cpp:301-322): with the mesh's
    edge adjacency and the rig's triangles on the geometry and no dist2 array of the caller's,
    fdsop_cook runs islands -> dist2 -> gate / fall-off -> deformation on the device.
    Checked stage by stage against the oracle (fdo_capture_islands, fdo_capture_dist2, fdo_deform)
    and end to end; plus the reference's caching: a radius / maxedges / dofalloff change alone does
    NOT re-capture (the FIXME at :309), a rest-rig or rest-pose change does."""
    from facedeform_amd.sop import FaceDeformSOP
    N, M, K = 30_000, 40, 6
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 1)
    offsets, nb = _knn_adjacency(P)
    tris = _rig_triangles(rest)
    node = FaceDeformSOP()
    node.set("kernel", 1)                      # thin-plate
    node.set("radius", 0.2); node.set("maxedges", K); node.set("dofalloff", 1); node.set("falloffrate", 1.5)
    kw = dict(edge_offsets=offsets, edge_neighbours=nb, rig_tris=tris, want_dist2=True)
    res = node.cook(P, rest, deform, **kw)
    assert res.severity == capi.FDSOP_MESSAGE, res.messages          # no "Can't find distance capture attribute"
    r2 = np.float32(0.2) * np.float32(0.2)
    # -- stage 1+2: the captured attribute against the oracle's capture
    mask = oracle.capture_islands(P, offsets, nb, rest, K)
    d2_ref = oracle.capture_dist2(P, tris, r2, True, mask)
    exact = oracle.capture_dist2(P, tris, 1e30, True)
    margin = 2e-6 * (exact + 1.0)
    decided = np.abs(exact - r2) > margin
    got = res.dist2
    assert np.all(got[mask == 0] == 0.0) and mask.sum() > M
    assert np.all(got[decided & (d2_ref == -1)] == -1.0)
    sel = decided & (d2_ref >= 0) & (mask == 1)
    assert sel.sum() > 50 and np.all(np.abs(got[sel] - d2_ref[sel]) <= margin[sel])
    # -- stage 3: the deformation that consumed it, oracle fed with the device's attribute: 1e-5
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [0.0], 0)
    ref, ref_fall = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=got, radius2=r2, falloffrate=1.5)
    assert parity_ratio(res.P, ref, P, 1e-5) <= 1.0
    assert np.allclose(res.fd_falloff, ref_fall, rtol=2e-6, atol=1e-7)
    # -- end to end: oracle capture -> oracle deform, nothing of the device's in between.  The bar is
    # the capture's: an fp32 distance within 2e-6 (d2 + L^2) moves the fall-off (1 - d2 / r2)^1.5 by up
    # to 1.5 * margin / r2 = 8e-5 of the displacement at r2 = 0.04.
    ref_e2e, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=d2_ref, radius2=r2, falloffrate=1.5)
    plain, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    assert parity_ratio(res.P[decided], ref_e2e[decided], P[decided], 1e-4, scale_out=plain[decided]) <= 1.0
    # -- caching (B12): parms alone do not re-capture ...
    node.set("radius", 0.35); node.set("maxedges", 2); node.set("dofalloff", 0)
    res2 = node.cook(P, rest, deform, rig_rest_unchanged=True, mesh_unchanged=True, **kw)
    assert np.array_equal(res2.dist2, got)
    # ... (the gate and the fall-off do use the new radius: :402, :423)
    r2b = np.float32(0.35) * np.float32(0.35)
    ref2, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=got, radius2=r2b, falloffrate=1.5)
    assert parity_ratio(res2.P, ref2, P, 1e-5) <= 1.0
    # ... a rest-rig change does: dofalloff is off now, so every island point reads 0 (capture.cpp:71-75)
    res3 = node.cook(P, rest, deform, rig_rest_unchanged=False, mesh_unchanged=True, **kw)
    assert np.all(res3.dist2 == 0.0)
    ref3, fall3 = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=np.zeros(N, np.float32), radius2=r2b, falloffrate=1.5)
    assert parity_ratio(res3.P, ref3, P, 1e-5) <= 1.0 and np.array_equal(res3.fd_falloff, fall3)
    # ... and so does a new mesh (rest pose changed): fall-off on again, the wider radius, two rings
    node.set("dofalloff", 1)
    P2 = (P * np.float32(1.01)).astype(np.float32)
    res4 = node.cook(P2, rest, deform, rig_rest_unchanged=True, mesh_unchanged=False, **kw)
    mask4 = oracle.capture_islands(P2, offsets, nb, rest, 2)
    d4 = oracle.capture_dist2(P2, tris, r2b, True, mask4)
    ex4 = oracle.capture_dist2(P2, tris, 1e30, True)
    dec4 = np.abs(ex4 - r2b) > 2e-6 * (ex4 + 1.0)
    assert np.all(res4.dist2[mask4 == 0] == 0.0) and mask4.sum() < mask.sum()
    s4 = dec4 & (d4 >= 0) & (mask4 == 1)
    assert np.all(np.abs(res4.dist2[s4] - d4[s4]) <= 2e-6 * (ex4[s4] + 1.0))
    # a radius inside the islands' extent: island points beyond it read -1 (capture.cpp:76,88), pass the
    # gate and overshoot (B4: pow(1 - (-1 / r2), rate) > 1)
    node.set("radius", 0.03); node.set("maxedges", K)
    res6 = node.cook(P, rest, deform, rig_rest_unchanged=False, mesh_unchanged=False, **kw)     # back to the first mesh
    r2c = np.float32(0.03) * np.float32(0.03)
    assert (res6.dist2 == -1.0).any() and (res6.fd_falloff > 1.0).any()
    ref6, fall6 = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=res6.dist2, radius2=r2c, falloffrate=1.5)
    assert parity_ratio(res6.P, ref6, P, 1e-5) <= 1.0 and np.allclose(res6.fd_falloff, fall6, rtol=2e-6, atol=1e-7)
    # without the capture's inputs and without a dist2 array: the reference's warning, no gate, no fall-off
    res5 = node.cook(P, rest, deform)
    assert res5.warnings == ["Can't find distance capture attribute. Won't apply radius nor falloff."]
    assert np.array_equal(res5.fd_falloff, np.ones(N, np.float32))
    node.close()

"""GPU parity for BASELINE.json configs[1] (PointPillars KITTI: DynamicPillarVFE scatter + pillar pseudo-image + rotated NMS)
and the other dynamic VFEs, through the C ABI, against oracle/dynamic_vfe.py on identical seeded clouds.
Bar: voxel coordinates, point->voxel map, counts, scatter_max, decorated point features and the pseudo-image bit-exact;
scatter_mean rel 1e-6 (the reference's own binary32 atomic sum is only reproducible to ~1e-6); module outputs that pass through
Linear/BatchNorm rel-L2 1e-5 against the same layers evaluated by torch on the CPU."""
import numpy as np
import pytest
import torch

from oracle import dynamic_vfe as O
from oracle import loader
from openpcdet_amd import model_nms_utils, scatter, synth, vfe

pytestmark = pytest.mark.gpu


def _batch_cloud(name, seeds, n_points=None):
    frames = [synth.config_cloud(name, s, n_points=n_points) for s in seeds]
    return np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])


def _check_voxelize(pts, vs, rg, grid, ndim, dev, batch=None):
    want = O.dynamic_voxelize(pts, vs, rg, grid, ndim)
    got = vfe.dynamic_voxelize(torch.from_numpy(pts).to(dev), vs, rg, grid, ndim, batch)
    m = len(want["unq_cnt"])
    assert got["num_voxels"] == m
    pv = got["point_voxel"].cpu().numpy()
    assert np.array_equal(pv >= 0, want["keep"])
    assert np.array_equal(pv[want["keep"]], want["unq_inv"])
    assert np.array_equal(got["voxel_coords"].cpu().numpy(), want["voxel_coords"])
    assert np.array_equal(got["voxel_num_points"].cpu().numpy(), want["unq_cnt"])
    cell = got["point_cell"].cpu().numpy()
    assert np.array_equal(cell[want["keep"]][:, :ndim], want["coords"]) and (cell[~want["keep"]] == -1).all()
    return want, got


def test_dynamic_voxelize_small_ragged(cuda_device):
    r = np.random.default_rng(0)
    pts = np.empty((7000, 5), np.float32)
    pts[:, 0] = r.integers(0, 3, 7000)
    pts[:, 1:4] = r.uniform(-1.3, 1.3, (7000, 3))
    pts[:, 4] = r.uniform(0, 1, 7000)
    pts[::101, 1] = np.nan
    for ndim in (3, 2):
        _check_voxelize(pts, [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], [20, 10, 4], ndim, cuda_device)
    # one point / every point in one cell / every point outside / empty
    _check_voxelize(pts[:1], [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], [20, 10, 4], 3, cuda_device, batch=3)
    same = np.tile(np.array([[1, 0.01, 0.01, 0.01, 1.0]], np.float32), (1000, 1))
    w, _ = _check_voxelize(same, [0.1, 0.1, 0.1], [-1, -1, -1, 1, 1, 1], [20, 20, 20], 3, cuda_device)
    assert w["unq_cnt"].tolist() == [1000]
    out = pts.copy(); out[:, 1:4] += 100
    got = vfe.dynamic_voxelize(torch.from_numpy(out).to(cuda_device), [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], [20, 10, 4], 3, 3)
    assert got["num_voxels"] == 0 and (got["point_voxel"] == -1).all()
    got = vfe.dynamic_voxelize(torch.zeros((0, 5), device=cuda_device), [0.1, 0.2, 0.5], [-1, -1, -1, 1, 1, 1], [20, 10, 4], 3, 1)
    assert got["num_voxels"] == 0 and got["voxel_coords"].shape == (0, 4)


@pytest.mark.parametrize("name,ndim", [("kitti_pillar", 2), ("kitti_second", 3), ("nuscenes_centerpoint", 3), ("argo2_voxelnext", 3)])
def test_dynamic_voxelize_config_frames(name, ndim, cuda_device):
    cfg = synth.CONFIGS[name]
    pts = _batch_cloud(name, [1, 2], n_points=60000)
    grid = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    w, _ = _check_voxelize(pts, cfg["voxel"], cfg["range"], grid, ndim, cuda_device, batch=2)
    assert len(w["unq_cnt"]) > 1000


def test_scatter_mean_max(cuda_device):
    r = np.random.default_rng(2)
    n, m = 200000, 9000
    idx = np.minimum(r.geometric(0.001, n), m - 50).astype(np.int64)     # heavy-tailed segment sizes, some segments empty
    for c in (1, 3, 64):
        src = r.normal(size=(n, c)).astype(np.float32)
        d_src, d_idx = torch.from_numpy(src).to(cuda_device), torch.from_numpy(idx).to(cuda_device)
        mean = scatter.scatter_mean(d_src, d_idx, dim=0, dim_size=m).cpu().numpy()
        assert np.allclose(mean, O.scatter_mean(src, idx, m), rtol=1e-6, atol=1e-7)
        mx, arg = scatter.scatter_max(d_src, d_idx, dim=0, dim_size=m)
        w_mx, w_arg = O.scatter_max(src, idx, m)
        assert np.array_equal(mx.cpu().numpy().view(np.uint32), w_mx.view(np.uint32))
        assert np.array_equal(arg.cpu().numpy(), w_arg)
    # torch_scatter call shape: dim_size inferred, int32 index, negative index skipped
    d_idx32 = torch.from_numpy(idx).to(cuda_device).int()
    d_idx32[:10] = -1
    got = scatter.scatter_max(d_src, d_idx32)[0]
    assert got.shape[0] == int(idx.max()) + 1


def test_scatter_autograd(cuda_device):
    torch.manual_seed(0)
    n, c, m = 3000, 8, 200
    idx = torch.randint(0, m, (n,), device=cuda_device)
    src = torch.randn(n, c, device=cuda_device, requires_grad=True)
    w = torch.randn(m, c, device=cuda_device)
    (scatter.scatter_max(src, idx, dim_size=m)[0] * w).sum().backward()
    g_max = src.grad.clone(); src.grad = None
    ref = torch.zeros(m, c, device=cuda_device).scatter_reduce(0, idx.unsqueeze(1).expand(n, c), src, 'amax', include_self=False)
    (ref * w).sum().backward()
    assert torch.allclose(g_max, src.grad); src.grad = None
    (scatter.scatter_mean(src, idx, dim_size=m) * w).sum().backward()
    g_mean = src.grad.clone(); src.grad = None
    cnt = torch.bincount(idx, minlength=m).clamp(min=1).float().unsqueeze(1)
    ((torch.zeros(m, c, device=cuda_device).index_add(0, idx, src) / cnt) * w).sum().backward()
    assert torch.allclose(g_mean, src.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("z_uses_cell,use_abs,with_dist", [(0, 1, 0), (1, 1, 1), (1, 0, 0)])
def test_point_features_bit_exact(z_uses_cell, use_abs, with_dist, cuda_device):
    cfg = synth.CONFIGS["kitti_pillar"]
    pts = _batch_cloud("kitti_pillar", [3], n_points=50000)
    grid = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    ndim = 3 if z_uses_cell else 2
    w = O.dynamic_voxelize(pts, cfg["voxel"], cfg["range"], grid, ndim)
    g = vfe.dynamic_voxelize(torch.from_numpy(pts).to(cuda_device), cfg["voxel"], cfg["range"], grid, ndim, 1)
    kept = pts[w["keep"]]
    mean = O.scatter_mean(kept[:, 1:4], w["unq_inv"], len(w["unq_cnt"]))
    want = O.point_features(kept, w["coords"] if ndim == 3 else np.concatenate([w["coords"], np.zeros((len(kept), 1), np.int32)], 1),
                            w["unq_inv"], mean, cfg["voxel"], cfg["range"], z_uses_cell, use_abs, with_dist)
    got = vfe.dynamic_point_features(torch.from_numpy(pts).to(cuda_device), g["point_voxel"], g["point_cell"],
                                     torch.from_numpy(mean).to(cuda_device), cfg["voxel"], cfg["range"], z_uses_cell, use_abs,
                                     with_dist).cpu().numpy()
    assert (got[~w["keep"]] == 0).all()
    k = want.shape[1] - (1 if with_dist else 0)
    assert np.array_equal(got[w["keep"]][:, :k].view(np.uint32), want[:, :k].view(np.uint32))
    if with_dist:   # torch.norm's own reduction order is unspecified: 1 ulp
        assert np.allclose(got[w["keep"]][:, k], want[:, k], rtol=2e-7)


def _cpu_pfn(mod, feats, inv, m):
    """the PFN layers of `mod` evaluated by torch on the CPU with the oracle's scatter_max"""
    x = torch.from_numpy(feats)
    for pfn in mod.pfn_layers:
        y = torch.relu(pfn.norm(pfn.linear(x)) if pfn.use_norm else pfn.linear(x)).detach().numpy()
        mx = O.scatter_max(y, inv, m)[0]
        if pfn.last_vfe:
            return mx
        x = torch.from_numpy(np.concatenate([y, mx[inv]], 1))


@pytest.mark.parametrize("cls,ndim,filters", [(vfe.DynamicPillarVFE, 2, [64]), (vfe.DynamicVoxelVFE, 3, [32, 64]),
                                             (vfe.DynamicPillarVFESimple2D, 2, [32])])
def test_dynamic_vfe_modules(cls, ndim, filters, cuda_device):
    """configs[1]: DynamicPillarVFE (10 -> 64) on a KITTI PointPillars batch, then the pseudo-image"""
    torch.manual_seed(5)
    cfg = synth.CONFIGS["kitti_pillar"]
    pts = _batch_cloud("kitti_pillar", [1, 2], n_points=120000)
    grid = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    mc = dict(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=filters)
    cpu = cls(mc, 4, cfg["voxel"], grid, cfg["range"]).eval()
    for pfn in cpu.pfn_layers:
        pfn.norm.running_mean.normal_(0, 0.3); pfn.norm.running_var.uniform_(0.5, 2.0)
        pfn.norm.weight.data.uniform_(0.5, 1.5); pfn.norm.bias.data.normal_(0, 0.2)
    dev = cls(mc, 4, cfg["voxel"], grid, cfg["range"]).eval()
    dev.load_state_dict(cpu.state_dict())
    dev = dev.to(cuda_device)
    with torch.no_grad():
        out = dev({'points': torch.from_numpy(pts).to(cuda_device), 'batch_size': 2})
    w = O.dynamic_voxelize(pts, cfg["voxel"], cfg["range"], grid, ndim)
    m = len(w["unq_cnt"])
    kept = pts[w["keep"]]
    coords3 = w["coords"] if ndim == 3 else np.concatenate([w["coords"], np.zeros((len(kept), 1), np.int32)], 1)
    simple = cls is vfe.DynamicPillarVFESimple2D
    mean = np.zeros((m, 3), np.float32) if simple else O.scatter_mean(kept[:, 1:4], w["unq_inv"], m)
    f = O.point_features(kept, coords3, w["unq_inv"], mean, cfg["voxel"], cfg["range"], ndim == 3, True, False)
    if simple:
        f = np.concatenate([f[:, 7:10], f[:, :4]], 1)
    want = _cpu_pfn(cpu, f, w["unq_inv"], m)
    got = out['pillar_features'].cpu().numpy()
    assert got.shape == want.shape == (m, filters[-1])
    assert np.linalg.norm(got - want) <= 1e-5 * np.linalg.norm(want)
    if simple:
        assert np.array_equal(out['pillar_coords'].cpu().numpy(), w["voxel_coords"][:, [0, 2, 3]])
        return
    assert np.array_equal(out['voxel_coords'].cpu().numpy(), w["voxel_coords"])
    if ndim == 2:   # pillar pseudo-image [B, 64, 496, 432]
        sc = vfe.PointPillarScatter(dict(NUM_BEV_FEATURES=filters[-1]), grid)
        img = sc(out)['spatial_features']
        assert tuple(img.shape) == (2, filters[-1], grid[1], grid[0])
        assert np.array_equal(img.cpu().numpy(), O.pointpillar_scatter(got, w["voxel_coords"], grid[0], grid[1]))


def test_dynamic_mean_vfe_and_scatter3d(cuda_device):
    cfg = synth.CONFIGS["kitti_second"]
    pts = _batch_cloud("kitti_second", [4, 5], n_points=80000)
    grid = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    mod = vfe.DynamicMeanVFE(None, 4, cfg["voxel"], grid, cfg["range"])
    out = mod({'points': torch.from_numpy(pts).to(cuda_device), 'batch_size': 2})
    w = O.dynamic_voxelize(pts, cfg["voxel"], cfg["range"], grid, 3)
    assert np.array_equal(out['voxel_coords'].cpu().numpy(), w["voxel_coords"])
    want = O.scatter_mean(pts[w["keep"]][:, 1:], w["unq_inv"], len(w["unq_cnt"]))
    assert np.allclose(out['voxel_features'].cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    # PointPillarScatter3d on a coarse grid
    r = np.random.default_rng(0)
    nx, ny, nz, C = 16, 12, 4, 8
    cells = r.choice(2 * nz * ny * nx, 500, replace=False)
    coords = np.stack([cells // (nz * ny * nx), cells // (ny * nx) % nz, cells // nx % ny, cells % nx], 1).astype(np.int32)
    feats = r.normal(size=(500, C)).astype(np.float32)
    sc = vfe.PointPillarScatter3d(dict(INPUT_SHAPE=[nx, ny, nz], NUM_BEV_FEATURES=C * nz), None)
    img = sc({'pillar_features': torch.from_numpy(feats).to(cuda_device), 'voxel_coords': torch.from_numpy(coords).to(cuda_device)})
    assert np.array_equal(img['spatial_features'].cpu().numpy(), O.pointpillar_scatter(feats, coords, nx, ny, nz))


def test_pillar_vfe_hard(cuda_device):
    """hard-pillar path of configs[1]: PointToVoxel (T=32) -> PillarVFE -> PointPillarScatter; PillarVFE is dense torch ops on the
    voxelizer's output, checked against the same module evaluated on the CPU from the ORACLE voxelizer's output"""
    from openpcdet_amd import voxelize
    torch.manual_seed(1)
    cfg = synth.CONFIGS["kitti_pillar"]
    pts = synth.config_cloud("kitti_pillar", 6)
    grid = loader.voxel_grid_size(cfg["voxel"], cfg["range"])
    mc = dict(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64])
    cpu = vfe.PillarVFE(mc, 4, cfg["voxel"], cfg["range"]).eval()
    dev = vfe.PillarVFE(mc, 4, cfg["voxel"], cfg["range"]).eval()
    dev.load_state_dict(cpu.state_dict()); dev = dev.to(cuda_device)
    gen = voxelize.PointToVoxel(cfg["voxel"], cfg["range"], 4, cfg["max_voxels"], cfg["T"])
    v, c, n = gen(torch.from_numpy(pts).to(cuda_device))
    c4 = torch.cat([torch.zeros_like(c[:, :1]), c], 1)
    with torch.no_grad():
        got = dev({'voxels': v, 'voxel_num_points': n, 'voxel_coords': c4})['pillar_features'].cpu().numpy()
        wv, wc, wn, _ = loader.voxelize_hard(pts, cfg["voxel"], cfg["range"], cfg["T"], cfg["max_voxels"])
        wc4 = np.concatenate([np.zeros((len(wc), 1), np.int32), wc], 1)
        want = cpu({'voxels': torch.from_numpy(wv), 'voxel_num_points': torch.from_numpy(wn),
                    'voxel_coords': torch.from_numpy(wc4)})['pillar_features'].numpy()
    assert got.shape == want.shape and np.linalg.norm(got - want) <= 1e-5 * np.linalg.norm(want)
    img = vfe.PointPillarScatter(dict(NUM_BEV_FEATURES=64), grid)({'pillar_features': torch.from_numpy(want).to(cuda_device),
                                                                    'voxel_coords': c4})['spatial_features']
    assert np.array_equal(img.cpu().numpy(), O.pointpillar_scatter(want, wc4, grid[0], grid[1]))


def _np_nms(boxes, scores, thresh, pre, post):
    order = np.argsort(-scores, kind="stable")[:pre]
    keep, _ = loader.nms(boxes[order][:, :7].copy(), thresh)
    return order[keep[:post]]


def test_model_nms_utils(cuda_device):
    """class_agnostic_nms / multi_classes_nms / class_specific_nms vs a numpy restatement over the oracle NMS
    (configs[1]: 4096 boxes, thresh 0.01, post 500, score 0.1 -- second.yaml:94-99)"""
    cfg = synth.CONFIGS["kitti_pillar"]
    n = 6000
    b = synth.boxes(n, 9, cfg["range"], 7, clustered=300)[0]
    r = np.random.default_rng(8)
    scores = r.permutation(n).astype(np.float32) / n      # distinct
    nms_cfg = dict(NMS_TYPE='nms_gpu', NMS_THRESH=0.01, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500, MULTI_CLASSES_NMS=False)
    d_b, d_s = torch.from_numpy(b).to(cuda_device), torch.from_numpy(scores).to(cuda_device)
    sel, sel_scores = model_nms_utils.class_agnostic_nms(d_s, d_b, nms_cfg, score_thresh=0.1)
    passed = np.nonzero(scores >= 0.1)[0]
    want = passed[_np_nms(b[passed], scores[passed], 0.01, 4096, 500)]
    assert np.array_equal(sel.cpu().numpy(), want) and np.array_equal(sel_scores.cpu().numpy(), scores[want])
    sel2, _ = model_nms_utils.class_agnostic_nms(d_s, d_b, nms_cfg)
    assert np.array_equal(sel2.cpu().numpy(), _np_nms(b, scores, 0.01, 4096, 500))
    # per-class columns
    cls_scores = np.stack([scores, np.roll(scores, 17), np.roll(scores, 999)], 1)
    ps, pl, pb = model_nms_utils.multi_classes_nms(torch.from_numpy(cls_scores).to(cuda_device), d_b, nms_cfg, score_thresh=0.3)
    ws, wl, wb = [], [], []
    for k in range(3):
        passed = np.nonzero(cls_scores[:, k] >= 0.3)[0]
        s = passed[_np_nms(b[passed], cls_scores[passed, k], 0.01, 4096, 500)]
        ws.append(cls_scores[s, k]); wl.append(np.full(len(s), k)); wb.append(b[s])
    assert np.array_equal(ps.cpu().numpy(), np.concatenate(ws)) and np.array_equal(pl.cpu().numpy(), np.concatenate(wl))
    assert np.array_equal(pb.cpu().numpy(), np.concatenate(wb))
    # class-specific thresholds (VoxelNeXt / CenterPoint class_specific_nms)
    labels = r.integers(0, 3, n)
    cs = dict(NMS_THRESH=[0.2, 0.5, 0.05], NMS_PRE_MAXSIZE=[1000, 2000, 500], NMS_POST_MAXSIZE=[83, 100, 50])
    sel3, sc3 = model_nms_utils.class_specific_nms(d_s, d_b[:, :7].contiguous(), torch.from_numpy(labels).to(cuda_device), cs,
                                                   score_thresh=0.2)
    want3 = []
    for k in range(3):
        rows = np.nonzero((labels == k) & (scores > 0.2))[0]
        want3.append(rows[_np_nms(b[rows], scores[rows], cs["NMS_THRESH"][k], cs["NMS_PRE_MAXSIZE"][k], 10 ** 9)])
    assert np.array_equal(sel3.cpu().numpy(), np.concatenate(want3))

"""Seeded synthetic inputs of the shapes SURVEY.md 8d names (no dataset exists offline).

lidar_cloud(): a spinning-LiDAR-like scan -- `beams` elevation rings, uniform azimuth, range from intersecting
each ray with a ground plane and a set of random box obstacles, several sweeps with ego-motion offsets and a
timestamp channel.  Used identically by tests, bench.py and the CPU baseline.
"""
import numpy as np

CONFIGS = {
    # name: range, voxel, T, max_voxels, C
    "kitti_second": dict(range=[0, -40, -3, 70.4, 40, 1], voxel=[0.05, 0.05, 0.1], T=5, max_voxels=40000, C=4,
                         n_points=120000, beams=64, sweeps=1, elev=(-24.8, 2.0), sensor_z=1.73),
    "kitti_pillar": dict(range=[0, -39.68, -3, 69.12, 39.68, 1], voxel=[0.16, 0.16, 4], T=32, max_voxels=40000, C=4,
                         n_points=120000, beams=64, sweeps=1, elev=(-24.8, 2.0), sensor_z=1.73),
    "nuscenes_centerpoint": dict(range=[-54, -54, -5, 54, 54, 3], voxel=[0.075, 0.075, 0.2], T=10, max_voxels=160000,
                                 C=5, n_points=300000, beams=32, sweeps=10, elev=(-30.0, 10.0), sensor_z=1.84),
    "waymo_centerpoint": dict(range=[-75.2, -75.2, -2, 75.2, 75.2, 4], voxel=[0.1, 0.1, 0.15], T=5, max_voxels=180000,
                              C=6, n_points=1000000, beams=64, sweeps=4, elev=(-17.6, 2.4), sensor_z=2.0),
    "argo2_voxelnext": dict(range=[-200, -200, -20, 200, 200, 20], voxel=[0.1, 0.1, 0.2], T=20, max_voxels=160000,
                            C=4, n_points=100000, beams=64, sweeps=1, elev=(-25.0, 15.0), sensor_z=1.7),
}


def lidar_cloud(n_points, C, pc_range, seed, beams=32, sweeps=1, elev=(-30.0, 10.0), sensor_z=1.84, n_obstacles=40,
                max_range=None):
    """-> float32 [n_points, C]: x, y, z, intensity[, timestamp][, elongation...] in the sensor-centred frame whose
    ground plane is z = -sensor_z."""
    rng = np.random.default_rng(seed)
    pc_range = np.asarray(pc_range, np.float64)
    if max_range is None:
        max_range = float(np.max(np.abs(pc_range[[0, 1, 3, 4]]))) * 1.2
    per_sweep = int(np.ceil(n_points / sweeps))
    # obstacles: boxes standing on the ground
    obs_c = rng.uniform(-0.8 * max_range, 0.8 * max_range, (n_obstacles, 2))
    obs_r = rng.uniform(0.8, 3.0, n_obstacles)
    obs_h = rng.uniform(1.0, 3.5, n_obstacles)
    chunks = []
    for s in range(sweeps):
        az = rng.uniform(0, 2 * np.pi, per_sweep)
        beam = rng.integers(0, beams, per_sweep)
        el = np.deg2rad(elev[0] + (elev[1] - elev[0]) * (beam + 0.5) / beams) + rng.normal(0, 2e-4, per_sweep)
        dx, dy, dz = np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)
        # ground hit
        with np.errstate(divide="ignore", invalid="ignore"):
            t_ground = np.where(dz < -1e-3, -sensor_z / dz, np.inf)
        t = np.minimum(t_ground, max_range)
        # obstacle hit (vertical cylinders): nearest positive root in the xy-plane
        a = dx * dx + dy * dy
        for k in rng.choice(n_obstacles, size=min(n_obstacles, 12), replace=False):
            cx, cy = obs_c[k]
            b = -2 * (dx * cx + dy * cy)
            c = cx * cx + cy * cy - obs_r[k] ** 2
            disc = b * b - 4 * a * c
            with np.errstate(invalid="ignore"):
                root = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
            zhit = root * dz
            ok = (disc > 0) & (root > 0.5) & (zhit > -sensor_z) & (zhit < -sensor_z + obs_h[k])
            t = np.where(ok & (root < t), root, t)
        t = t * (1 + rng.normal(0, 1e-3, per_sweep))
        ego = np.array([0.4 * s, 0.02 * s, 0.0])      # ego motion between sweeps
        xyz = np.stack([dx * t, dy * t, dz * t], 1) - ego
        cols = [xyz, rng.uniform(0, 1, (per_sweep, 1))]
        if C >= 5:
            cols.append(np.full((per_sweep, 1), 0.05 * s))
        if C >= 6:
            cols.append(rng.uniform(0, 1, (per_sweep, C - 5)))
        chunks.append(np.concatenate(cols, 1)[:, :C])
    pts = np.concatenate(chunks, 0)[:n_points]
    rng.shuffle(pts, axis=0)      # DataProcessor.shuffle_points (data_processor.py:102-112)
    return np.ascontiguousarray(pts, np.float32)


def config_cloud(name, seed, n_points=None):
    cfg = CONFIGS[name]
    return lidar_cloud(n_points or cfg["n_points"], cfg["C"], cfg["range"], seed, beams=cfg["beams"], sweeps=cfg["sweeps"],
                       elev=cfg["elev"], sensor_z=cfg["sensor_z"])


def boxes(n, C, pc_range, seed, clustered=0):
    """-> (boxes [n, C>=7] float32, scores [n] distinct float32) like decoded detections."""
    rng = np.random.default_rng(seed)
    b = np.zeros((n, C), np.float32)
    lo, hi = np.asarray(pc_range[:2]), np.asarray(pc_range[3:5])
    if clustered:
        ctr = rng.uniform(lo, hi, (clustered, 2))
        pick = rng.integers(0, clustered, n)
        b[:, 0:2] = ctr[pick] + rng.normal(0, 0.7, (n, 2))
        b[:, 6] = rng.uniform(-np.pi, np.pi, clustered)[pick] + rng.normal(0, 0.2, n)
    else:
        b[:, 0:2] = rng.uniform(lo, hi, (n, 2))
        b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3:6] = np.abs(rng.normal((3.9, 1.6, 1.56), 0.2, (n, 3))) + 0.1
    if C > 7:
        b[:, 7:] = rng.normal(0, 1, (n, C - 7))
    scores = (rng.permutation(n).astype(np.float32) + 1) / (n + 1)
    return b, scores

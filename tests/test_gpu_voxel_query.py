"""pvx_voxel_query through the voxel_query drop-in against the oracle: indices and empty flags bit-exact."""
import numpy as np
import pytest
import torch

from openpcdet_amd import voxel_query as vq
from openpcdet_amd.spconv import pytorch as spconv
from oracle import voxel_query as ovq
from test_oracle_voxel_query import make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,rng,radius,ns", [(0, (1, 2, 2), 0.5, 16), (1, (2, 4, 4), 0.8, 8), (2, (0, 1, 1), 0.25, 4), (3, (4, 4, 4), 1.6, 32)])
def test_voxel_query_matches_oracle(seed, rng, radius, ns, cuda_device):
    coords, xyz, v2p, qc, qxyz = make_case(seed, m=3000)
    t = lambda a: torch.from_numpy(a).to(cuda_device)
    st = spconv.SparseConvTensor(torch.zeros((len(coords), 1), device=cuda_device), t(coords), list(v2p.shape[1:]), v2p.shape[0])
    dense = vq.generate_voxel2pinds(st)
    assert dense.dtype == torch.int32 and np.array_equal(dense.cpu().numpy(), v2p)
    idx, empty = vq.voxel_query(rng, radius, ns, t(xyz), t(qxyz), t(qc), dense)
    want_idx, want_empty = ovq.voxel_query(rng, radius, ns, xyz, qxyz, qc, v2p)
    assert idx.dtype == torch.int32 and empty.dtype == torch.bool
    assert np.array_equal(empty.cpu().numpy(), want_empty)
    assert np.array_equal(idx.cpu().numpy(), want_idx)


def test_voxel_query_edge_cases(cuda_device):
    coords, xyz, v2p, qc, qxyz = make_case(5, m=50)
    t = lambda a: torch.from_numpy(a).to(cuda_device)
    # no queries
    idx, empty = vq.voxel_query((1, 1, 1), 0.5, 8, t(xyz), t(qxyz[:0]), t(qc[:0]), t(v2p))
    assert idx.shape == (0, 8) and empty.shape == (0,)
    # an empty grid: everything is an empty ball, rows zeroed
    idx, empty = vq.voxel_query((1, 1, 1), 0.5, 8, t(xyz), t(qxyz), t(qc), t(np.full_like(v2p, -1)))
    assert bool(empty.all()) and int(idx.abs().sum()) == 0

/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for point-cloud voxelization (hard, first-come) and the mean VFE.
 *
 * PARITY UNPINNED by the reference: the algorithm lives in the third-party package `spconv`
 * (spconv.utils.Point2VoxelCPU3d; pins found in the reference: spconv_cu114==2.2.6 in
 * libraries/3DSparseConvolution/tool/requirements.txt:8, unpinned spconv-cu102 in docker/Dockerfile:55), which
 * is neither vendored under /root/reference nor installed, and the reference ships no fixture for it
 * (SURVEY.md section 4 / 8c).  This file restates spconv's published behaviour as written down in SURVEY.md
 * 8a-1 and is anchored on the reference's own call sites:
 *   VoxelGeneratorWrapper.generate                pcdet/datasets/processor/data_processor.py:45-61
 *   DataProcessor.transform_points_to_voxels      pcdet/datasets/processor/data_processor.py:200-247
 *   MeanVFE.forward                               pcdet/models/backbones_3d/vfe/mean_vfe.py:14-33
 *
 * Algorithm (sequential, first-come):
 *   grid_j = round((max_j - min_j) / vsize_j)                                   (binary32)
 *   for point i in input order:
 *       c_j = floor((p_j - min_j) / vsize_j) for j in x,y,z (binary32); drop the point if any c_j is outside
 *       [0, grid_j)
 *       look the cell up; if unseen: if num_voxels == max_voxels skip the point, else id = num_voxels++,
 *       coords[id] = (z, y, x)
 *       if num_points[id] < T: voxels[id][num_points[id]++] = all C features of the point
 *   unfilled slots stay zero.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

void orc_voxel_grid_size(const float* vsize_xyz, const float* range_xyz, int* grid_xyz) {
    for (int j = 0; j < 3; j++) grid_xyz[j] = (int)roundf((range_xyz[3 + j] - range_xyz[j]) / vsize_xyz[j]);
}

/* returns the number of voxels.  voxels [max_voxels, T, c] and num_per_voxel [max_voxels] are zeroed here.
 * point_to_voxel [n] (optional): voxel id the point's cell was given, or -1 if the point was dropped because it is out
 * of range or its cell arrived after the voxel cap (points beyond T in a kept voxel still report the voxel). */
int orc_voxelize_hard(const float* points, int n, int c, const float* vsize_xyz, const float* range_xyz, int T,
                      int max_voxels, float* voxels, int* coords_zyx, int* num_per_voxel, int* point_to_voxel) {
    int grid[3];
    orc_voxel_grid_size(vsize_xyz, range_xyz, grid);
    size_t cap = 1024;
    while (cap < (size_t)n * 2 + 2) cap <<= 1;
    uint64_t* keys = (uint64_t*)malloc(cap * sizeof(uint64_t));
    int* vals = (int*)malloc(cap * sizeof(int));
    memset(keys, 0xff, cap * sizeof(uint64_t));
    memset(voxels, 0, (size_t)max_voxels * T * c * sizeof(float));
    memset(num_per_voxel, 0, (size_t)max_voxels * sizeof(int));
    int nv = 0;
    for (int i = 0; i < n; i++) {
        int cc[3];
        int ok = 1;
        for (int j = 0; j < 3; j++) {
            float f = floorf((points[(size_t)i * c + j] - range_xyz[j]) / vsize_xyz[j]);
            if (!(f >= 0.f && f < (float)grid[j])) { ok = 0; break; }   /* also rejects NaN / out-of-int-range */
            cc[j] = (int)f;
        }
        if (point_to_voxel) point_to_voxel[i] = -1;
        if (!ok) continue;
        uint64_t key = ((uint64_t)cc[2] * (uint64_t)grid[1] + (uint64_t)cc[1]) * (uint64_t)grid[0] + (uint64_t)cc[0];
        size_t h = (size_t)(mix64(key) & (cap - 1));
        while (keys[h] != ~0ULL && keys[h] != key) h = (h + 1) & (cap - 1);
        int id;
        if (keys[h] == key) {
            id = vals[h];
        } else {
            if (nv >= max_voxels) continue;   /* cell never registered: a later point of it is skipped again */
            id = nv++;
            keys[h] = key; vals[h] = id;
            coords_zyx[id * 3 + 0] = cc[2]; coords_zyx[id * 3 + 1] = cc[1]; coords_zyx[id * 3 + 2] = cc[0];
        }
        if (point_to_voxel) point_to_voxel[i] = id;
        if (num_per_voxel[id] < T) {
            memcpy(voxels + ((size_t)id * T + num_per_voxel[id]) * c, points + (size_t)i * c, (size_t)c * sizeof(float));
            num_per_voxel[id]++;
        }
    }
    free(keys); free(vals);
    return nv;
}

/* MeanVFE (mean_vfe.py:26-31): sum over the T slots / max(num, 1) */
void orc_mean_vfe(const float* voxels, const int* num_per_voxel, int m, int T, int c, float* out) {
    for (int v = 0; v < m; v++) {
        float norm = (float)(num_per_voxel[v] < 1 ? 1 : num_per_voxel[v]);
        for (int ch = 0; ch < c; ch++) {
            float s = 0.f;
            for (int t = 0; t < T; t++) s += voxels[((size_t)v * T + t) * c + ch];
            out[(size_t)v * c + ch] = s / norm;
        }
    }
}

/* Plain-C restatement of the reference voxelizer -- TEST INFRASTRUCTURE ONLY.
 *
 * Follows modules/data/Preprocessing.py:75-116 (`group`, the 9-channel voxelizer that
 * train.py:44 calls) and the hashing/capping policy of cpp/voxelutil.cpp:325-360:
 *   - the shuffle permutation is an explicit input (Preprocessing.py:86),
 *   - idx = (int32)(((double)xyz - low) / size), truncation toward zero (:87-90),
 *   - a voxel is created at the first appearance of its index triple (:96-99),
 *   - the first <= T points of a voxel are kept in stream order (:100-104),
 *   - centroid = sequential f64 sum over the T rows / count (:112-113),
 *   - columns 3:6 = xyz - centroid for ALL T rows, padding included (:115).
 * Used by tests and by bench.py's cpu_baseline leg; never by the product path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t k[3]; int64_t vid; } slot_t;

static uint64_t mix(int32_t a, int32_t b, int32_t c) {
    uint64_t h = (uint64_t)(uint32_t)a * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)(uint32_t)b * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= (uint64_t)(uint32_t)c * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    return h ^ (h >> 29);
}

/* voxel: cap*T*9 doubles, uidx: cap*3 doubles, cnt: cap int64 (cap >= number of voxels,
 * P always suffices).  Returns the number of voxels V, or -1 on allocation failure. */
int64_t oracle_group9(const float *pcd, int64_t ncol, const int32_t *perm, int64_t P,
                      const double *range, const double *size, int32_t T,
                      double *voxel, double *uidx, int64_t *cnt)
{
    int64_t cap = 16;
    while (cap < 2 * P + 2) cap <<= 1;
    slot_t *tab = (slot_t *)malloc((size_t)cap * sizeof(slot_t));
    if (!tab) return -1;
    for (int64_t i = 0; i < cap; i++) tab[i].vid = -1;
    int64_t V = 0;
    for (int64_t s = 0; s < P; s++) {
        const float *p = pcd + (int64_t)perm[s] * ncol;
        int32_t k[3];
        for (int a = 0; a < 3; a++)
            k[a] = (int32_t)(((double)p[a] - range[a]) / size[a]);
        uint64_t h = mix(k[0], k[1], k[2]) & (uint64_t)(cap - 1);
        while (tab[h].vid >= 0 &&
               (tab[h].k[0] != k[0] || tab[h].k[1] != k[1] || tab[h].k[2] != k[2]))
            h = (h + 1) & (uint64_t)(cap - 1);
        if (tab[h].vid < 0) {
            tab[h].k[0] = k[0]; tab[h].k[1] = k[1]; tab[h].k[2] = k[2];
            tab[h].vid = V;
            memset(voxel + V * T * 9, 0, sizeof(double) * (size_t)T * 9);
            for (int a = 0; a < 3; a++) uidx[V * 3 + a] = (double)k[a];
            cnt[V] = 0;
            V++;
        }
        int64_t v = tab[h].vid;
        if (cnt[v] < T) {
            double *row = voxel + (v * T + cnt[v]) * 9;
            row[0] = p[0]; row[1] = p[1]; row[2] = p[2];
            row[6] = p[3];
            row[7] = ncol > 4 ? p[4] : 0.0;
            row[8] = ncol > 5 ? p[5] : 0.0;
            cnt[v]++;
        }
    }
    for (int64_t v = 0; v < V; v++) {
        double c[3] = {0.0, 0.0, 0.0};
        double *blk = voxel + v * T * 9;
        for (int t = 0; t < T; t++)
            for (int a = 0; a < 3; a++) c[a] += blk[t * 9 + a];
        for (int a = 0; a < 3; a++) c[a] /= (double)cnt[v];
        for (int t = 0; t < T; t++)
            for (int a = 0; a < 3; a++) blk[t * 9 + 3 + a] = blk[t * 9 + a] - c[a];
    }
    free(tab);
    return V;
}

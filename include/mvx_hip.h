/* libmvx_hip -- C ABI of the MI355X-native MVXNet hot path (gfx950).
 *
 * This library stands where the reference's pybind11 extension stood
 * (modules/Extension.py:1-3 -> cpp/voxelutil.cpp:362-368, entry `_group`) and additionally
 * carries the GPU kernels that the reference obtained implicitly from ATen/cuDNN for the
 * modules on the hot path.
 *
 * Conventions (every entry point):
 *   - plain C types only; every data pointer is a DEVICE pointer unless its name ends in
 *     `_host`; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - returns 0 on success, a negative MVX_E* code for an argument error, or a positive
 *     hipError_t from the launch;
 *   - stream-ordered, no hidden synchronisation, no persistent allocation: scratch memory
 *     comes from the caller (`*_workspace_bytes` queries), so calls can be graph-captured;
 *   - re-entrant: no global mutable state.
 */
#ifndef MVX_HIP_H
#define MVX_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVX_OK 0
#define MVX_EINVAL (-1)   /* bad argument (null pointer, size, unsupported combination) */
#define MVX_ESIZE (-2)    /* a size exceeds what the kernel supports */

/* ABI version; bumped whenever a signature changes. */
int mvx_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Voxelizer.  Replaces cpp/voxelutil.cpp:325-360 (`_group`) together with the Python around
 * it: modules/data/Preprocessing.py:75-116 (`group`, out_channels = 9) and :57-73 (`group_`,
 * out_channels = 7).
 *
 * Frames are batched with a fixed stride: frame f owns points  pcd[f*cap_points ...] and
 * outputs [f*cap_voxels ...]; the live point count of each frame is read from DEVICE memory
 * (n_points[f]) so a preceding GPU crop can feed it without a host round trip.
 *
 *   pcd        f32 [F][cap_points][ncol]   ncol >= 4: x y z r (row col)
 *   perm       i32 [F][cap_points] or NULL shuffle permutation (Preprocessing.py:86 is an
 *                                          input here: stream position s reads point perm[s])
 *   n_points   i32 [F]
 *   ext_idx    i32 [F][cap_points][3] or NULL: precomputed voxel indices in STREAM order, the
 *                                          `idx` argument of the reference `_group`
 *                                          (voxelutil.cpp:325); NULL = computed here
 *   lo[3], size[3]                         range minimum and voxel size, f64 (Config.py:7)
 *   T                                      samplesPerVoxel (<= 64)
 *   out_channels                           9: x y z dx dy dz r row col, centroid = sequential
 *                                             f64 sum / count (Preprocessing.py:112-115)
 *                                          7: x y z dx dy dz r, centroid = f32 sum / count in
 *                                             f64 (Preprocessing.py:71-72)
 *   voxels     f32 [F][cap_voxels][T][out_channels]   (the reference's f64 rounded once to
 *                                          f32, which is what train.py:125 feeds the model)
 *   coords     i64 [F][cap_voxels][4]      (0, ix, iy, iz)  == train.py:119 layout
 *   counts     i32 [F][cap_voxels]         kept points per voxel (<= T)
 *   n_voxels   i32 [F]
 *   status     i32 [1]                     OR-ed flags: bit0 = an index fell outside the
 *                                          21-bit key range, bit1 = cap_voxels exceeded
 * cap_voxels >= cap_points always suffices.
 */
size_t mvx_voxelize_workspace_bytes(int32_t n_frames, int32_t cap_points);

int mvx_voxelize(const float *pcd, const int32_t *perm, const int32_t *n_points,
                 const int32_t *ext_idx, int32_t n_frames, int32_t cap_points, int32_t ncol,
                 double lo_x, double lo_y, double lo_z,
                 double size_x, double size_y, double size_z,
                 int32_t T, int32_t out_channels, int32_t cap_voxels,
                 float *voxels, int64_t *coords, int32_t *counts, int32_t *n_voxels,
                 int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVX_HIP_H */

// route_internal.h -- the routed render's internal entry points (csrc/render.hip), called by mnr_render_fwd (csrc/step.hip).
#pragma once
#include "common.h"

namespace mnr {

constexpr int ROUTE_PREP_MAX = 64;
struct RoutePrepSeg {
    mnr_mlp_cell *table;           // device table to write [n]
    int32_t *lists, *counts;       // [n][B], [n]
    float *sub_out;                // [n][B][out_stride]
    long B;
    int n, out_stride;
    const void *packed[ROUTE_PREP_MAX];
    const float *emb[ROUTE_PREP_MAX];
};
struct RoutePrep { RoutePrepSeg s[2]; };
int route_prepare_launch(const RoutePrep &a, hipStream_t s);
// two routing problems / two blends over the same centroids in ONE launch each (the two containers of a render pass)
struct RouteProblem {
    const float *pos; long pos_stride, B; const int32_t *n_dev; int rows_per_unit; int pos_rows;
    float *weights; int32_t *lists, *counts, *inverse;
    const float *ray_depth;      // NULL, or per-row depths: `pos` then holds RAYS (stride pos_stride, one per pos_rows rows) and a row routes on o + d * ray_depth[row]
    int depth_flip;              // ... ray_depth is stored in the opposite sample order of the rows (coarse background pass)
};
struct CombineProblem {
    float *out; long out_stride; const float *sub; long cell_stride, sub_stride; const int32_t *pos; const float *weights; long B;
    const int32_t *n_dev; int rows_per_unit;
};
int route2_launch(const RouteProblem &a, const RouteProblem &b, const float *centroids_host, int n_sub, int d0, float margin, hipStream_t s);
int combine2_launch(const CombineProblem &a, const CombineProblem &b, int n_cols, int n_sub, hipStream_t s);
int route_launch(const float *pos, long pos_stride, int pos_rows, long B, const int32_t *n_dev, int rows_per_unit, const float *centroids_host, int n_sub,
                 int d0, float margin, float *weights, int32_t *lists, int32_t *counts, int32_t *inverse, hipStream_t s);
int bg_exit_points_launch(const float *rays_bg, const int32_t *n_bg, long N_max, const float *center, const float *radius, float *out, hipStream_t s);


}  // namespace mnr

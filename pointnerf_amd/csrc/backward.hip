// backward.hip -- (stub while the forward path is brought up; replaced by the real kernels)
#include "mlp_common.h"
size_t pn_wgrad_partials_bytes() { return 256; }
int pn_agg_backward_launch(const pnerf_camera *, const pnerf_points *, const float *, const void *,
                           const float *, const float *, const int32_t *, const int32_t *, const int32_t *, int, int, int,
                           const float *, const float *, const float *, const PnSaved &, long long, float *, const pnerf_point_grads *,
                           float *, hipStream_t) { return PNERF_E_UNSUP; }

/* gsr_constants.h — every numeric constant of the rasterizer, in one place.
 *
 * Shared by the CPU oracle (oracle/gsr_oracle_impl.h) and the HIP kernels
 * (structured-gaussian-splatting_amd/csrc/). Values follow SURVEY.md Appendix A
 * (A.1 .. A.10); where the reference holds an in-tree twin it is cited.
 */
#ifndef GSR_CONSTANTS_H
#define GSR_CONSTANTS_H

#define GSR_TILE              16      /* binning tile edge in pixels (A.0)             */
#define GSR_TILE_PIXELS       256
#define GSR_NEAR_CUT          0.2     /* p_view.z <= 0.2 -> culled (A.1)               */
#define GSR_HOM_EPS           1e-7    /* w = 1/(p_hom.w + 1e-7) (A.2); utils/graphics_utils.py:28 */
#define GSR_FOV_CLAMP         1.3     /* tx/tz clamp factor (A.4)                      */
#define GSR_COV2D_DILATE      0.3     /* low-pass added to the 2D covariance diagonal  */
#define GSR_LAMBDA_FLOOR      0.1     /* max(0.1, mid^2 - det) (A.4)                   */
#define GSR_RADIUS_SIGMAS     3.0     /* radius = ceil(3 sqrt(lambda_max))             */
#define GSR_ALPHA_MAX         0.99    /* alpha clamp (A.8)                             */
#define GSR_ALPHA_MIN         (1.0 / 255.0)
#define GSR_T_CUTOFF          1e-4    /* transmittance termination (A.8)               */
#define GSR_CONIC_BWD_EPS     1e-7    /* k = 1/(den^2 + 1e-7) (A.10)                   */
#define GSR_SH_OFFSET         0.5     /* rgb = sh_eval + 0.5; gaussian_renderer/__init__.py:78 */

/* Real spherical-harmonics constants, utils/sh_utils.py:26-54 */
#define GSR_SH_C0   0.28209479177387814
#define GSR_SH_C1   0.4886025119029199
#define GSR_SH_C2_0 1.0925484305920792
#define GSR_SH_C2_1 (-1.0925484305920792)
#define GSR_SH_C2_2 0.31539156525252005
#define GSR_SH_C2_3 (-1.0925484305920792)
#define GSR_SH_C2_4 0.5462742152960396
#define GSR_SH_C3_0 (-0.5900435899266435)
#define GSR_SH_C3_1 2.890611442640554
#define GSR_SH_C3_2 (-0.4570457994644658)
#define GSR_SH_C3_3 0.3731763325901154
#define GSR_SH_C3_4 (-0.4570457994644658)
#define GSR_SH_C3_5 1.445305721320277
#define GSR_SH_C3_6 (-0.5900435899266435)

#endif /* GSR_CONSTANTS_H */

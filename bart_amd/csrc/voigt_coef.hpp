// Weideman (1994, SIAM J. Numer. Anal. 31, 1497) rational approximation of the
// Faddeeva function, N = 36: w(z) = 2 p(Z)/(L - iz)^2 + pi^-1/2 /(L - iz),
// Z = (L + iz)/(L - iz).  Coefficients of p, highest degree first (generated
// with the paper's FFT recipe, tools/gen_voigt_coef.py).  The kernels use it for |z| < 8 and
// y > 0.13 (smaller y: the expansion about the real axis, lbl.hip voigt_taylor); there the error of Re w
// is 1e-15 of the line-centre value and 3e-13 relative (over all of |z| < 8 down to y = 1e-9 it would be
// 8e-10 relative wherever the function is above 1e-6; N = 40: 1e-10).
#pragma once
namespace bartrt {
constexpr int kWeidN = 36;
constexpr double kWeidL = 5.0453784915222872;
__device__ __constant__ const double kWeidA[kWeidN] = {
    5.35354939391731272e-14, -8.06116843801410101e-14, -3.24026763416563408e-13,
    4.42984937890695336e-13, 2.09794730416171249e-12, -2.11703453357760295e-12,
    -1.43125851415249576e-11, 6.34627660937055194e-12, 9.93932734844919644e-11,
    3.19721039881697083e-11, -6.63484656720661016e-10, -9.09223809304155717e-10,
    3.77344307541904587e-09, 1.18838872102435991e-08, -1.09622779261273633e-08,
    -1.13031571986833943e-07, -1.28948429258683140e-07, 6.74165566301323994e-07,
    2.76540866563956346e-06, 1.41870584793015483e-06, -2.17411865654944552e-05,
    -8.81779714184929473e-05, -1.13966306444594309e-04, 4.62903169399885147e-04,
    3.54844470869966925e-03, 1.38982537632514024e-02, 4.10510430165768880e-02,
    1.00842933718479494e-01, 2.15016363201073951e-01, 4.07342418950334073e-01,
    6.95662191897100102e-01, 1.08135803717658874e+00, 1.54016257881536522e+00,
    2.01939764361135055e+00, 2.44537849285192088e+00, 2.74074502740986015e+00,
};
}  // namespace bartrt

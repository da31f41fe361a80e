// mpc_core.hpp - what the solver (mpc_wave.hpp), the observation preamble (mpc_preamble.hpp) and their host test
// harnesses share: solver parameters, the bounds of the NLP of PureMPC_Agent._solve (reference
// agents/pure_mpc.py:80-318), the layout of the reference-path table, the kinematic bicycle model and the lean FP64
// math the kernels use instead of the generic libm expansions.
//
// (Up to round 1 this header also held a one-lane-per-instance version of the solver; the wave-cooperative solver
// replaced it for every horizon - DESIGN.md section 5 keeps its measurements.)
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif

namespace mpc {

// reference-table columns served by WS::ref(k, c)
enum : int { R_X = 0, R_Y = 1, R_H = 2, R_SIN = 3, R_COS = 4, REF_COLS = 5 };

struct SolveParams {
    int N;         // horizon
    int V;         // other vehicles used by the collision-cost term (0 when the term is off)
    int max_iter;
    double dt;
    double tol;
    double mu_init;
    double w_distance;  // weight_distance
    int stall_window;   // mpc_config.stall_window: 0 = off
    int strict_kink;    // MPC_FLAG_STRICT_DISCONTINUITY: a solve that ends on the d = 1 discontinuity is reported unsolved
};

// single v_max_f64 / v_min_f64 on the device (a compare + two v_cndmask otherwise)
MPC_HD double fmax2(double a, double b) { return __builtin_fmax(a, b); }
MPC_HD double fmin2(double a, double b) { return __builtin_fmin(a, b); }

// bounds of the reference NLP (agents/pure_mpc.py:272-280), relaxed like IPOPT's bound_relax_factor 1e-8.
// State bounds: index 0 = theta in [-pi, pi], 1 = v in [0, 30]  (|x|,|y| <= 500 can never be active here).
#define MPC_PI 3.14159265358979323846
MPC_HD double xlo_r(int i) { return i == 0 ? -MPC_PI - 1e-8 * MPC_PI : -1e-8; }
MPC_HD double xhi_r(int i) { return i == 0 ? MPC_PI + 1e-8 * MPC_PI : 30.0 + 30e-8; }
MPC_HD double ulo_r(int i) { return i == 0 ? -5.0 - 5e-8 : -(MPC_PI / 3.0) - 1e-8 * (MPC_PI / 3.0); }
MPC_HD double uhi_r(int i) { return i == 0 ? 5.0 + 5e-8 : (MPC_PI / 3.0) + 1e-8 * (MPC_PI / 3.0); }

constexpr double kInvWheelbase = 1.0 / 2.5;  // Vehicle.LENGTH, agents/utils.py:18

// warm start (opt-in, not in the reference): given initial controls are moved 0.1 % of the range inside their bounds
MPC_HD double warm_clamp(double u, int i) {
    const double lo = i == 0 ? -5.0 : -(MPC_PI / 3.0), hi = -lo, m = 1e-3 * (hi - lo);
    return fmin2(fmax2(u, lo + m), hi - m);
}

// ---- lean FP64 math -------------------------------------------------------------------------------
// The solver is bound by instruction issue / instruction-cache footprint, not by memory, so the generic
// libm expansions (large-argument trig reduction, IEEE division with denormal fix-ups) are replaced by
// bounded-range versions.  All arguments here are bounded by the NLP itself: |theta| <= pi(1+1e-8),
// |delta| <= pi/3(1+1e-8), slacks and determinants are positive normal numbers.
MPC_HD double frcp(double x) {  // 1/x for positive normal x
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rcp_f64 is good to 2^-24 (measured, tools/ubench/fp64_latency.hip): one third-order step gives 2^-72
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
#else
    return 1.0 / x;
#endif
}
MPC_HD double frsqrt(double x) {  // 1/sqrt(x) for positive normal x
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rsq_f64 is good to 2^-24 (measured): one third-order step gives 2^-72
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);     // 1 - x y^2
    return fma(y * e, fma(0.375, e, 0.5), y);
#else
    return 1.0 / sqrt(x);
#endif
}
// sin and cos of |x| <= ~2 pi: two-constant Cody-Waite reduction to |r| <= pi/4, fdlibm kernel polynomials
MPC_HD void sincos_b(double x, double &s, double &c) {
    const double n = rint(x * 6.36619772367581382433e-01);  // x * 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                              2.75573137070700676789e-06),
                                       -1.98412698298579493134e-04),
                                8.33333333332248946124e-03),
                         -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                              -2.75573143513906633035e-07),
                                       2.48015872894767294178e-05),
                                -1.38888888888741095749e-03),
                         4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = ((int)n) & 3;
    const double sa = (q & 1) ? cr : sr, ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

// The twelve coefficients of the fdlibm sine / cosine kernels (valid on |x| <= pi/4).  The solver keeps them in a table
// in LDS and loads them at the start of every phase that evaluates the dynamics: as 64-bit literals the compiler hoists
// them out of the iteration loop into 24 vector registers for the whole solve - a seventh of the 128 registers a wave
// may use if all 4096 waves of a batch are to be resident at once.
struct TrigCoef {
    double s[6];   // s[0] z^5 + ... : sin r = r + z r (s5 + z (s4 + ...)), listed from the highest power
    double c[6];
};
constexpr int kTrigWords = 12;
MPC_HD double trig_coef(int i) {
    constexpr double t[kTrigWords] = {1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,
                                      -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01,
                                      -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,
                                      2.48015872894767294178e-05,  -1.38888888888741095749e-03, 4.16666666666666019037e-02};
    return t[i];
}

// sin and cos of the two angles of the bicycle model, the steering angle |delta| <= pi/3 (1 + 1e-8) and the heading
// |theta| <= pi (1 + 1e-8) (the bounds of the NLP), without range reduction or quadrant logic: the fdlibm kernel
// polynomials (valid on |x| <= pi/4) are evaluated at delta/2 and theta/4 - four independent Horner chains that share
// every coefficient, so a lone wave overlaps their latencies - followed by one resp. two angle doublings
// (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a; absolute error < 1e-15).
// Measured on MI355X: the generic two-argument version (reduction, quadrant selects on 64-bit values) was ~190
// instructions of the ~330 of a rollout stage, this one is ~50.
MPC_HD void sincos_delta_theta(const TrigCoef &K, double delta, double theta, double &sd, double &cd, double &st, double &ct) {
    const double rx = 0.5 * delta, ry = 0.25 * theta;
    const double zx = rx * rx, zy = ry * ry;
    double psx = fma(zx, K.s[0], K.s[1]);
    double psy = fma(zy, K.s[0], K.s[1]);
    double pcx = fma(zx, K.c[0], K.c[1]);
    double pcy = fma(zy, K.c[0], K.c[1]);
#pragma unroll
    for (int i = 2; i < 6; ++i) {
        psx = fma(zx, psx, K.s[i]);
        psy = fma(zy, psy, K.s[i]);
        pcx = fma(zx, pcx, K.c[i]);
        pcy = fma(zy, pcy, K.c[i]);
    }
    const double sx = fma(zx * rx, psx, rx), sy = fma(zy * ry, psy, ry);
    const double cx = fma(zx * zx, pcx, fma(-0.5, zx, 1.0)), cy = fma(zy * zy, pcy, fma(-0.5, zy, 1.0));
    sd = 2.0 * sx * cx;
    cd = fma(-2.0 * sx, sx, 1.0);
    const double s2 = 2.0 * sy * cy, c2 = fma(-2.0 * sy, sy, 1.0);     // theta / 2
    st = 2.0 * s2 * c2;
    ct = fma(-2.0 * s2, s2, 1.0);
}

// sin and cos of |x| <= pi/2 (1 + tiny) without range reduction: the fdlibm kernel polynomials at x/2 and one angle
// doubling (absolute error < 5e-16)
MPC_HD void sincos_half(const TrigCoef &K, double x, double &s, double &c) {
    const double r = 0.5 * x, z = r * r;
    double ps = fma(z, K.s[0], K.s[1]);
    double pc = fma(z, K.c[0], K.c[1]);
#pragma unroll
    for (int i = 2; i < 6; ++i) {
        ps = fma(z, ps, K.s[i]);
        pc = fma(z, pc, K.c[i]);
    }
    const double sh = fma(z * r, ps, r), ch = fma(z * z, pc, fma(-0.5, z, 1.0));
    s = 2.0 * sh * ch;
    c = fma(-2.0 * sh, sh, 1.0);
}

// atan(t) for |t| <= ~4 (the steering angle whose tangent is t; used on the path that projects a rollout back into the
// theta bounds): rational first guess (error < 5e-3), then two Newton steps on sin(d) - t cos(d) = 0, whose update
// (sin d - t cos d) / (cos d + t sin d) = tan(d - atan t) converges cubically (5e-3 -> 4e-8 -> 2e-23)
MPC_HD double atan_b(const TrigCoef &K, double t) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double at = fabs(t);
    // |t| <= 1: t / (1 + 0.28125 t^2); above: pi/2 - the same in 1/t
    const double r = at <= 1.0 ? at : frcp(at);
    const double g = r * frcp(fma(0.28125 * r, r, 1.0));
    double d = at <= 1.0 ? g : 1.57079632679489655800e+00 - g;
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        double sd, cd;
        sincos_half(K, d, sd, cd);
        d -= fma(-at, cd, sd) * frcp(fma(at, sd, cd));
    }
    return t < 0.0 ? -d : d;
#else
    (void)K;
    return atan(t);
#endif
}

// log(x) for positive normal x with the coefficients passed in (the solver's LDS table, like TrigCoef): fdlibm's e_log
// (x = 2^k (1 + f), sqrt(1/2) <= 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2))), error < 1 ulp.
constexpr int kLogWords = 10;
MPC_HD double log_coef(int i) {
    constexpr double t[kLogWords] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
                                     2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
                                     1.479819860511658591e-01, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
                                     7.07106781186547524401e-01};
    return t[i];
}
MPC_HD double log_pos(const double *K, double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x);      // x = m 2^e, 1/2 <= m < 1
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lt = m < K[9];
    m = lt ? 2.0 * m : m;
    e = lt ? e - 1 : e;
    const double f = m - 1.0, s = f * frcp(2.0 + f), z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, K[5], K[3]), K[1]);
    const double t2 = z * fma(w, fma(w, fma(w, K[6], K[4]), K[2]), K[0]);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    return fma(dk, K[7], -((hfsq - fma(s, hfsq + R, dk * K[8])) - f));
#else
    (void)K;
    return log(x);
#endif
}

// kinematic bicycle model (agents/pure_mpc.py:220-228): beta = atan(LENGTH_REAR/LENGTH * tan(delta)).
// tan and atan are eliminated algebraically: with q = (4 cos^2 delta + sin^2 delta)^-1/2,
//   cos(beta) = 2 cos(delta) q,  sin(beta) = sin(delta) q,  sin/cos(theta+beta) by the addition theorems.
MPC_HD void dyn_eval(const TrigCoef &K, double theta, double delta, double &S, double &C, double &sb, double &cb) {
    double sd, cd, st, ct;
    sincos_delta_theta(K, delta, theta, sd, cd, st, ct);
    const double q = frsqrt(fma(3.0 * cd, cd, 1.0));
    cb = 2.0 * cd * q;
    sb = sd * q;
    S = fma(st, cb, ct * sb);
    C = fma(ct, cb, -st * sb);
}
// beta' = 2 q^2 and beta'' = 12 sin(delta) cos(delta) q^4 expressed through sin/cos(beta): q^2 = sb^2 + cb^2/4
MPC_HD void beta_derivs(double sb, double cb, double &bp, double &bpp) {
    const double q2 = fma(sb, sb, 0.25 * cb * cb);
    bp = 2.0 * q2;
    bpp = 6.0 * sb * cb * q2;
}

}  // namespace mpc

// mpc_preamble.hpp - everything PureMPC_Agent.predict() does BEFORE the NLP solve, for one environment:
//   a2  observation parsing                     agents/base_agent.py:81-116   (Vehicle: agents/utils.py:16-40)
//   a4  path-crossing "collision" detector       agents/pure_mpc.py:552-676    with its 10-step memory
//       ego path prediction along the reference  agents/pure_mpc.py:459-527
//       constant-velocity prediction of others   agents/pure_mpc.py:529-550
//   a5  rewrite of the reference speed profile   agents/pure_mpc.py:678-724
//   a6  problem data of the solve                agents/pure_mpc.py:95-117
// written once for host and device (MPC_HD): the HIP kernel runs one thread per environment, the CPU test harness
// (tests/cpu_preamble_harness.cpp) loops over environments.  The per-environment state of the detector lives in
// `EnvState` records inside the engine handle.
//
// Arithmetic fidelity.  The reference runs on numpy 2.1 (requirements.txt), whose promotion rules keep float32
// scalars float32 when they meet Python floats, so parts of its preamble are float32 arithmetic on the float32
// observation: heading wrap, speed = |v|, the other vehicles' predicted polylines, the ego speed ramp and the first
// metres of its arc length, and np.linspace of the stop profile.  Those places use `float` with explicitly unfused
// operations here; everything else is double like the reference.  What the reference delegates to shapely
// (LineString.intersection, agents/pure_mpc.py:608-633) is the segment arithmetic of `first_crossing` below, the
// same construction as the host mirror `pure_mpc.first_path_crossing`.
#pragma once

#include <stdint.h>

#include "mpc_core.hpp"

namespace mpc {
namespace pre {

constexpr int kPredHorizon = 30;     // agents/pure_mpc.py:554
constexpr int kTimeThreshold = 30;   // agents/pure_mpc.py:555
constexpr int kMemorySteps = 10;     // agents/pure_mpc.py:39
constexpr int kSafetyBuffer = 5;     // agents/pure_mpc.py:681
constexpr double kMaxSpeed = 30.0;   // agents/pure_mpc.py:680
constexpr int kMaxOthers = 16;       // MPC_MAX_OTHERS
constexpr int kObsCols = 8;          // presence, x, y, vx, vy, heading, sin_h, cos_h  (config/config.py:13)
constexpr double kPi = 3.141592653589793;

// persistent detector state of one environment (agents/pure_mpc.py:38-43,63); all-zero = fresh episode
struct EnvState {
    int32_t collision_memory;
    int32_t has_memorized;    // memorized_conflict_indices is not None
    int32_t n_memorized;
    int32_t n_conflict;       // entries of conflict[] (= vehicles seen by the last full detection)
    int32_t is_collide;
    int32_t ego_index;
    int32_t stop_index1;      // stop_point as reference index + 1 (0 = None)
    int32_t last_valid_stop1; // last_valid_stop_point, same encoding
    int32_t conflict[kMaxOthers];   // conflict_index per vehicle, -1 = None
    int32_t memorized[kMaxOthers];
    double conflict_pt[kMaxOthers][2];    // conflict_points (agents/pure_mpc.py:590, 656): where the paths cross
    double memorized_pt[kMaxOthers][2];   // memorized_conflict_points (:40, 666)
};

// ---- operations that must round one by one like numpy's (no FMA contraction, also after inlining) -----------
// hipcc contracts a*b+c by default and HIP's __fmul_rn/__fadd_rn are plain operators, so the pragma is what counts;
// the host harness is additionally compiled with -ffp-contract=off.  Float division and sqrtf are correctly rounded
// by hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
MPC_HD float f32mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
MPC_HD float f32add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
MPC_HD double f64mul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
MPC_HD double f64add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
MPC_HD double f64sub(double a, double b) { return f64add(a, -b); }

struct P2 {
    double x, y;
};
MPC_HD double dist2d(double ax, double ay, double bx, double by) {
    const double dx = f64sub(ax, bx), dy = f64sub(ay, by);
    return sqrt(f64add(f64mul(dx, dx), f64mul(dy, dy)));
}

// reference table: [M][REF_COLS] (x, y, heading, sin, cos) followed by the speed column [M]
struct RefTable {
    const double *t;
    int M;
    MPC_HD double x(int i) const { return t[i * REF_COLS + R_X]; }
    MPC_HD double y(int i) const { return t[i * REF_COLS + R_Y]; }
    MPC_HD double v(int i) const { return t[M * REF_COLS + i]; }
    // argmin_i |ref_i - p| (first minimum), agents/pure_mpc.py:106-109, 567-570, 651-653
    MPC_HD int nearest(double px, double py) const {
        int best = 0;
        double bd = dist2d(x(0), y(0), px, py);
        for (int i = 1; i < M; ++i) {
            const double d = dist2d(x(i), y(i), px, py);
            if (d < bd) {
                bd = d;
                best = i;
            }
        }
        return best;
    }
};

// ---- a4: ego polyline, agents/pure_mpc.py:459-527 ------------------------------------------------------------
// Returns the number of points written to out (<= kPredHorizon + 1).
// start = R.nearest(px, py), which the caller has (it needs the same index for the reference speed and the ego index)
MPC_HD int ego_future(const RefTable &R, float px, float py, float speed, double reference_speed, double dt, P2 *out,
                      int start) {
    out[0] = P2{(double)px, (double)py};
    int n = 1;
    const int npts = R.M - start;
    if (npts < 2) return n;
    // current_speed / current_distance start as float32 (the observation's dtype) and become double once the ramp
    // reaches the float64 reference speed
    bool s32 = true, d32 = true;
    float cs_f = speed, cd_f = 0.0f;
    double cs_d = 0.0, cd_d = 0.0;
    const float acc_dt_f = (float)(3.5 * dt);   // Vehicle.max_acceleration * dt, weakly typed -> float32
    const float dt_f = (float)dt;
    // np.searchsorted(cumulative, distance) state: the travelled distance never decreases, so the search resumes
    // where the previous step stopped (the cumulative sums are formed in the same order as np.cumsum)
    int idx = 0;
    double cum = 0.0, prev = 0.0;
    for (int step = 0; step < kPredHorizon; ++step) {
        const double cs_now = s32 ? (double)cs_f : cs_d;
        if (cs_now < reference_speed) {
            if (s32) {
                const float t = f32add(cs_f, acc_dt_f);
                if (reference_speed < (double)t) {      // min(t, reference_speed) returns the float64 reference
                    s32 = false;
                    cs_d = reference_speed;
                } else {
                    cs_f = t;
                }
            } else {
                const double t = f64add(cs_d, 3.5 * dt);
                cs_d = reference_speed < t ? reference_speed : t;
            }
        } else {
            s32 = false;
            cs_d = reference_speed;
        }
        if (s32) {
            const float inc = f32mul(cs_f, dt_f);
            if (d32) cd_f = f32add(cd_f, inc);
            else cd_d = f64add(cd_d, (double)inc);
        } else {
            const double inc = f64mul(cs_d, dt);
            if (d32) {
                cd_d = f64add((double)cd_f, inc);
                d32 = false;
            } else {
                cd_d = f64add(cd_d, inc);
            }
        }
        const double cd = d32 ? (double)cd_f : cd_d;
        // next_idx = searchsorted(cumulative, cd) (left): first index with cumulative[idx] >= cd
        while (idx < npts && cum < cd) {
            ++idx;
            if (idx < npts) {
                prev = cum;
                cum = f64add(cum, dist2d(R.x(start + idx), R.y(start + idx), R.x(start + idx - 1), R.y(start + idx - 1)));
            }
        }
        if (idx >= npts) break;
        if (idx == 0) {
            out[n++] = P2{R.x(start), R.y(start)};
        } else {
            double alpha = (cum != prev) ? f64sub(cd, prev) / f64sub(cum, prev) : 1.0;
            alpha = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
            const double ax = R.x(start + idx - 1), ay = R.y(start + idx - 1);
            out[n++] = P2{f64add(ax, f64mul(alpha, f64sub(R.x(start + idx), ax))),
                          f64add(ay, f64mul(alpha, f64sub(R.y(start + idx), ay)))};
        }
    }
    if (n <= 1) {                                   // agents/pure_mpc.py:524-526
        for (int i = 0; i < kPredHorizon; ++i) out[i] = P2{(double)px, (double)py};
        return kPredHorizon;
    }
    return n;
}

// ---- segment intersection (what shapely does for the reference) ----------------------------------------------
struct Hit {
    double t;
    P2 p;
};
MPC_HD bool close2(P2 a, P2 b) {   // np.allclose(a, b, atol=1e-12) with the default rtol 1e-5
    return fabs(a.x - b.x) <= 1e-12 + 1e-5 * fabs(b.x) && fabs(a.y - b.y) <= 1e-12 + 1e-5 * fabs(b.y);
}
MPC_HD double clamp01(double t) { return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }
// hits of segment p-q with segment a-b: 0, 1 or 2 (collinear overlap ends); same arithmetic as
// pure_mpc._seg_intersections
MPC_HD int seg_intersections(P2 p, P2 q, P2 a, P2 b, Hit *h) {
    const double eps = 1e-12;
    const double rx = q.x - p.x, ry = q.y - p.y, sx = b.x - a.x, sy = b.y - a.y;
    const double rxs = f64sub(f64mul(rx, sy), f64mul(ry, sx));
    const double apx = a.x - p.x, apy = a.y - p.y;
    double m = 1.0;
    m = fmax(m, fmax(fabs(rx), fabs(ry)));
    m = fmax(m, fmax(fabs(sx), fabs(sy)));
    const double scale = m * m;
    if (fabs(rxs) > eps * scale) {
        const double t = f64sub(f64mul(apx, sy), f64mul(apy, sx)) / rxs;
        const double u = f64sub(f64mul(apx, ry), f64mul(apy, rx)) / rxs;
        if (-1e-12 <= t && t <= 1 + 1e-12 && -1e-12 <= u && u <= 1 + 1e-12) {
            const double tc = clamp01(t);
            h[0] = Hit{tc, P2{f64add(p.x, f64mul(tc, rx)), f64add(p.y, f64mul(tc, ry))}};
            return 1;
        }
        return 0;
    }
    if (fabs(f64sub(f64mul(apx, ry), f64mul(apy, rx))) > eps * scale) return 0;   // parallel, not collinear
    const double rr = f64add(f64mul(rx, rx), f64mul(ry, ry));
    if (rr == 0.0) {                                                             // p-q is a point
        const double ss = f64add(f64mul(sx, sx), f64mul(sy, sy));
        if (ss == 0.0) {
            // np.allclose(p, a): rtol 1e-5, atol 1e-8
            const bool same = fabs(p.x - a.x) <= 1e-8 + 1e-5 * fabs(a.x) && fabs(p.y - a.y) <= 1e-8 + 1e-5 * fabs(a.y);
            if (same) {
                h[0] = Hit{0.0, p};
                return 1;
            }
            return 0;
        }
        const double u = f64add(f64mul(p.x - a.x, sx), f64mul(p.y - a.y, sy)) / ss;
        if (-1e-12 <= u && u <= 1 + 1e-12) {
            h[0] = Hit{0.0, p};
            return 1;
        }
        return 0;
    }
    const double t0 = f64add(f64mul(apx, rx), f64mul(apy, ry)) / rr;
    const double t1 = f64add(f64mul(b.x - p.x, rx), f64mul(b.y - p.y, ry)) / rr;
    const double lo = fmax(0.0, fmin(t0, t1)), hi = fmin(1.0, fmax(t0, t1));
    if (lo > hi) return 0;
    h[0] = Hit{lo, P2{f64add(p.x, f64mul(lo, rx)), f64add(p.y, f64mul(lo, ry))}};
    if (hi - lo < 1e-15) return 1;
    h[1] = Hit{hi, P2{f64add(p.x, f64mul(hi, rx)), f64add(p.y, f64mul(hi, ry))}};
    return 2;
}

// The other vehicle's predicted path: 31 float32 points (the reference accumulates them in float32, agents/pure_mpc.py:
// 529-550), kept as float pairs - 248 bytes, in LDS on the device - and widened to double where they are used.
struct AgentPath {
    const float *xy;   // [n][2]
    int n;
    MPC_HD P2 at(int m) const { return P2{(double)xy[2 * m], (double)xy[2 * m + 1]}; }
};

// Intersections of the ego polyline with the (straight) agent polyline, as the candidate list the reference
// builds from shapely's result (agents/pure_mpc.py:615-633): every transversal crossing is a candidate, a collinear overlap
// contributes the middle vertex of the overlapping stretch.  ORDER: along the ego's direction of travel.  shapely / GEOS
// returns the members of a Multi* result in the iteration order of a hash map of its overlay graph, which cannot be
// reproduced (or relied on); with a single crossing - a straight agent path meets the straight-arc-straight ego path
// more than once only when it cuts the arc twice - there is nothing to order.  Returns the number of candidates (<= maxc).
//
// No per-lane arrays: the nodes of a collinear overlap (the overlap's ends on every ego segment it covers, and the
// agent's own vertices inside it) are needed sorted along the ego's direction with duplicates dropped, and of that list
// only the middle element.  Both families already come in that order - the ego-derived ones by construction, the agent's
// vertices by index, ascending or descending - so the list is walked as a two-way merge, twice: once to count, once to
// stop at the middle.  (Until round 3 the list was materialised and insertion-sorted in 1.5 KB of scratch per lane.)
constexpr int kMaxCross = 4;

// views of the data the crossing code walks over: plain arrays (the one-thread-per-environment form and the host harness) or
// LDS words of the wave-cooperative kernel (mpc_preamble_wave.hpp)
struct EgoPtr {
    const P2 *p;
    MPC_HD P2 at(int i) const { return p[i]; }
};
struct CandPtr {            // where the candidates go (indexed dynamically: LDS on the device, never a local array)
    P2 *p;
    MPC_HD void put(int q, P2 v) const { p[q] = v; }
    MPC_HD P2 get(int q) const { return p[q]; }
};
struct NoNodes {            // the one-thread form finds extent and middle node of a collinear stretch itself (OverlapWalk) ...
    static constexpr bool kCached = false;
    MPC_HD P2 far(int) const { return P2{0.0, 0.0}; }
    MPC_HD P2 mid(int) const { return P2{0.0, 0.0}; }
};

template <class EGO, class AG>
struct OverlapWalk {
    const EGO *ego;
    int i, jend;          // ego segments i .. jend - 1 overlap the agent's line
    P2 a, b;
    const AG *ag;
    double klo, khi, sx, sy, ss, dx, dy;
    bool asc;             // agent vertices by ascending index are ascending along the ego's direction
    P2 e_i;               // ego->at(i)
    // ego-derived node q: 0, 1 = the ends of the overlap on segment i; q >= 2 = the far end on segment i + q - 1
    MPC_HD P2 ego_node(int q) const {
        Hit h[2];
        if (q < 2) {
            seg_intersections(ego->at(i), ego->at(i + 1), a, b, h);
            return q == 0 ? h[0].p : h[1].p;
        }
        seg_intersections(ego->at(i + q - 1), ego->at(i + q), a, b, h);
        return h[1].p;
    }
    MPC_HD double key(P2 p) const { return f64add(f64mul(p.x - e_i.x, dx), f64mul(p.y - e_i.y, dy)); }
    MPC_HD bool agent_inside(int m) const {
        const P2 v = ag->at(m);
        const double kv = f64add(f64mul(v.x - a.x, sx), f64mul(v.y - a.y, sy));
        return ss > 0 && klo <= kv && kv <= khi;
    }
    // (key, x, y) order of the reference's sort
    MPC_HD static bool before(double ka, P2 pa, double kb, P2 pb) {
        return ka < kb || (ka == kb && (pa.x < pb.x || (pa.x == pb.x && pa.y < pb.y)));
    }
    // walks the merged, de-duplicated list; returns its length, and in `out` its element number `want` (if want >= 0)
    MPC_HD int walk(int want, P2 &out) const {
        const int ne_nodes = 2 + (jend - (i + 1));
        const int na = ag->n;
        int qe = 0;                                   // next ego-derived node
        int ma = asc ? 0 : na - 1;                    // next agent vertex (index), skipping those outside the overlap
        const int mstep = asc ? 1 : -1;
        while (ma >= 0 && ma < na && !agent_inside(ma)) ma += mstep;
        int nu = 0;
        P2 last{0.0, 0.0};
        for (;;) {
            const bool he = qe < ne_nodes, ha = ma >= 0 && ma < na;
            if (!he && !ha) break;
            P2 pe{0.0, 0.0}, pa{0.0, 0.0};
            double ke = 0.0, ka = 0.0;
            if (he) {
                pe = ego_node(qe);
                ke = key(pe);
            }
            if (ha) {
                pa = ag->at(ma);
                ka = key(pa);
            }
            const bool take_e = he && (!ha || !before(ka, pa, ke, pe));
            const P2 p = take_e ? pe : pa;
            if (take_e) {
                ++qe;
            } else {
                ma += mstep;
                while (ma >= 0 && ma < na && !agent_inside(ma)) ma += mstep;
            }
            if (nu == 0 || !close2(last, p)) {
                if (nu == want) {
                    out = p;
                    return nu + 1;      // (round 6) the second walk wants this element only: it need not go on to the end
                }
                last = p;
                ++nu;
            }
        }
        return nu;
    }
};

// `hits`: bit i set = ego segment i meets the agent's line at all (the wave kernel tests all (vehicle, segment) pairs in
// parallel first and the serial part below then visits only those); ~0 = test every segment here.  `twos` (read only with
// NODES::kCached, the wave kernel): bit i set = segment i OVERLAPS the line (two hits), nodes.far(i) holds the overlap's far end
// and nodes.mid(r) the middle node of the r-th collinear stretch (= r-th run of ones in `twos`), found by the whole wave before
// this is called (mpc_preamble_wave.hpp overlap_middle_wave).  Until round 6 the lane found extent and middle itself - up to 30
// intersection tests and two merges of up to 63 nodes, serial in one lane: ~50 k cycles per same-lane vehicle, the whole
// difference between a 13 us and a 90 us preamble.
template <class EGO, class AG, class CAND, class NODES>
MPC_HD int path_crossings_t(const EGO &ego, int ne, const AG &ag, const CAND &out, int maxc, unsigned hits, unsigned twos,
                            const NODES &nodes) {
    const int na = ag.n;
    if (ne < 2 || na < 2) return 0;
    const P2 a = ag.at(0), b = ag.at(na - 1);
    int nc = 0;
    bool last_is_point = false;      // what out[nc - 1] came from
    P2 line_end{0.0, 0.0};           // far end of the last collinear stretch
    int nrun = 0;                    // collinear stretches met so far
    for (int i = 0; i < ne - 1 && nc < maxc; ++i) {
        if (!((hits >> i) & 1u)) continue;
        Hit h[2];
        const int nh = seg_intersections(ego.at(i), ego.at(i + 1), a, b, h);
        if (nh == 0) continue;
        if (nh == 1) {
            // a crossing exactly at an ego vertex is found by both segments that share it
            bool dup = nc > 0 && last_is_point && close2(out.get(nc - 1), h[0].p);
            // the intersection is a point SET: where the ego joins or leaves the other path's line at a vertex, the segment
            // before / behind the collinear stretch touches the line in the stretch's end point, which is part of that piece
            if (!dup && nc > 0 && !last_is_point && close2(line_end, h[0].p)) dup = true;
            if (!dup && i + 2 < ne && (!NODES::kCached || ((twos >> (i + 1)) & 1u))) {
                Hit h2[2];
                dup = seg_intersections(ego.at(i + 1), ego.at(i + 2), a, b, h2) == 2 && close2(h2[0].p, h[0].p);
            }
            if (!dup) {
                out.put(nc++, h[0].p);
                last_is_point = true;
            }
            continue;
        }
        // collinear overlap starting on ego segment i: it goes on over the following segments that overlap too
        int j = i + 1;
        P2 pl = h[1].p;
        P2 mid{0.0, 0.0};
        if constexpr (NODES::kCached) {
            const unsigned rest = ~(twos >> j);                  // bit 0 = segment j; the mask has no bits at or above ne - 1
            j += __builtin_ctz(rest | 0x80000000u);
            if (j > i + 1) pl = nodes.far(j - 1);
            mid = nodes.mid(nrun++);
        } else {
            for (; j < ne - 1; ++j) {
                Hit h2[2];
                if (seg_intersections(ego.at(j), ego.at(j + 1), a, b, h2) != 2) break;
                pl = h2[1].p;
            }
            OverlapWalk<EGO, AG> w;
            w.ego = &ego;
            w.i = i;
            w.a = a;
            w.b = b;
            w.ag = &ag;
            w.e_i = ego.at(i);
            w.jend = j;
            w.sx = b.x - a.x;
            w.sy = b.y - a.y;
            w.ss = f64add(f64mul(w.sx, w.sx), f64mul(w.sy, w.sy));
            const P2 pf = h[0].p;
            const double k0 = f64add(f64mul(pf.x - a.x, w.sx), f64mul(pf.y - a.y, w.sy));
            const double k1 = f64add(f64mul(pl.x - a.x, w.sx), f64mul(pl.y - a.y, w.sy));
            w.klo = fmin(k0, k1) - 1e-12;
            w.khi = fmax(k0, k1) + 1e-12;
            const P2 e1 = ego.at(i + 1);
            w.dx = e1.x - w.e_i.x;
            w.dy = e1.y - w.e_i.y;
            w.asc = f64add(f64mul(w.sx, w.dx), f64mul(w.sy, w.dy)) >= 0.0;
            P2 dummy{0.0, 0.0};
            const int nu = w.walk(-1, dummy);
            w.walk(nu / 2, mid);
        }
        out.put(nc++, mid);
        last_is_point = false;
        line_end = pl;
        i = j - 1;      // go on behind the overlap
    }
    return nc;
}
MPC_HD int path_crossings(const P2 *ego, int ne, const AgentPath &ag, P2 *out, int maxc) {
    return path_crossings_t(EgoPtr{ego}, ne, ag, CandPtr{out}, maxc, ~0u, 0u, NoNodes{});
}
// the first candidate (the only one in all but double-crossing scenes)
MPC_HD bool first_crossing(const P2 *ego, int ne, const AgentPath &ag, P2 &out) {
    P2 c[1];
    if (path_crossings(ego, ne, ag, c, 1) == 0) return false;
    out = c[0];
    return true;
}

MPC_HD int argmin_dist(const P2 *pts, int n, P2 p) {
    int best = 0;
    double bd = dist2d(pts[0].x, pts[0].y, p.x, p.y);
    for (int i = 1; i < n; ++i) {
        const double d = dist2d(pts[i].x, pts[i].y, p.x, p.y);
        if (d < bd) {
            bd = d;
            best = i;
        }
    }
    return best;
}
MPC_HD int argmin_dist(const AgentPath &ag, P2 p) {
    int best = 0;
    double bd = dist2d(ag.at(0).x, ag.at(0).y, p.x, p.y);
    for (int i = 1; i < ag.n; ++i) {
        const P2 v = ag.at(i);
        const double d = dist2d(v.x, v.y, p.x, p.y);
        if (d < bd) {
            bd = d;
            best = i;
        }
    }
    return best;
}

// float32 |v| like np.linalg.norm of a float32 pair (agents/utils.py:36)
MPC_HD float speed_f32(float vx, float vy) { return sqrtf(f32add(f32mul(vx, vx), f32mul(vy, vy))); }

// heading wrap of the ego row (agents/base_agent.py:156-170) on a float32 scalar: the subtraction stays float32
// ... and so do the comparisons: numpy (NEP 50) converts the Python float np.pi to float32 before comparing it with a
// float32 scalar, so a heading of exactly float32(-pi) = -3.1415927 - below -pi as a double, and what an ego on the exit
// straight (table heading -pi) is observed with - is NOT wrapped (tests/golden/reference_random.npz, case 6)
MPC_HD float normalize_angle_f32(float a) {
    const float two_pi = (float)(2.0 * kPi), pi_f = (float)kPi;
    while (a > pi_f) a = f32add(a, -two_pi);
    while (a < -pi_f) a = f32add(a, two_pi);
    return a;
}

// ---- the preamble of one environment in three parts (the HIP kernel spreads part 2 over lanes) -------------------
struct Parsed {
    float ex, ey, eh, ev;   // ego x, y, wrapped heading, speed (float32 like the reference's Vehicle fields)
    int observed;           // other vehicles present (rows 1..observed)
};

// part 1 - a2: parse (agents/base_agent.py:81-116)
MPC_HD Parsed parse_obs(const float *obs, int rows) {
    int present = 0;
    for (int r = 0; r < rows; ++r) present += (obs[r * kObsCols + 0] == 1.0f) ? 1 : 0;
    int observed = present - 1;
    observed = observed < 0 ? 0 : observed;
    observed = observed > kMaxOthers ? kMaxOthers : observed;
    return Parsed{obs[1], obs[2], normalize_angle_f32(obs[5]), speed_f32(obs[3], obs[4]), observed};
}
// problem data that comes straight from the observation: state[4], others[observed][4] (x, y, speed, heading)
MPC_HD void write_vehicles(const float *obs, const Parsed &p, double *state, double *others) {
    state[0] = (double)p.ex;
    state[1] = (double)p.ey;
    state[2] = (double)p.eh;
    state[3] = (double)p.ev;
    for (int j = 0; j < p.observed; ++j) {
        const float *o = obs + (j + 1) * kObsCols;
        others[j * 4 + 0] = (double)o[1];
        others[j * 4 + 1] = (double)o[2];
        others[j * 4 + 2] = (double)speed_f32(o[3], o[4]);
        others[j * 4 + 3] = (double)o[5];             // not wrapped (agents/base_agent.py:112)
    }
}
// the detector replays its memory instead of looking at the scene (agents/pure_mpc.py:558-563)
MPC_HD bool replays_memory(const EnvState &st) { return st.collision_memory > 0 && st.has_memorized; }

// part 2 - a4 for ONE other vehicle (agents/pure_mpc.py:575-660): reference index of the conflict point or -1.
// o = its observation row; the constant-velocity polyline is float32 arithmetic (agents/pure_mpc.py:529-550:
// step = speed * dt * [cos h, sin h], positions accumulate).
// ag_xy (2 x 31 floats) and cand (kMaxCross points) are the caller's work space: LDS on the device, where a dynamically
// indexed local array would be scratch memory.
// predict_future_positions (agents/pure_mpc.py:529-550) for the observation row o: 31 float32 points into ag_xy
MPC_HD void agent_path(const float *o, double dt, float *ag_xy) {
    const float sp = speed_f32(o[3], o[4]);
    const float sdt = f32mul(sp, (float)dt);
    const float stx = f32mul(sdt, (float)cos((double)o[5])), sty = f32mul(sdt, (float)sin((double)o[5]));
    float ax = o[1], ay = o[2];
    ag_xy[0] = ax;
    ag_xy[1] = ay;
    for (int m = 1; m <= kPredHorizon; ++m) {
        ax = f32add(ax, stx);
        ay = f32add(ay, sty);
        ag_xy[2 * m] = ax;
        ag_xy[2 * m + 1] = ay;
    }
}

MPC_HD int detect_vehicle(const float *o, const P2 *ego, int ne, const RefTable &R, double dt, P2 &pt_out, float *ag_xy,
                          P2 *cand) {
    agent_path(o, dt, ag_xy);
    const AgentPath ag{ag_xy, kPredHorizon + 1};
    // candidate loop of agents/pure_mpc.py:635-654: the first intersection point whose ego / agent sample indices differ
    // by less than TIME_THRESHOLD decides (with 31 samples that is every point except the pairing 0 / 30)
    pt_out = P2{0.0, 0.0};
    const int nc = path_crossings(ego, ne, ag, cand, kMaxCross);
    for (int q = 0; q < nc; ++q) {
        const P2 pt = cand[q];
        const int ego_time = argmin_dist(ego, ne, pt);
        const int agent_time = argmin_dist(ag, pt);
        int dtm = ego_time - agent_time;
        dtm = dtm < 0 ? -dtm : dtm;
        if (dtm < kTimeThreshold) {
            pt_out = pt;
            return R.nearest(pt.x, pt.y);
        }
    }
    return -1;
}

// part 3 - detector state machine (agents/pure_mpc.py:558-563, 661-676), a6 ego index, a5 speed profile
// (agents/pure_mpc.py:678-724).  conflict: the part-2 results of vehicles 0..observed-1 (unused when replaying).
// advance = false: the state machine has already seen this observation (the caller ran _check_collision on its own,
// MPC_FLAG_DETECTED): only the ego index and the speed profile are derived, from the record as it stands.
MPC_HD void finish_env(const Parsed &p, const RefTable &R, int N, const double *ref_speed, const int32_t *conflict,
                       const P2 *conflict_pt, EnvState &st, int32_t &ego_index_out, double *vref, uint8_t &collide_out,
                       bool advance = true, int ego_nearest = -1) {
    if (!advance) {
        // nothing: st.is_collide, the conflict indices and the memory are those of the detection call
    } else if (replays_memory(st)) {
        st.n_conflict = st.n_memorized;
        for (int j = 0; j < kMaxOthers; ++j) {
            st.conflict[j] = st.memorized[j];
            st.conflict_pt[j][0] = st.memorized_pt[j][0];
            st.conflict_pt[j][1] = st.memorized_pt[j][1];
        }
        st.is_collide = 1;
        st.collision_memory -= 1;
    } else {
        bool any = false;
        st.n_conflict = p.observed;
        for (int j = 0; j < kMaxOthers; ++j) {
            st.conflict[j] = j < p.observed ? conflict[j] : -1;
            const bool hit = st.conflict[j] >= 0;
            st.conflict_pt[j][0] = hit ? conflict_pt[j].x : 0.0;
            st.conflict_pt[j][1] = hit ? conflict_pt[j].y : 0.0;
            any = any || hit;
        }
        st.is_collide = any ? 1 : 0;
        if (any) {
            st.collision_memory = kMemorySteps;
            st.has_memorized = 1;
            st.n_memorized = st.n_conflict;
            for (int j = 0; j < kMaxOthers; ++j) {
                st.memorized[j] = st.conflict[j];
                st.memorized_pt[j][0] = st.conflict_pt[j][0];
                st.memorized_pt[j][1] = st.conflict_pt[j][1];
            }
        } else if (st.collision_memory > 0) {
            st.collision_memory -= 1;
            st.is_collide = 1;
        } else {
            st.has_memorized = 0;
        }
    }

    // a6: ego_index is refreshed by _solve in every call (agents/pure_mpc.py:106-109)
    const float ev = p.ev;
    const int e = ego_nearest >= 0 ? ego_nearest : R.nearest((double)p.ex, (double)p.ey);
    st.ego_index = e;
    ego_index_out = e;
    collide_out = st.is_collide ? 1 : 0;

    // a5: speed profile over the horizon window
    const int M = R.M;
    if (ref_speed) {
        const double v = *ref_speed < 0.0 ? 0.0 : (*ref_speed > kMaxSpeed ? kMaxSpeed : *ref_speed);
        for (int k = 0; k <= N; ++k) vref[k] = v;
        return;
    }
    int stop = -1, pts = 0;
    if (st.is_collide) {
        const int32_t *ci = (st.collision_memory > 0 && st.has_memorized) ? st.memorized : st.conflict;
        const int nc = (st.collision_memory > 0 && st.has_memorized) ? st.n_memorized : st.n_conflict;
        int mn = -1;
        for (int j = 0; j < nc; ++j)
            if (ci[j] >= 0 && (mn < 0 || ci[j] < mn)) mn = ci[j];
        if (mn >= 0) {
            stop = mn - kSafetyBuffer;
            stop = stop < e + 1 ? e + 1 : stop;
            stop = stop > M - 1 ? M - 1 : stop;
            pts = stop - e;
            if (pts > 0) {
                st.stop_index1 = stop + 1;
                st.last_valid_stop1 = stop + 1;
            } else if (st.last_valid_stop1 > 0) {
                st.stop_index1 = st.last_valid_stop1;
            }
        }
    }
    // np.linspace(ego_speed, 0, pts) in float32 (start is a float32 scalar)
    const float fstep = pts > 1 ? (-ev) / (float)(pts - 1) : 0.0f;
    for (int k = 0; k <= N; ++k) {
        int idx = e + k;
        idx = idx > M - 1 ? M - 1 : idx;
        double v = R.v(idx);
        if (pts > 0) {
            if (idx >= stop) {
                v = 0.0;
            } else {
                const int i = idx - e;
                float y;
                if (pts == 1) y = ev;                               // div = 0: y = 0 * delta + start
                else if (i == pts - 1) y = 0.0f;                    // endpoint is set exactly
                else if (fstep == 0.0f) y = f32add(f32mul((float)i / (float)(pts - 1), -ev), ev);
                else y = f32add(f32mul((float)i, fstep), ev);
                v = (double)y;
            }
        }
        vref[k] = v;
    }
}

// ---- the three parts in sequence (host harness; the kernel calls the parts itself) -----------------------------
// obs: [rows][8] float32.  Outputs: state[4], vref[N+1], others[(rows-1)][4] (absent slots are not written), nveh,
// ego_index, is_collide.  ref_speed: RL override or nullptr.
MPC_HD void preamble_env(const float *obs, int rows, const RefTable &R, int N, double dt, const double *ref_speed,
                         EnvState &st, double *state, int32_t &ego_index_out, double *vref, uint8_t &collide_out,
                         double *others, int32_t &nveh_out, bool advance = true) {
    const Parsed p = parse_obs(obs, rows);
    write_vehicles(obs, p, state, others);
    nveh_out = p.observed;
    int32_t conflict[kMaxOthers];
    P2 cpt[kMaxOthers];
    const int e0 = R.nearest((double)p.ex, (double)p.ey);
    // An ego standing on the LAST reference point has a predicted path of one point (agents/pure_mpc.py:476-478): LineString
    // of a single point raises, _check_collision prints a warning and returns with the detector state untouched (:582-587)
    bool degenerate = false;
    if (advance && !replays_memory(st)) {
        P2 ego[kPredHorizon + 1];
        const int ne = ego_future(R, p.ex, p.ey, p.ev, R.v(e0), dt, ego, e0);
        degenerate = ne < 2;
        float ag_xy[2 * (kPredHorizon + 1)];
        P2 cand[kMaxCross];
        for (int j = 0; j < p.observed && !degenerate; ++j)
            conflict[j] = detect_vehicle(obs + (j + 1) * kObsCols, ego, ne, R, dt, cpt[j], ag_xy, cand);
    }
    finish_env(p, R, N, ref_speed, conflict, cpt, st, ego_index_out, vref, collide_out, advance && !degenerate, e0);
}

}  // namespace pre
}  // namespace mpc

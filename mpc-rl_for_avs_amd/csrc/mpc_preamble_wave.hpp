// mpc_preamble_wave.hpp - the observation preamble of mpc_preamble.hpp (a2 parse, a4 detector, a5 speed profile, a6 problem
// data; reference agents/base_agent.py:81-116, agents/pure_mpc.py:459-724) spread over the 64 lanes of ONE wave per
// environment.  Same arithmetic, statement for statement, as the one-thread form `preamble_env` (which stays the host
// harness's reference and the specification of every rounding): what changes is who computes what.
//
//   observation (base_agent.py:81-116)     every lane its own words, the presence count a ballot (round 6)
//   nearest reference point of the ego     64 lanes x M / 64 points, two wave reductions (value, then first index)
//   ego speed ramp / travelled distance    30 dependent steps of float32, then float64 additions, wave-uniform (they ARE serial): one
//                                          plain loop per phase of the ramp, the distance of step k kept in lane k (round 6)
//   arc length along the route             segment lengths for 128 points at once, kept in two registers per lane; the running sum
//                                          (np.cumsum order: sequential by definition) fetches them with v_readlane, lane i keeps
//                                          sum i, and it stops where the 3 s look-ahead ends (round 6)
//   the 30 predicted ego points            lane k = prediction step k: binary search in the running sums + interpolation
//   vehicle paths (30 float32 additions)   lane j = vehicle j
//   path x path crossing tests             all (vehicle, ego segment) pairs at once, 2 vehicles x 32 segments per pass; the
//                                          hit masks (wave ballots) tell the serial candidate logic which segments to visit, the
//                                          overlap masks where collinear stretches are, each overlap's far end stays in LDS
//   collinear stretches (:619-622)         the whole wave per stretch: overlap_middle_wave below (round 6)
//   candidate logic (agents/pure_mpc.py:615-633: points, stretches, order)             lane j, only on segments with a hit
//   candidate test (:635-654: nearest sample of either path, nearest reference point)  the whole wave per candidate
//   state machine, speed profile (:558-563, 661-724)    lane j = vehicle j's entries of the record, lane k = node k of the
//                                          profile, the record's scalars by lane 0 (round 6; finish_env stays the specification)
//
// Until round 3 the kernel ran 4 environments per wave with lane 0 of each 16-lane group doing the ego path, the state
// machine and the speed profile alone: 76 us per call at 256 environments, 303 us when every environment runs the full
// detection (profiles/r03_rollout_kernel_stats.csv), 256 VGPRs.  Rounds 4 - 5: one wave per environment, 32 - 36 us (120 us when
// every environment runs the full detection); round 6: 15.9 us (28 us), profiles/r06_rollout_kernel_stats.csv.
//
// CTX: the wave interface of mpc_wave.hpp (phase / ld / st / wave_min / ballot / wave_bcast): WaveOpsT on the device
// (mpc_wave_dev.hpp), HostCtx in tests/host_wave_ctx.hpp, where tests/cpu_preamble_harness.cpp runs this file on the CPU
// against `preamble_env`, the numpy mirror and the reference-generated fixtures.
#pragma once

#include "mpc_preamble.hpp"
#include "mpc_wave.hpp"

namespace mpc {
namespace pre {

constexpr int kWin = 128;      // reference points the arc-length walk stages at once (the table has 85)
// LDS words (doubles) of one environment
enum : int {
    PL_EGO = 0,                                   // 31 x (x, y)   predicted ego polyline
    PL_CD = PL_EGO + 2 * (kPredHorizon + 1),      // 31            travelled distance after step k (k = 1 .. 30)
    PL_SEG = PL_CD + kPredHorizon + 1,            // kWin          lengths of the route segments behind the start point
    PL_CUM = PL_SEG + kWin,                       // kWin + 1      their running sums (np.cumsum)
    PL_SORT = PL_SEG,                             // (once the ego points exist) 63 x (x, y): the nodes of one collinear stretch, sorted
    PL_MID = PL_CUM,                              // (the same)    16 x 4 x (x, y): middle node of each vehicle's r-th collinear stretch
    PL_AG = PL_CUM + kWin + 1,                    // 16 x 31 x (x, y)   vehicle paths (float32 values)
    PL_NODE = PL_AG + kMaxOthers * 2 * (kPredHorizon + 1),   // 16 x 31 x (x, y)   nodes of a collinear stretch (per vehicle)
    PL_CAND = PL_NODE + kMaxOthers * 2 * (kPredHorizon + 1), // 16 x 4 x (x, y)    crossing candidates
    PL_SIZE = PL_CAND + kMaxOthers * kMaxCross * 2
};
constexpr int preamble_wave_lds_doubles() { return PL_SIZE; }
static_assert(2 * (2 * (kPredHorizon + 1) + 1) <= kWin && kMaxOthers * kMaxCross * 2 <= kWin + 1, "PL_SORT / PL_MID reuse PL_SEG / PL_CUM");

template <class CTX>
struct LdsPts {             // n points (x, y) at consecutive LDS words
    const CTX *c;
    int base, n;
    MPC_HD P2 at(int i) const { return P2{c->ld(base + 2 * i), c->ld(base + 2 * i + 1)}; }
};
template <class CTX>
struct LdsSink {
    CTX *c;
    int base;
    MPC_HD void put(int q, P2 v) const {
        c->st(base + 2 * q, v.x);
        c->st(base + 2 * q + 1, v.y);
    }
    MPC_HD P2 get(int q) const { return P2{c->ld(base + 2 * q), c->ld(base + 2 * q + 1)}; }
};

// section ids for CTX::tick (cycle attribution in tools/ubench/preamble_sections.hip; a no-op in the product kernel and on the host)
enum : int { PT_PARSE = 0, PT_NEAREST, PT_RAMP, PT_ARC, PT_POINTS, PT_PATHS, PT_HITS, PT_STRETCHES, PT_CROSSINGS, PT_CANDIDATES,
             PT_EXPORT, PT_FINISH, PT_COUNT };

// what the wave has ready about one vehicle's collinear stretches when lane j walks over its hit segments (path_crossings_t)
template <class CTX>
struct StretchStore {
    static constexpr bool kCached = true;
    const CTX *c;
    int fars, mids;         // LDS words: far end of the overlap on ego segment i at fars + 2 i, middle node of stretch r at mids + 2 r
    MPC_HD P2 far(int i) const { return P2{c->ld(fars + 2 * i), c->ld(fars + 2 * i + 1)}; }
    MPC_HD P2 mid(int r) const { return P2{c->ld(mids + 2 * r), c->ld(mids + 2 * r + 1)}; }
};

// wave minimum of a per-lane value that is still needed afterwards (the host model of the reductions works in place)
template <class CTX>
MPC_HD double wmin(CTX &ctx, const wave::PerLane<double> &p) {
    wave::PerLane<double> t = p;
    return ctx.wave_min(t);
}
MPC_HD int popc64(unsigned long long v) { return __builtin_popcountll(v); }
MPC_HD int ctz64(unsigned long long v) { return __builtin_ctzll(v); }

// Middle node of the collinear stretch over ego segments i .. jend - 1 with the line of the vehicle whose path is at LDS word
// `agbase` (far ends of the overlaps at `fars`), by the whole wave: agents/pure_mpc.py:619-622 takes `coords[len(coords) // 2]` of
// the LineString shapely returns, whose coordinates are the overlap's ends on every ego segment and the vehicle's own vertices
// inside it, in order along the line, duplicates dropped (tests/host_preamble.py sorts; OverlapWalk in mpc_preamble.hpp, the
// one-thread form, merges the two families, which come sorted).  Here: lane q < 32 holds ego-derived node q (0 = the near end on
// segment i, q >= 1 the far end on segment i + q - 1), lane 32 + m the vehicle's vertex m if it lies inside; every node finds
// its rank in the (key, x, y) order by counting the nodes before it (n <= 63 steps by key alone, then the full comparison against
// the few nodes that share a key), the nodes go to LDS by rank, "close to the last node kept" (np.allclose: it depends on what was kept before) is decided from each node's distance to
// its one and two predecessors when no two nodes in a row are dropped - else by the serial rule over the sorted array - and
// the middle one of the kept is read back.  ~4 k cycles where one lane's two merges took ~50 k
// (tools/gpu_preamble_sections.py, profiles/r06_preamble_sections.txt).
template <class CTX>
MPC_HD P2 overlap_middle_wave(CTX &ctx, const LdsPts<CTX> &ego, int agbase, int fars, int i, int jend) {
    using wave::PerLane;
    const LdsPts<CTX> ag{&ctx, agbase, kPredHorizon + 1};
    const P2 a = ag.at(0), b = ag.at(kPredHorizon);
    const P2 e_i = ego.at(i), e1 = ego.at(i + 1);
    Hit h[2];
    seg_intersections(e_i, e1, a, b, h);                         // two hits: segment i starts the stretch
    const P2 pf = h[0].p;
    const P2 pl = jend > i + 1 ? P2{ctx.ld(fars + 2 * (jend - 1)), ctx.ld(fars + 2 * (jend - 1) + 1)} : h[1].p;
    const double sx = b.x - a.x, sy = b.y - a.y;
    const double ss = f64add(f64mul(sx, sx), f64mul(sy, sy));
    const double k0 = f64add(f64mul(pf.x - a.x, sx), f64mul(pf.y - a.y, sy));
    const double k1 = f64add(f64mul(pl.x - a.x, sx), f64mul(pl.y - a.y, sy));
    const double klo = fmin(k0, k1) - 1e-12, khi = fmax(k0, k1) + 1e-12;
    const double dx = e1.x - e_i.x, dy = e1.y - e_i.y;
    const int ne_nodes = jend - i + 1;
    PerLane<double> kk, px, py;
    PerLane<int> val, rank;
    ctx.phase([&](int lane) {
        P2 q{0.0, 0.0};
        int ok = 0;
        if (lane < 32) {
            if (lane < ne_nodes) {
                ok = 1;
                q = pf;
                if (lane >= 1) q = P2{ctx.ld(fars + 2 * (i + lane - 1)), ctx.ld(fars + 2 * (i + lane - 1) + 1)};
            }
        } else if (lane - 32 <= kPredHorizon) {
            q = ag.at(lane - 32);
            const double kv = f64add(f64mul(q.x - a.x, sx), f64mul(q.y - a.y, sy));
            ok = (ss > 0 && klo <= kv && kv <= khi) ? 1 : 0;
        }
        kk.at(lane) = f64add(f64mul(q.x - e_i.x, dx), f64mul(q.y - e_i.y, dy));
        px.at(lane) = q.x;
        py.at(lane) = q.y;
        val.at(lane) = ok;
        rank.at(lane) = 0;
    });
    const unsigned long long vm = ctx.ballot(val);
    const int n = popc64(vm);
    // rank = number of nodes before this one in the (key, x, y) order (equal nodes: by lane).  First by KEY alone over all nodes (two
    // lane reads, two compares per step), remembering who shares a key with someone; then the full comparison only against the
    // nodes that do (typically two: an end of the overlap and the vehicle's vertex on it)
    PerLane<int> tie;
    ctx.phase([&](int lane) { tie.at(lane) = 0; });
    for (unsigned long long t = vm; t; t &= t - 1) {
        const int y = ctz64(t);
        const double ky = ctx.lane_get(kk, y);
        ctx.phase([&](int lane) {
            const double kl = kk.at(lane);
            rank.at(lane) += ky < kl ? 1 : 0;
            tie.at(lane) |= (ky == kl && y != lane) ? 1 : 0;
        });
    }
    ctx.phase([&](int lane) { tie.at(lane) = (tie.at(lane) && val.at(lane)) ? 1 : 0; });
    for (unsigned long long t = ctx.ballot(tie); t; t &= t - 1) {
        const int y = ctz64(t);
        const double ky = ctx.lane_get(kk, y), xy = ctx.lane_get(px, y), yy = ctx.lane_get(py, y);
        ctx.phase([&](int lane) {
            const double kl = kk.at(lane), xl = px.at(lane), yl = py.at(lane);
            const bool bef = xy < xl || (xy == xl && yy < yl);                                 // OverlapWalk::before, keys equal
            const bool same = xy == xl && yy == yl;                                            // equal nodes: by lane
            rank.at(lane) += (ky == kl && (bef || (same && y < lane))) ? 1 : 0;
        });
    }
    ctx.phase([&](int lane) {
        if (val.at(lane)) {
            ctx.st(PL_SORT + 2 * rank.at(lane), px.at(lane));
            ctx.st(PL_SORT + 2 * rank.at(lane) + 1, py.at(lane));
        }
    });
    PerLane<int> c1, c2;
    ctx.phase([&](int lane) {
        int f1 = 0, f2 = 0;
        if (lane >= 1 && lane < n) {
            const P2 s0{ctx.ld(PL_SORT + 2 * lane), ctx.ld(PL_SORT + 2 * lane + 1)};
            f1 = close2(P2{ctx.ld(PL_SORT + 2 * lane - 2), ctx.ld(PL_SORT + 2 * lane - 1)}, s0) ? 1 : 0;
            if (lane >= 2) f2 = close2(P2{ctx.ld(PL_SORT + 2 * lane - 4), ctx.ld(PL_SORT + 2 * lane - 3)}, s0) ? 1 : 0;
        }
        c1.at(lane) = f1;
        c2.at(lane) = f2;
    });
    const unsigned long long C = ctx.ballot(c1), D = ctx.ballot(c2);
    unsigned long long keep;
    if ((C & (C >> 1)) == 0 && (D & (C << 1)) == 0) {
        // no two in a row close to their predecessor, and whoever follows a dropped node is not close to the node before that
        // one: the last node kept is always the predecessor or the one before it, and "kept" = "not close to the predecessor"
        keep = ((1ull << n) - 1) & ~C;
    } else {
        keep = 1ull;
        P2 last{ctx.ld(PL_SORT), ctx.ld(PL_SORT + 1)};
        for (int r = 1; r < n; ++r) {
            const P2 q{ctx.ld(PL_SORT + 2 * r), ctx.ld(PL_SORT + 2 * r + 1)};
            if (!close2(last, q)) {
                keep |= 1ull << r;
                last = q;
            }
        }
    }
    const int want = popc64(keep) / 2;
    for (int q = 0; q < want; ++q) keep &= keep - 1;
    const int r = ctz64(keep);
    return P2{ctx.ld(PL_SORT + 2 * r), ctx.ld(PL_SORT + 2 * r + 1)};
}

// what the diagnostics export wants to see (mpc_get_last_paths); nullptr = off
struct PreDiag {
    double *ego;      // [31][2]
    int32_t *len;     // [1]
    float *agents;    // [Vslots][31][2]
    int vslots;
};

// The preamble of ONE environment by one wave.  conf / cpt: 16 words each of workgroup memory for the detection results.
template <class CTX>
MPC_HD void preamble_env_wave(CTX &ctx, const float *obs, int rows, const RefTable &R, int N, double dt, const double *ref_speed,
                              EnvState &st, double *state, int32_t &ego_index_out, double *vref, uint8_t &collide_out,
                              double *others, int vslots, int32_t &nveh_out, bool advance, int32_t *conf, P2 *cpt,
                              const PreDiag &diag) {
    using wave::PerLane;
    const int M = R.M;
    // a2 (agents/base_agent.py:81-116): the observation is read ONCE, every lane its own words (round 6: parse_obs's ten dependent
    // wave-uniform loads cost 3.3 k cycles, `tools/gpu_preamble_sections.py`), the presence count is a ballot, the ego's fields
    // are handed out of lanes 1 .. 5
    Parsed p;
    {
        PerLane<int> w0, w1, pr0, pr1;
        const int nw = rows * kObsCols;                           // <= 17 * 8 = 136 words
        ctx.phase([&](int lane) {
            float a = 0.0f, b = 0.0f;
            if (lane < nw) a = obs[lane];
            if (lane + wave::kLanes < nw) b = obs[lane + wave::kLanes];
            int ia, ib;
            __builtin_memcpy(&ia, &a, 4);
            __builtin_memcpy(&ib, &b, 4);
            w0.at(lane) = ia;
            w1.at(lane) = ib;
            pr0.at(lane) = ((lane % kObsCols) == 0 && lane < nw && a == 1.0f) ? 1 : 0;
            pr1.at(lane) = ((lane % kObsCols) == 0 && lane + wave::kLanes < nw && b == 1.0f) ? 1 : 0;
        });
        int present = popc64(ctx.ballot(pr0));
        if (nw > wave::kLanes) present += popc64(ctx.ballot(pr1));
        if (nw > 2 * wave::kLanes)                                 // rows 16 (words 128 .. 135): beyond two words per lane
            for (int r = 2 * wave::kLanes / kObsCols; r < rows; ++r) present += (obs[r * kObsCols] == 1.0f) ? 1 : 0;
        int observed = present - 1;
        observed = observed < 0 ? 0 : observed;
        observed = observed > kMaxOthers ? kMaxOthers : observed;
        auto field = [&](int i) {
            const int bits = ctx.wave_bcast(w0, i);
            float f;
            __builtin_memcpy(&f, &bits, 4);
            return f;
        };
        p = Parsed{field(1), field(2), normalize_angle_f32(field(5)), speed_f32(field(3), field(4)), observed};
    }
    ctx.tick(PT_PARSE);
    const double ex = (double)p.ex, ey = (double)p.ey;

    // ---- nearest reference point of the ego (first minimum; agents/pure_mpc.py:106-109, 567-570, 471-474)
    PerLane<double> bd, bi;
    ctx.phase([&](int lane) {
        double d0 = INFINITY;
        int i0 = 0;                                               // like RefTable::nearest: 0 when nothing compares smaller
        for (int i = lane; i < M; i += wave::kLanes) {
            const double d = dist2d(R.x(i), R.y(i), ex, ey);
            if (d < d0) {
                d0 = d;
                i0 = i;
            }
        }
        bd.at(lane) = d0;
        bi.at(lane) = (double)i0;
    });
    const double dmin = wmin(ctx, bd);
    ctx.phase([&](int lane) { bi.at(lane) = bd.at(lane) == dmin ? bi.at(lane) : 1e9; });
    int e0 = (int)wmin(ctx, bi);
    e0 = (e0 < 0 || e0 >= M) ? 0 : e0;

    ctx.tick(PT_NEAREST);
    const bool replay = !advance || replays_memory(st);
    int ne = 0;
    bool degenerate = false;      // the ego's predicted path has a single point (it stands on the last reference point):
                                  // LineString() of one point raises, _check_collision returns early (agents/pure_mpc.py:582-587)
    if (!replay) {
        const int npts = M - e0;
        if (npts < 2) {
            ne = 1;
            degenerate = true;
            ctx.phase([&](int lane) {
                if (lane == 0) {
                    ctx.st(PL_EGO, ex);
                    ctx.st(PL_EGO + 1, ey);
                }
            });
        } else {
            // ---- speed ramp and travelled distance, agents/pure_mpc.py:489-499: float32 until the ramp reaches the float64
            //      reference speed (see ego_future); wave-uniform, every lane keeps the same values
            const double reference_speed = R.v(e0);
            float cs_f = p.ev, cd_f = 0.0f;
            const float acc_dt_f = (float)(3.5 * dt);
            const float dt_f = (float)dt;
            double cd_last = 0.0;
            PerLane<double> cdv;
            ctx.lanes([&](int lane) { cdv.at(lane) = 0.0; });
            // ego_future's state machine (mpc_preamble.hpp) has two phases and never goes back: (A) speed and distance float32,
            // the speed climbing by 3.5 dt per step while it stays below the reference speed; (B) from the step at which the
            // ramp reaches or passes it the speed IS the float64 reference and the distance a float64 that grows by the
            // constant reference_speed * dt (its first value: the float32 distance widened, plus that).  Round 6, second pass: as
            // two plain loops - 8 and 1 arithmetic instructions per step - instead of one loop that carried both continuations
            // and selected (45 instructions, 25 of them selects, on a lone wave's dependent-issue clock: 9.1 k cycles per call,
            // the largest section of a typical preamble; tools/gpu_preamble_sections.py).  Lane step + 1 keeps the step's
            // distance (one store per lane after the loops: 64 lanes writing ONE LDS word per step serialise on its bank).
            int step = 0;
#pragma unroll 1
            for (; step < kPredHorizon; ++step) {
                if (!((double)cs_f < reference_speed)) break;                    // at (or NaN against) the reference speed
                const float tf = f32add(cs_f, acc_dt_f);
                if (reference_speed < (double)tf) break;                        // min(t, reference) returns the float64 reference
                cs_f = tf;
                cd_f = f32add(cd_f, f32mul(cs_f, dt_f));
                cd_last = (double)cd_f;
                ctx.lanes([&](int lane) { cdv.at(lane) = lane == step + 1 ? cd_last : cdv.at(lane); });
            }
            if (step < kPredHorizon) {
                const double incd = f64mul(reference_speed, dt);
                cd_last = (double)cd_f;
#pragma unroll 1
                for (; step < kPredHorizon; ++step) {
                    cd_last = f64add(cd_last, incd);
                    ctx.lanes([&](int lane) { cdv.at(lane) = lane == step + 1 ? cd_last : cdv.at(lane); });
                }
            }
            ctx.phase([&](int lane) {
                if (lane >= 1 && lane <= kPredHorizon) ctx.st(PL_CD + lane, cdv.at(lane));
            });
            ctx.tick(PT_RAMP);
            // ---- route segments behind the start point, kWin at a time; running sums in np.cumsum's order.  Round 6: the segment
            //      lengths stay in registers (lane i: segments i and i + 64), the serial sum fetches them with v_readlane and
            //      lane i keeps sum i - until then every step was an LDS load the addition waited for plus a 64-lane store to
            //      one word (~110 cycles per step)
            const int nseg = npts - 1 < kWin ? npts - 1 : kWin;
            PerLane<double> sg0, sg1, cu0, cu1;
            ctx.phase([&](int lane) {
                const int i0 = lane, i1 = lane + wave::kLanes;
                sg0.at(lane) = i0 < nseg ? dist2d(R.x(e0 + i0 + 1), R.y(e0 + i0 + 1), R.x(e0 + i0), R.y(e0 + i0)) : 0.0;
                sg1.at(lane) = i1 < nseg ? dist2d(R.x(e0 + i1 + 1), R.y(e0 + i1 + 1), R.x(e0 + i1), R.y(e0 + i1)) : 0.0;
                cu0.at(lane) = 0.0;
                cu1.at(lane) = 0.0;
            });
            int ncum = 1;
            double cum_end = 0.0;
            {
                // segment i - 1 and running sum i sit in lanes i - 1 and i of the first register pair while they exist (i < 64),
                // then of the second; a loop per pair and the two seams (i = 64, i = kWin) apart keep each loop body at one
                // v_readlane pair, the addition, one select pair and the test (as ONE loop with the pair chosen per step the
                // compiler carried both pairs through every step: six register copies and four branches per step)
                double cum = 0.0;
                bool more = true;
                static_assert(kWin == 2 * wave::kLanes, "two running sums per lane; the last one (index kWin) apart");
                const int n0 = nseg < wave::kLanes - 1 ? nseg : wave::kLanes - 1;
                int i = 1;
#pragma unroll 1
                for (; i <= n0; ++i) {
                    cum = f64add(cum, ctx.lane_get(sg0, i - 1));
                    ctx.lanes([&](int lane) { cu0.at(lane) = lane == i ? cum : cu0.at(lane); });
                    if (!(cum < cd_last)) {                        // the look-ahead ends here: nothing beyond is searched
                        more = false;
                        ++i;
                        break;
                    }
                }
                if (more && i <= nseg) {                           // i = 64: the last segment of the first pair, the first sum of the second
                    cum = f64add(cum, ctx.lane_get(sg0, wave::kLanes - 1));
                    ctx.lanes([&](int lane) { cu1.at(lane) = lane == 0 ? cum : cu1.at(lane); });
                    more = cum < cd_last;
                    ++i;
                    const int n1 = nseg < kWin - 1 ? nseg : kWin - 1;
#pragma unroll 1
                    for (; more && i <= n1; ++i) {
                        cum = f64add(cum, ctx.lane_get(sg1, i - 1 - wave::kLanes));
                        ctx.lanes([&](int lane) { cu1.at(lane) = lane + wave::kLanes == i ? cum : cu1.at(lane); });
                        more = cum < cd_last;
                    }
                    if (more && i <= nseg) {                       // i = kWin: its sum has no lane of its own
                        cum = f64add(cum, ctx.lane_get(sg1, wave::kLanes - 1));
                        cum_end = cum;
                        ++i;
                    }
                }
                ncum = i;                                          // sums 0 .. i - 1 exist
            }
            ctx.phase([&](int lane) {
                if (lane < ncum) ctx.st(PL_CUM + lane, cu0.at(lane));                                     // (sum 0 = 0)
                if (lane + wave::kLanes < ncum && lane + wave::kLanes < kWin) ctx.st(PL_CUM + lane + wave::kLanes, cu1.at(lane));
                if (lane == 0 && ncum == kWin + 1) ctx.st(PL_CUM + kWin, cum_end);
            });
            ctx.tick(PT_ARC);
            // ---- the 30 predicted points: lane k = step k (agents/pure_mpc.py:501-521)
            PerLane<int> inside;
            ctx.phase([&](int lane) {
                int ok = 0;
                if (lane >= 1 && lane <= kPredHorizon) {
                    const double cd = ctx.ld(PL_CD + lane);
                    int idx;
                    double prev = 0.0, cum = 0.0;
                    if (!(ctx.ld(PL_CUM + ncum - 1) < cd)) {
                        // np.searchsorted(cumulative, cd): first index with cumulative >= cd
                        int lo = 0, hi = ncum - 1;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (ctx.ld(PL_CUM + mid) < cd) lo = mid + 1;
                            else hi = mid;
                        }
                        idx = lo;
                        cum = ctx.ld(PL_CUM + idx);
                        prev = idx > 0 ? ctx.ld(PL_CUM + idx - 1) : 0.0;
                    } else {
                        // beyond what was summed up (the staged window was too short, or the route ends): go on from there
                        idx = ncum - 1;
                        cum = ctx.ld(PL_CUM + idx);
                        prev = idx > 0 ? ctx.ld(PL_CUM + idx - 1) : 0.0;
                        while (idx < npts && cum < cd) {
                            ++idx;
                            if (idx < npts) {
                                prev = cum;
                                cum = f64add(cum, dist2d(R.x(e0 + idx), R.y(e0 + idx), R.x(e0 + idx - 1), R.y(e0 + idx - 1)));
                            }
                        }
                    }
                    if (idx < npts) {
                        ok = 1;
                        P2 q;
                        if (idx == 0) {
                            q = P2{R.x(e0), R.y(e0)};
                        } else {
                            double alpha = (cum != prev) ? f64sub(cd, prev) / f64sub(cum, prev) : 1.0;
                            alpha = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
                            const double ax = R.x(e0 + idx - 1), ay = R.y(e0 + idx - 1);
                            q = P2{f64add(ax, f64mul(alpha, f64sub(R.x(e0 + idx), ax))),
                                   f64add(ay, f64mul(alpha, f64sub(R.y(e0 + idx), ay)))};
                        }
                        ctx.st(PL_EGO + 2 * lane, q.x);
                        ctx.st(PL_EGO + 2 * lane + 1, q.y);
                    }
                } else if (lane == 0) {
                    ctx.st(PL_EGO, ex);
                    ctx.st(PL_EGO + 1, ey);
                }
                inside.at(lane) = ok;
            });
            // the loop of :501 ends at the first step beyond the route: the travelled distance never decreases, so the steps
            // that stay inside are a prefix
            ne = 1 + popc64(ctx.ballot(inside));
            if (ne <= 1) {                                         // agents/pure_mpc.py:524-526
                ne = kPredHorizon;
                ctx.phase([&](int lane) {
                    if (lane < kPredHorizon) {
                        ctx.st(PL_EGO + 2 * lane, ex);
                        ctx.st(PL_EGO + 2 * lane + 1, ey);
                    }
                });
            }
        }
    }

    ctx.tick(PT_POINTS);
    // ---- the other vehicles
    const int V = p.observed;
    if (!replay && !degenerate) {
        // constant-velocity paths, agents/pure_mpc.py:529-550: lane j = vehicle j
        ctx.phase([&](int lane) {
            if (lane < V) {
                const float *o = obs + (lane + 1) * kObsCols;
                const float sp = speed_f32(o[3], o[4]);
                const float sdt = f32mul(sp, (float)dt);
                const float stx = f32mul(sdt, (float)cos((double)o[5])), sty = f32mul(sdt, (float)sin((double)o[5]));
                float ax = o[1], ay = o[2];
                const int b = PL_AG + lane * 2 * (kPredHorizon + 1);
                ctx.st(b, (double)ax);
                ctx.st(b + 1, (double)ay);
                for (int m = 1; m <= kPredHorizon; ++m) {
                    ax = f32add(ax, stx);
                    ay = f32add(ay, sty);
                    ctx.st(b + 2 * m, (double)ax);
                    ctx.st(b + 2 * m + 1, (double)ay);
                }
                conf[lane] = -1;
                cpt[lane] = P2{0.0, 0.0};
            }
        });
        ctx.tick(PT_PATHS);
        // which ego segments meet which vehicle's line at all: 2 vehicles x 32 segment slots per pass
        const LdsPts<CTX> ego{&ctx, PL_EGO, ne};
        // (round 6) ... and which of them OVERLAP it (two hits), with the overlap's far end kept per (vehicle, segment): the
        // serial logic below then finds the extent of a collinear stretch from the mask instead of testing segment after segment
        PerLane<int> hits, twos, hit, two;
        ctx.phase([&](int lane) {
            hits.at(lane) = 0;
            twos.at(lane) = 0;
        });
        for (int r = 0; 2 * r < V; ++r) {
            ctx.phase([&](int lane) {
                const int i = lane & 31, j = 2 * r + (lane >> 5);
                int nh = 0;
                if (j < V && i < ne - 1) {
                    const LdsPts<CTX> ag{&ctx, PL_AG + j * 2 * (kPredHorizon + 1), kPredHorizon + 1};
                    Hit h[2];
                    nh = seg_intersections(ego.at(i), ego.at(i + 1), ag.at(0), ag.at(kPredHorizon), h);
                    if (nh == 2) {
                        ctx.st(PL_NODE + j * 2 * (kPredHorizon + 1) + 2 * i, h[1].p.x);
                        ctx.st(PL_NODE + j * 2 * (kPredHorizon + 1) + 2 * i + 1, h[1].p.y);
                    }
                }
                hit.at(lane) = nh > 0 ? 1 : 0;
                two.at(lane) = nh == 2 ? 1 : 0;
            });
            const unsigned long long bm = ctx.ballot(hit), bt = ctx.ballot(two);
            ctx.phase([&](int lane) {
                if ((lane >> 1) == r && lane < V) {
                    hits.at(lane) = (int)(unsigned)((lane & 1) ? (bm >> 32) : (bm & 0xffffffffull));
                    twos.at(lane) = (int)(unsigned)((lane & 1) ? (bt >> 32) : (bt & 0xffffffffull));
                }
            });
        }
        ctx.tick(PT_HITS);
        // collinear stretches (same-lane traffic): the middle node of each, by the whole wave, vehicle after vehicle
        for (unsigned long long ov = ctx.ballot(twos); ov; ov &= ov - 1) {
            const int j = ctz64(ov);
            unsigned tw = (unsigned)ctx.wave_bcast(twos, j);
            for (int r = 0; tw != 0 && r < kMaxCross; ++r) {         // a stretch is a candidate: the lane stops after kMaxCross
                const int i = __builtin_ctz(tw);
                const int len = __builtin_ctz(~(tw >> i) | 0x80000000u);
                const P2 m = overlap_middle_wave(ctx, ego, PL_AG + j * 2 * (kPredHorizon + 1), PL_NODE + j * 2 * (kPredHorizon + 1), i, i + len);
                ctx.phase([&](int lane) {
                    if (lane == 0) {
                        ctx.st(PL_MID + (j * kMaxCross + r) * 2, m.x);
                        ctx.st(PL_MID + (j * kMaxCross + r) * 2 + 1, m.y);
                    }
                });
                tw &= ~(((1u << len) - 1u) << i);
            }
        }
        ctx.tick(PT_STRETCHES);
        // candidates (agents/pure_mpc.py:615-633), lane j = vehicle j, only the segments with a hit are visited
        PerLane<int> ncand;
        ctx.phase([&](int lane) {
            int nc = 0;
            if (lane < V && hits.at(lane) != 0) {
                const LdsPts<CTX> ag{&ctx, PL_AG + lane * 2 * (kPredHorizon + 1), kPredHorizon + 1};
                const LdsSink<CTX> cand{&ctx, PL_CAND + lane * 2 * kMaxCross};
                const StretchStore<CTX> nodes{&ctx, PL_NODE + lane * 2 * (kPredHorizon + 1), PL_MID + lane * kMaxCross * 2};
                nc = path_crossings_t(ego, ne, ag, cand, kMaxCross, (unsigned)hits.at(lane), (unsigned)twos.at(lane), nodes);
            }
            ncand.at(lane) = nc;
        });
        ctx.tick(PT_CROSSINGS);
        // candidate loop of agents/pure_mpc.py:635-654, the whole wave per candidate: nearest sample of the ego's path (lanes
        // 0..30) and of the vehicle's (lanes 32..62); the first candidate whose sample indices differ by less than
        // TIME_THRESHOLD decides, and its nearest reference point is the conflict index
        unsigned long long todo = ctx.ballot(ncand);
        while (todo) {
            const int j = ctz64(todo);
            todo &= todo - 1;
            const int nc = ctx.wave_bcast(ncand, j);
            for (int q = 0; q < nc; ++q) {
                const P2 pt{ctx.ld(PL_CAND + j * 2 * kMaxCross + 2 * q), ctx.ld(PL_CAND + j * 2 * kMaxCross + 2 * q + 1)};
                PerLane<double> de, da, ie, ia;
                ctx.phase([&](int lane) {
                    double d1 = INFINITY, d2 = INFINITY;
                    if (lane < ne && lane <= kPredHorizon) d1 = dist2d(ctx.ld(PL_EGO + 2 * lane), ctx.ld(PL_EGO + 2 * lane + 1), pt.x, pt.y);
                    if (lane >= 32 && lane - 32 <= kPredHorizon) {
                        const int m = lane - 32, b = PL_AG + j * 2 * (kPredHorizon + 1);
                        d2 = dist2d(ctx.ld(b + 2 * m), ctx.ld(b + 2 * m + 1), pt.x, pt.y);
                    }
                    de.at(lane) = d1;
                    da.at(lane) = d2;
                });
                const double me = wmin(ctx, de), ma = wmin(ctx, da);
                ctx.phase([&](int lane) {
                    ie.at(lane) = (lane <= kPredHorizon && de.at(lane) == me) ? (double)lane : 1e9;
                    ia.at(lane) = (lane >= 32 && da.at(lane) == ma) ? (double)(lane - 32) : 1e9;
                });
                const int ego_time = (int)wmin(ctx, ie), agent_time = (int)wmin(ctx, ia);
                int dtm = ego_time - agent_time;
                dtm = dtm < 0 ? -dtm : dtm;
                if (dtm < kTimeThreshold) {
                    PerLane<double> rd, ri;
                    ctx.phase([&](int lane) {
                        double d0 = INFINITY;
                        int i0 = 0;
                        for (int i = lane; i < M; i += wave::kLanes) {
                            const double d = dist2d(R.x(i), R.y(i), pt.x, pt.y);
                            if (d < d0) {
                                d0 = d;
                                i0 = i;
                            }
                        }
                        rd.at(lane) = d0;
                        ri.at(lane) = (double)i0;
                    });
                    const double mr = wmin(ctx, rd);
                    ctx.phase([&](int lane) { ri.at(lane) = rd.at(lane) == mr ? ri.at(lane) : 1e9; });
                    int ci = (int)wmin(ctx, ri);
                    ci = (ci < 0 || ci >= M) ? 0 : ci;
                    ctx.phase([&](int lane) {
                        if (lane == 0) {
                            conf[j] = ci;
                            cpt[j] = pt;
                        }
                    });
                    break;
                }
            }
        }
    }

    ctx.tick(PT_CANDIDATES);
    // ---- diagnostics export (mpc_get_last_paths)
    if (diag.len) {
        const int nexp = replay ? 0 : ne;
        ctx.phase([&](int lane) {
            if (lane == 0) diag.len[0] = nexp;
            if (lane < nexp) {
                diag.ego[2 * lane] = ctx.ld(PL_EGO + 2 * lane);
                diag.ego[2 * lane + 1] = ctx.ld(PL_EGO + 2 * lane + 1);
            }
            if (!replay && !degenerate && lane < V && lane < diag.vslots)
                for (int m = 0; m < 2 * (kPredHorizon + 1); ++m)
                    diag.agents[lane * 2 * (kPredHorizon + 1) + m] = (float)ctx.ld(PL_AG + lane * 2 * (kPredHorizon + 1) + m);
        });
    }

    ctx.tick(PT_EXPORT);
    // ---- problem data that comes straight from the observation, state machine, ego index, speed profile.  Round 6: spread over
    //      the lanes (until round 5 lane 0 ran write_vehicles + finish_env alone: ~10 k cycles of dependent global-memory round
    //      trips on EVERY call, also when the detector only replays its memory - tools/gpu_preamble_sections.py).  Same
    //      statements as finish_env (mpc_preamble.hpp), which stays the specification: lane j owns vehicle j's entries of the
    //      record, lane k node k of the speed profile, the record's scalars are read once and written by lane 0.
    const bool adv = advance && !degenerate;
    const int32_t cm_old = st.collision_memory, hm_old = st.has_memorized, nmem_old = st.n_memorized, ncon_old = st.n_conflict;
    const int32_t col_old = st.is_collide, lvs_old = st.last_valid_stop1;
    const bool replay_mem = adv && cm_old > 0 && hm_old;          // replays_memory(st)
    PerLane<int> con, mem, hitl;
    ctx.phase([&](int lane) {
        int cj = -1, mj = -1, hj = 0;
        if (lane < kMaxOthers) {
            const int j = lane;
            if (!adv) {
                cj = st.conflict[j];
                mj = st.memorized[j];
            } else if (replay_mem) {
                mj = st.memorized[j];
                cj = mj;
                st.conflict[j] = cj;
                st.conflict_pt[j][0] = st.memorized_pt[j][0];
                st.conflict_pt[j][1] = st.memorized_pt[j][1];
            } else {
                cj = j < p.observed ? conf[j] : -1;
                hj = cj >= 0 ? 1 : 0;
                st.conflict[j] = cj;
                st.conflict_pt[j][0] = hj ? cpt[j].x : 0.0;
                st.conflict_pt[j][1] = hj ? cpt[j].y : 0.0;
                mj = st.memorized[j];
            }
        }
        con.at(lane) = cj;
        mem.at(lane) = mj;
        hitl.at(lane) = hj;
    });
    const bool any = adv && !replay_mem && ctx.ballot(hitl) != 0;
    // the record's scalars after the update
    int32_t cm = cm_old, hm = hm_old, nmem = nmem_old, ncon = ncon_old, col = col_old;
    if (adv) {
        if (replay_mem) {
            ncon = nmem_old;
            col = 1;
            cm = cm_old - 1;
        } else {
            ncon = p.observed;
            col = any ? 1 : 0;
            if (any) {
                cm = kMemorySteps;
                hm = 1;
                nmem = ncon;
            } else if (cm_old > 0) {
                cm = cm_old - 1;
                col = 1;
            } else {
                hm = 0;
            }
        }
    }
    if (any) {
        ctx.phase([&](int lane) {
            if (lane < kMaxOthers) {
                const int j = lane, cj = con.at(lane);
                st.memorized[j] = cj;
                st.memorized_pt[j][0] = cj >= 0 ? cpt[j].x : 0.0;
                st.memorized_pt[j][1] = cj >= 0 ? cpt[j].y : 0.0;
                mem.at(lane) = cj;
            }
        });
    }
    // a5: which node the ego stops at (the smallest conflict index of the list that counts - the memorised one while the
    // memory runs), finish_env's stop / pts
    const int e = e0;
    int stop = -1, pts = 0;
    int32_t stop1 = st.stop_index1, lvs = lvs_old;
    bool stop_dirty = false;
    if (!ref_speed && col) {
        const bool use_mem = cm > 0 && hm;
        const int nc = use_mem ? nmem : ncon;
        PerLane<double> cv;
        ctx.phase([&](int lane) {
            const int cj = use_mem ? mem.at(lane) : con.at(lane);
            cv.at(lane) = (lane < nc && lane < kMaxOthers && cj >= 0) ? (double)cj : 1e9;
        });
        const double mnd = ctx.wave_min(cv);
        if (mnd < 1e8) {
            const int mn = (int)mnd;
            stop = mn - kSafetyBuffer;
            stop = stop < e + 1 ? e + 1 : stop;
            stop = stop > M - 1 ? M - 1 : stop;
            pts = stop - e;
            if (pts > 0) {
                stop1 = stop + 1;
                lvs = stop + 1;
                stop_dirty = true;
            } else if (lvs_old > 0) {
                stop1 = lvs_old;
                stop_dirty = true;
            }
        }
    }
    const float ev = p.ev;
    const float fstep = pts > 1 ? (-ev) / (float)(pts - 1) : 0.0f;
    const double vover = ref_speed ? (*ref_speed < 0.0 ? 0.0 : (*ref_speed > kMaxSpeed ? kMaxSpeed : *ref_speed)) : 0.0;
    ctx.phase([&](int lane) {
        // ---- speed profile, lane k = node k (np.linspace(ego_speed, 0, pts) in float32, finish_env)
        for (int k = lane; k <= N; k += wave::kLanes) {
            double v;
            if (ref_speed) {
                v = vover;
            } else {
                int idx = e + k;
                idx = idx > M - 1 ? M - 1 : idx;
                v = R.v(idx);
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
            }
            vref[k] = v;
        }
        // ---- the other vehicles' problem data (write_vehicles), lane j = vehicle j; absent slots zero
        if (lane < vslots) {
            const int j = lane;
            double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
            if (j < p.observed) {
                const float *o = obs + (j + 1) * kObsCols;
                o0 = (double)o[1];
                o1 = (double)o[2];
                o2 = (double)speed_f32(o[3], o[4]);
                o3 = (double)o[5];             // not wrapped (agents/base_agent.py:112)
            }
            others[j * 4 + 0] = o0;
            others[j * 4 + 1] = o1;
            others[j * 4 + 2] = o2;
            others[j * 4 + 3] = o3;
        }
        if (lane == 0) {
            state[0] = (double)p.ex;
            state[1] = (double)p.ey;
            state[2] = (double)p.eh;
            state[3] = (double)p.ev;
            nveh_out = p.observed;
            if (adv) {
                st.collision_memory = cm;
                st.has_memorized = hm;
                st.n_memorized = nmem;
                st.n_conflict = ncon;
                st.is_collide = col;
            }
            st.ego_index = e;
            ego_index_out = e;
            collide_out = col ? 1 : 0;
            if (stop_dirty) {
                st.stop_index1 = stop1;
                st.last_valid_stop1 = lvs;
            }
        }
    });
    ctx.tick(PT_FINISH);
}

}  // namespace pre
}  // namespace mpc

// mpc_synth_env.hpp - one step of the synthetic intersection environment for ONE environment (SURVEY section 8 f-1:
// "env step as a HIP/torch kernel"), written once for device and host (MPC_HD): the kernel in mpc_engine.hip runs one
// thread per environment, tests/cpu_synth_env_harness.cpp loops over environments.
//
// It is the fused counterpart of rollout.SyntheticIntersectionEnv's torch implementation (step + observe + auto-reset +
// observe, ~100 small torch kernels per step) and follows it statement by statement:
//   * observation layout of the reference, config/config.py:10-26: 10 rows x [presence, x, y, vx, vy, heading, sin_h,
//     cos_h], absolute, float32, ego first, the others sorted by distance (`order: sorted`), absent rows zero;
//   * ego = the MPC's own vehicle model, agents/pure_mpc.py:220-228 (explicit Euler, beta = atan(tan(delta) / 2)), with
//     the action limits of config/config.py:29-31; the others drive with constant velocity on the four approach lanes;
//   * reward / termination SHAPE of envs/intersection_env_Feb2025_v1.py:80-155 (collision, speed, arrival, lane centring,
//     off-road; crash or arrival terminates, 200 steps truncate), spawn of envs/intersection_env_Feb2025_v1.py:397-410.
// Random draws (respawn of vehicles that left, reset of finished episodes) come from a counter-based generator keyed by
// (seed, environment, that environment's step counter, draw slot): no generator state besides the counter, so the step is
// replayable inside a captured hipGraph and independent of how environments are mapped to threads.
#pragma once

#include <stdint.h>

#include "mpc_core.hpp"

namespace mpc {
namespace env {

constexpr int kRows = 10;          // config/cfg.yaml:2 vehicles_count
constexpr int kCols = 8;
constexpr int kMaxOthers = kRows - 1;
constexpr int kEpisodeSteps = 200; // duration 10 s + 10 s at 10 Hz
constexpr double kLaneHalfWidth = 2.0, kCrashDistance = 2.5, kWheelbase = 2.5;
constexpr double kRewardCollision = -200.0, kRewardHighSpeed = 15.0, kRewardArrived = 50.0, kRewardCenter = 5.0,
                 kRewardOffRoad = -50.0;
constexpr double kPiE = 3.14159265358979323846;

// ---- counter-based random numbers: splitmix64 finaliser over (seed, env, counter, slot) -----------------------------
MPC_HD uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t key;   // mix of seed, environment and step counter
    MPC_HD Rng(uint64_t seed, int env, int64_t ctr) : key(mix64(mix64(seed ^ 0xA5A5A5A5ull) + (uint64_t)env * 0x100000001B3ull) ^ mix64((uint64_t)ctr)) {}
    MPC_HD uint64_t bits(int slot) const { return mix64(key + (uint64_t)slot * 0xD1342543DE82EF95ull); }
    MPC_HD double u01(int slot) const { return (double)(bits(slot) >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
    MPC_HD double normal(int slot) const {          // Box-Muller on two slots
        const double u1 = 1.0 - u01(slot), u2 = u01(slot + 1);
        return sqrt(-2.0 * log(u1)) * cos(2.0 * kPiE * u2);
    }
};

// one environment's state, as views into the batched arrays
struct View {
    double *ego;       // [4] x, y, heading, speed
    double *opos;      // [K][2]
    double *ospeed;    // [K]
    double *ohead;     // [K]
    uint8_t *oactive;  // [K]
    int32_t *t;
    int64_t *ctr;
};

// vehicle on one of the four approach lanes, driving towards the centre (right-hand traffic, lane offset 2 m); slots
// s .. s + 3 of the generator
MPC_HD void spawn_other(const Rng &r, int s, double dlo, double dhi, double &x, double &y, double &sp, double &h) {
    int lane = (int)(r.u01(s) * 4.0);
    lane = lane > 3 ? 3 : lane;
    h = lane == 0 ? 0.0 : (lane == 1 ? kPiE / 2 : (lane == 2 ? kPiE : -kPiE / 2));
    const double d = dlo + (dhi - dlo) * r.u01(s + 1);
    x = -d * cos(h) + (lane == 1 ? -2.0 : 0.0) + (lane == 3 ? 2.0 : 0.0);
    y = -d * sin(h) + (lane == 0 ? 2.0 : 0.0) + (lane == 2 ? -2.0 : 0.0);
    const double n = r.normal(s + 2);
    sp = 8.0 + n;
    sp = sp < 0.0 ? 0.0 : sp;
}

// draw slots of one step: 5 j .. 5 j + 4 respawn decision and vehicle j (j < 9), 64 ego spawn, 80 + 4 j .. reset vehicle j
constexpr int kSlotRespawn = 0, kSlotEgo = 64, kSlotReset = 80;

MPC_HD void reset_env(const View &v, int K, const Rng &r) {
    v.ego[0] = 2.0;
    v.ego[1] = 45.0 + (-5.0 + 10.0 * r.u01(kSlotEgo));   // envs/intersection_env_Feb2025_v1.py:397-410
    v.ego[2] = -kPiE / 2;
    v.ego[3] = 10.0;
    for (int j = 0; j < K; ++j) {
        double x, y, sp, h;
        spawn_other(r, kSlotReset + 4 * j, 5.0, 60.0, x, y, sp, h);
        v.opos[2 * j] = x;
        v.opos[2 * j + 1] = y;
        v.ospeed[j] = sp;
        v.ohead[j] = h;
        v.oactive[j] = 1;
    }
    *v.t = 0;
}

// observation of the reference (config/config.py:10-26), float32, others sorted by distance (stable), absent rows zero
MPC_HD void observe(const View &v, int K, float *obs) {
    for (int i = 0; i < kRows * kCols; ++i) obs[i] = 0.0f;
    const double x = v.ego[0], y = v.ego[1], th = v.ego[2], sp = v.ego[3];
    const double s = sin(th), c = cos(th);
    obs[0] = 1.0f;
    obs[1] = (float)x;
    obs[2] = (float)y;
    obs[3] = (float)(sp * c);
    obs[4] = (float)(sp * s);
    obs[5] = (float)th;
    obs[6] = (float)s;
    obs[7] = (float)c;
    int order[kMaxOthers];
    double dist[kMaxOthers];
    int n = 0;
    for (int j = 0; j < K; ++j) {
        if (!v.oactive[j]) continue;
        const double dx = v.opos[2 * j] - x, dy = v.opos[2 * j + 1] - y;
        const double d = sqrt(dx * dx + dy * dy);
        int p = n++;
        while (p > 0 && dist[p - 1] > d) {       // insertion sort, stable: equal distances keep their slot order
            dist[p] = dist[p - 1];
            order[p] = order[p - 1];
            --p;
        }
        dist[p] = d;
        order[p] = j;
    }
    for (int q = 0; q < n; ++q) {
        const int j = order[q];
        float *row = obs + (1 + q) * kCols;
        const double hh = v.ohead[j], sh = sin(hh), ch = cos(hh);
        row[0] = 1.0f;
        row[1] = (float)v.opos[2 * j];
        row[2] = (float)v.opos[2 * j + 1];
        row[3] = (float)(v.ospeed[j] * ch);
        row[4] = (float)(v.ospeed[j] * sh);
        row[5] = (float)hh;
        row[6] = (float)sh;
        row[7] = (float)ch;
    }
}

struct StepOut {
    float reward;
    uint8_t done, truncated, crashed, arrived;
};

// one policy step: action = (acceleration m/s^2, steering angle rad) as the RL wrappers hand it to env.step
// (agents/ppo_mpc.py:430-432).  Writes the terminal observation, auto-resets a finished episode and writes the observation
// the policy sees next.
MPC_HD StepOut step_env(const View &v, int K, double dt, double spawn_probability, uint64_t seed, int env_id,
                        const double *ref_xy, int M, const double *action, float *terminal_obs, float *obs) {
    const Rng r(seed, env_id, *v.ctr);
    *v.ctr += 1;
    double a = action[0], delta = action[1];
    a = a < -5.0 ? -5.0 : (a > 5.0 ? 5.0 : a);                                              // config/config.py:31
    delta = delta < -kPiE / 4 ? -kPiE / 4 : (delta > kPiE / 4 ? kPiE / 4 : delta);          // config/config.py:30
    const double x = v.ego[0], y = v.ego[1], th = v.ego[2], sp = v.ego[3];
    const double beta = atan(0.5 * tan(delta));
    v.ego[0] = x + sp * cos(th + beta) * dt;
    v.ego[1] = y + sp * sin(th + beta) * dt;
    v.ego[2] = th + sp / kWheelbase * sin(beta) * dt;
    double nv = sp + a * dt;
    v.ego[3] = nv < 0.0 ? 0.0 : (nv > 30.0 ? 30.0 : nv);
    bool crashed = false;
    for (int j = 0; j < K; ++j) {
        v.opos[2 * j] += v.ospeed[j] * dt * cos(v.ohead[j]);
        v.opos[2 * j + 1] += v.ospeed[j] * dt * sin(v.ohead[j]);
        const double ax = fabs(v.opos[2 * j]), ay = fabs(v.opos[2 * j + 1]);
        const bool gone = (ax > ay ? ax : ay) > 65.0 || !v.oactive[j];
        if (gone) {
            const bool respawn = r.u01(kSlotRespawn + 5 * j) < spawn_probability;
            if (respawn) {
                double px, py, ps, ph;
                spawn_other(r, kSlotRespawn + 5 * j + 1, 40.0, 60.0, px, py, ps, ph);
                v.opos[2 * j] = px;
                v.opos[2 * j + 1] = py;
                v.ospeed[j] = ps;
                v.ohead[j] = ph;
            }
            v.oactive[j] = respawn ? 1 : 0;
        }
        if (v.oactive[j]) {
            const double dx = v.opos[2 * j] - v.ego[0], dy = v.opos[2 * j + 1] - v.ego[1];
            crashed = crashed || sqrt(dx * dx + dy * dy) < kCrashDistance;
        }
    }
    double lateral = INFINITY;
    int idx = 0;
    for (int i = 0; i < M; ++i) {
        const double dx = ref_xy[2 * i] - v.ego[0], dy = ref_xy[2 * i + 1] - v.ego[1];
        const double d = sqrt(dx * dx + dy * dy);
        if (d < lateral) {
            lateral = d;
            idx = i;
        }
    }
    const bool on_road = lateral <= kLaneHalfWidth;
    const bool arrived = idx >= M - 3 && on_road;
    double cen = lateral / kLaneHalfWidth;
    cen = 1.0 - (cen > 1.0 ? 1.0 : cen);
    const double reward = kRewardCollision * (crashed ? 1.0 : 0.0) + kRewardHighSpeed * (v.ego[3] / 10.0) +
                          kRewardArrived * (arrived ? 1.0 : 0.0) + (on_road ? kRewardCenter * cen : kRewardOffRoad);
    *v.t += 1;
    const bool terminated = crashed || arrived;
    const bool truncated = *v.t >= kEpisodeSteps && !terminated;
    const bool done = terminated || truncated;
    observe(v, K, terminal_obs);
    if (done) {
        reset_env(v, K, r);
        observe(v, K, obs);
    } else {
        for (int i = 0; i < kRows * kCols; ++i) obs[i] = terminal_obs[i];
    }
    StepOut o;
    o.reward = (float)reward;
    o.done = done;
    o.truncated = truncated;
    o.crashed = crashed;
    o.arrived = arrived;
    return o;
}

}  // namespace env
}  // namespace mpc

// mpc_rollout_glue.hpp - the two pieces of a rollout step (BASELINE configs 4-5) that are neither the MPC nor the environment:
//
//   policy_act      the SB3 `MlpPolicy` shaped actor-critic the reference's PPO_MPC / A2C_MPC build (trainers/trainer.py:375-449,
//                   net_arch default [64, 64], tanh; obs 10 x 8 = 80 inputs; agents/ppo_mpc.py:390-394 `self.policy(obs_tensor)`):
//                   both towers as ONE 80 -> 2H -> 2H -> (A + 1) network (first layers side by side, second layer block
//                   diagonal, both heads in one matrix - the layout ActorCritic.refresh_fused() keeps), the Gaussian sample
//                   mean + std * noise, its log-probability, the clip to the Box(-1, 1) action space (:399-407) and the mapping
//                   of the action onto the MPC's inputs (v0 :410-414 reference speed, v1 :416-420 cost weights).
//   rollout_record  the buffer row of the step (rollout_buffer.add, :462-469), the carry-over of observation and episode
//                   starts to the next step, and the episode counters.
//   rollout_finish  what ends a rollout (:471-476 `rollout_buffer.compute_returns_and_advantage`, SB3's RolloutBuffer used as
//                   is by the reference): the truncation bootstrap rewards += gamma V(terminal observation) (:451-461, here
//                   for the whole rollout at once) and generalised advantage estimation, a reversed recurrence over the
//                   rollout's T steps per environment.  In torch: 11 launches per step of the rollout, issued from a Python
//                   loop after the last step - ~3 ms per 64-step rollout, 6 % of a config-4 rollout.  Here: one launch.
//
// In torch these were ~40 small launches per step (3 GEMMs through hipBLASLt at 6 - 20 us each for 10 Mflop, 2 tanh, the
// sample, clamp, casts, one concatenation, index copies, a stack and a sum): ~200 us of a 0.95 ms config-4 step at 256
// environments (profiles/r03_rollout_kernel_stats.csv).  Here: two launches.  One workgroup of 2H = 128 threads per
// environment, thread j = hidden unit j; every output accumulates its inputs in index order with fused multiply-adds, so the
// host build (tests/cpu_rollout_glue_harness.cpp) and the device agree to the rounding of tanhf.
#pragma once

#include <stdint.h>

#include "mpc_core.hpp"
#include "mpc_synth_env.hpp"      // the counter-based generator (mpc::env::Rng)

namespace mpc {
namespace glue {

constexpr int kObsDim = 80;      // VEHICLES_COUNT x 8 (config/config.py:10-26)
constexpr int kMaxHidden2 = 256; // 2 H
constexpr int kMaxAction = 8;

struct PolicyWeights {           // device pointers, float32, the layout of ActorCritic._fz
    const float *w1, *b1;        // [80][2H], [2H]
    const float *w2, *b2;        // [2H][2H] block diagonal, [2H]
    const float *wh, *bh;        // [2H][A + 1], [A + 1]
    const float *std;            // [A]   exp(log_std)
    const float *c0;             // [1]   sum(log_std) + A / 2 log(2 pi)
};

// The Gaussian sample's standard-normal draw `a` of environment `env` (its GLOBAL id: the shards of one job draw distinct
// streams) at policy step `step` (a device counter mpc_rollout_record advances): counter-based like the environment's draws,
// so a step needs no generator state and replays inside a captured hipGraph.  (torch's generator costs three launches per
// step there - the draw and two updates of its graph-safe offset, 14 us of a 770 us config-4 step.)
MPC_HD float policy_noise(uint64_t seed, int env, long long step, int a) {
    const env::Rng r(seed ^ 0x706F6C696379ull, env, step);
    return (float)r.normal(2 * a);
}

// hidden unit j of layer 1 / 2 for one environment (x: 80 inputs; h1: 2H values)
MPC_HD float layer1_unit(const PolicyWeights &W, int H2, const float *x, int j) {
    float s = W.b1[j];
#pragma unroll 16      // the weight loads do not depend on the sum: sixteen of them in flight per thread
    for (int i = 0; i < kObsDim; ++i) s = fmaf(x[i], W.w1[i * H2 + j], s);
    return tanhf(s);
}
MPC_HD float layer2_unit(const PolicyWeights &W, int H2, const float *h1, int j) {
    // block diagonal: unit j of the policy tower (j < H) reads h1[0 .. H), of the value tower h1[H .. 2H)
    const int H = H2 / 2, lo = j < H ? 0 : H;
    float s = W.b2[j];
#pragma unroll 16
    for (int i = lo; i < lo + H; ++i) s = fmaf(h1[i], W.w2[i * H2 + j], s);
    return tanhf(s);
}
// head o (o < A: action mean o from the policy tower; o == A: the value from the value tower)
MPC_HD float head_unit(const PolicyWeights &W, int H2, int A, const float *h2, int o) {
    const int H = H2 / 2, lo = o < A ? 0 : H;
    float s = W.bh[o];
#pragma unroll 16
    for (int i = lo; i < lo + H; ++i) s = fmaf(h2[i], W.wh[i * (A + 1) + o], s);
    return s;
}
// sample, log-probability, clip, MPC inputs of one environment.  out_mean [A + 1] = heads; noise [A]
MPC_HD void finish_action(const PolicyWeights &W, int A, const float *heads, const float *noise, int version_v1, int clip,
                          const double *default_weights, float *actions, float *value, float *log_prob, double *mpc_weights,
                          double *mpc_ref_speed) {
    float q = 0.0f;
    for (int a = 0; a < A; ++a) {
        actions[a] = fmaf(noise[a], W.std[a], heads[a]);      // torch.addcmul(mean, noise, std)
        q += noise[a] * noise[a];
    }
    *value = heads[A];
    *log_prob = -0.5f * q - W.c0[0];
    auto clipped = [&](int a) {
        const float v = actions[a];
        return (double)(clip ? (v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v)) : v);
    };
    if (version_v1) {
        if (mpc_weights)
            for (int a = 0; a < 3; ++a) mpc_weights[a] = clipped(a);
    } else {
        if (mpc_weights && default_weights)       // v0: the caller usually hands its default weights to the MPC directly
            for (int a = 0; a < 3; ++a) mpc_weights[a] = default_weights[a];
        if (mpc_ref_speed) *mpc_ref_speed = clipped(0);
    }
}

// What thread j (0 .. 127) of environment b's workgroup writes for the buffer row `pos` (RolloutBuffer's layout: [obs 80 |
// action A | reward | episode_start | value | log_prob | (terminal_obs 80 | truncated)]) and the carry-over; returns, for
// j == 0, the bits to count: 1 episode finished, 2 crashed, 4 arrived, 8 solve not converged.
struct RecordArgs {
    int B, A, cols, keep_terminal;
    float *row;                 // [T][B][cols]
    double *mpc_actions_buf;    // [T][B][2]
    float *last_obs;            // [B][80] in / out
    float *last_starts;         // [B]     in / out
    const float *actions, *values, *log_probs;
    const double *mpc_act;      // [B][2]
    const int32_t *mpc_status;  // [B]
    const float *new_obs, *reward;
    const uint8_t *done;
    const float *terminal_obs;
    const uint8_t *truncated, *crashed, *arrived;
    uint8_t *dones_out;
};
// inside = false (the position is outside the buffer's T rows): nothing is written to the buffer; carry-over and counts as usual
MPC_HD int record_thread(const RecordArgs &r, long long pos, int b, int j, bool inside = true) {
    constexpr int O = kObsDim;
    float *row = r.row + ((size_t)(inside ? pos : 0) * r.B + b) * r.cols;
    if (j < O) {
        if (inside) row[j] = r.last_obs[(size_t)b * O + j];                  // the observation the policy acted on
        r.last_obs[(size_t)b * O + j] = r.new_obs[(size_t)b * O + j];        // ... and the one it sees next
        if (inside && r.keep_terminal) row[O + r.A + 4 + j] = r.terminal_obs[(size_t)b * O + j];
    }
    if (inside && j < r.A) row[O + j] = r.actions[(size_t)b * r.A + j];
    if (j != 0) return 0;
    const int c = O + r.A;
    if (inside) {
        row[c] = r.reward[b];
        row[c + 1] = r.last_starts[b];
        row[c + 2] = r.values[b];
        row[c + 3] = r.log_probs[b];
        if (r.keep_terminal) row[c + 4 + O] = r.truncated[b] ? 1.0f : 0.0f;
        r.mpc_actions_buf[((size_t)pos * r.B + b) * 2 + 0] = r.mpc_act[(size_t)b * 2 + 0];
        r.mpc_actions_buf[((size_t)pos * r.B + b) * 2 + 1] = r.mpc_act[(size_t)b * 2 + 1];
    }
    r.last_starts[b] = r.done[b] ? 1.0f : 0.0f;
    r.dones_out[b] = r.done[b] ? 1 : 0;
    const int st = r.mpc_status[b];
    return (r.done[b] ? 1 : 0) | (r.crashed[b] ? 2 : 0) | (r.arrived[b] ? 4 : 0) | ((st != 0 && (st < 5 || st > 7)) ? 8 : 0);
}

// ---- end of a rollout: truncation bootstrap + GAE ---------------------------------------------------------------------
// Every operation is a separate float32 multiplication / addition in the order torch's elementwise kernels apply them
// (RolloutBuffer.compute_returns_and_advantage / bootstrap_truncated in rollout.py, the torch form of the same arithmetic),
// so the two agree bit for bit; no multiply-add contraction.
MPC_HD float gmul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
MPC_HD float gadd(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
struct GaeArgs {
    int T, B, A, cols, keep_terminal;
    float *row;                       // [T][B][cols]   rewards are updated in place by the bootstrap
    const float *last_values;         // [B]    V(observation after the last step)
    const uint8_t *dones;             // [B]    the last step ended an episode
    const float *terminal_values;     // [T][B] V(terminal observation of step t) or null (a2c / nothing to bootstrap)
    float gamma, gamma_lambda;        // float32(gamma), float32(gamma * gae_lambda) - Python scalars enter torch kernels so
    float *advantages, *returns;      // [T][B]
};
// step t of environment b: bootstrapped reward (written back), TD residual delta_t and the recurrence's coefficient
MPC_HD void gae_terms(const GaeArgs &g, int b, int t, float *delta, float *coef) {
    constexpr int O = kObsDim;
    const int c = O + g.A;
    float *row = g.row + ((size_t)t * g.B + b) * g.cols;
    float r = row[c];
    if (g.keep_terminal && g.terminal_values) {
        r = gadd(r, gmul(gmul(g.gamma, g.terminal_values[(size_t)t * g.B + b]), row[c + 4 + O]));
        row[c] = r;
    }
    float nnt, nv;                    // "next non terminal", value of the next observation
    if (t == g.T - 1) {
        nnt = gadd(1.0f, -(g.dones[b] ? 1.0f : 0.0f));
        nv = g.last_values[b];
    } else {
        const float *nx = row + (size_t)g.B * g.cols;
        nnt = gadd(1.0f, -nx[c + 1]);
        nv = nx[c + 2];
    }
    *delta = gadd(gadd(r, gmul(gmul(g.gamma, nv), nnt)), -row[c + 2]);
    *coef = gmul(g.gamma_lambda, nnt);
}
MPC_HD float gae_step(float delta, float coef, float gae_next) { return gadd(delta, gmul(coef, gae_next)); }
MPC_HD void gae_store(const GaeArgs &g, int b, int t, float gae) {
    const float v = g.row[((size_t)t * g.B + b) * g.cols + kObsDim + g.A + 2];
    g.advantages[(size_t)t * g.B + b] = gae;
    g.returns[(size_t)t * g.B + b] = gadd(gae, v);
}

}  // namespace glue
}  // namespace mpc

/*
 * mpc_mi355x.h - C ABI of the MI355X (gfx950) batched nonlinear-MPC solve engine.
 *
 * Drop-in boundary for the hot path of SaeedRahmani/MPC-RL_for_AVs:
 *   PureMPC_Agent._solve()   reference agents/pure_mpc.py:80-318   (one NLP per call, CasADi -> IPOPT)
 *   PureMPC_Agent.predict()  reference agents/pure_mpc.py:68-78
 * The reference has no FFI for this path (it is a plain Python method that builds a CasADi graph and
 * calls `ca.nlpsol('solver','ipopt',...)`, agents/pure_mpc.py:285-300); the entry points below are what a
 * binding for it binds instead: plain pointers and sizes, no Python or torch types.  INTEGRATION.md shows
 * the ctypes stub that replaces the body of `_solve`.
 *
 * Conventions
 *   - all floating-point data is IEEE double; indices are int32; flags uint8 / uint32
 *   - arrays are dense row-major with the batch index slowest: state[B][4], vref[B][N+1], ...
 *   - every call returns 0 on success or a negative MPC_ERR_* code; mpc_last_error() gives the text
 *   - no exceptions cross this boundary; per-instance solver outcomes go to status[] / iters[]
 *   - a handle belongs to one (process, device); calls on one handle must not overlap, and neither may the work they
 *     enqueue (MPC_FLAG_NO_SYNC) on different streams - the handle owns scratch such as the batch's launch order and the
 *     per-environment records - with one exception: mpc_solve_batch with MPC_FLAG_THROUGHPUT, made for batches in flight
 */
#ifndef MPC_MI355X_H
#define MPC_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_ABI_VERSION 8
#define MPC_MAX_HORIZON 64
#define MPC_MAX_OTHERS 16

/* error codes (return values) */
#define MPC_OK 0
#define MPC_ERR_INVALID_ARG (-1)
#define MPC_ERR_NO_DEVICE (-2)   /* no HIP device / HIP runtime error: the engine has no CPU fallback */
#define MPC_ERR_HIP (-3)
#define MPC_ERR_NO_REFERENCE (-4) /* mpc_set_reference() has not been called */

/* mpc_solve_batch flags */
#define MPC_FLAG_COLLISION_COST 1u /* add the distance/collision terms of agents/archive/pure_mpc.py:189-206 */
#define MPC_FLAG_DEVICE_PTRS 2u    /* all data pointers are device memory (else host memory, copied internally) */
#define MPC_FLAG_NO_SYNC 4u        /* with DEVICE_PTRS: enqueue only, do not synchronise the stream */
#define MPC_FLAG_WARM_START 8u     /* NOT in the reference (which cold-starts every solve, agents/pure_mpc.py:240-246):
                                      start from given controls instead of zeros.  They are clamped 0.1 % inside their
                                      bounds; if their rollout leaves the state bounds the cold start is used.  Changes
                                      the iterates (fewer of them), not the KKT point aimed at.
                                      mpc_solve_batch: U (required) holds the initial controls on entry;
                                      mpc_predict_batch: each environment starts from its previous solution advanced by
                                      one stage (kept in the handle, forgotten by the reset calls);
                                      mpc_reset_env_mask: forget only that memory, keep the detector state */

/* mpc_predict_batch only: the reference's call sequence _parse_obs -> _check_collision -> _solve as separate calls
 * (agents/pure_mpc.py:68-78).  Without either flag one call does all of it. */
#define MPC_FLAG_DETECT_ONLY 16u   /* _check_collision alone (agents/pure_mpc.py:552-676): advance the detector records of
                                      the B environments for this observation, solve nothing (weights, act may be NULL);
                                      mpc_get_env_state then serves is_collide, conflict_index, ... */
#define MPC_FLAG_DETECTED 32u      /* _solve after such a call for the SAME observation: the records are not advanced
                                      again; ego index, speed profile (agents/pure_mpc.py:678-724) and the solve only */

#define MPC_FLAG_THROUGHPUT 64u     /* the caller keeps several batches in flight on different streams: launch the build of the
                                      solve kernel for four resident waves per SIMD whatever the batch size.  By default the
                                      engine picks the build by how deep ONE batch fills the SIMDs (B <= 4 waves per SIMD, 4096
                                      on an MI355X: the latency build, 218 registers, two resident waves; else the 128-register
                                      one), which is the faster choice when batches run one at a time and the slower one when
                                      six of them share the GPU. */

#define MPC_FLAG_STRICT_DISCONTINUITY 128u /* reference-strict reading of the d = 1 discontinuity of the collision cost
                                      (agents/archive/pure_mpc.py:189-196): an instance that ends on it - MPC_STATUS_CONVERGED_ON_KINK
                                      or MPC_STATUS_ACCEPTABLE_ON_KINK - is reported as MPC_STATUS_KINK_UNSOLVED, i.e. NOT solved,
                                      with the last iterate as the result: what the reference's IPOPT reports for a point no
                                      smooth method accepts (solver.stats()['success'] false, the last iterate is used,
                                      agents/pure_mpc.py:303-305).  The iterates OF THAT SOLVE are the same with and without the flag; with
                                      MPC_FLAG_WARM_START the next closed-loop step differs (a solve reported unsolved does not seed it). */

/* per-instance solver status written to status[] */
#define MPC_STATUS_CONVERGED 0
#define MPC_STATUS_MAX_ITER 1       /* last iterate returned, like the reference (agents/pure_mpc.py:303-305) */
#define MPC_STATUS_FACTORIZATION 2
#define MPC_STATUS_INFEASIBLE_START 3 /* the initial state violates the state bounds of agents/pure_mpc.py:272-274 */
#define MPC_STATUS_STALLED 4          /* no acceptable step in 3 consecutive iterations, or (mpc_config.stall_window) no
                                         halving of the KKT error within the window; last iterate returned */
#define MPC_STATUS_CONVERGED_ON_KINK 5 /* converged, with the collision cost on, to a point that holds a vehicle exactly at
                                         the d = 1 m discontinuity of that cost (agents/archive/pure_mpc.py:189-196:
                                         100/d^2 outside, 1000/d^2 inside): a KKT point of the outer branch with
                                         |p - o|^2 >= 1 as a constraint, i.e. a local minimiser of the discontinuous
                                         objective.  IPOPT has no such notion.  Observed with the restatement of its
                                         algorithm at the reference's settings on the 148 such instances of BASELINE
                                         config 3 (profiles/r03_parity_vs_ipopt.txt, part b): 99 end without success
                                         (70 with steps below alpha_min, where IPOPT proper would try its restoration
                                         phase; 29 at max_iter 1000, half of them within 1e-3 of d = 1) and the reference
                                         would act on that last iterate (agents/pure_mpc.py:303-305); 49 converge to a
                                         smooth minimiser on one side of the jump (26 inside d < 1, 23 outside), a
                                         different action than this status returns in 35 of them. */

#define MPC_STATUS_ACCEPTABLE 6        /* IPOPT's "Solved To Acceptable Level" with IPOPT's defaults, which are live in the
                                         reference (it sets only max_iter, tol and print options, agents/pure_mpc.py:291-296):
                                         the scaled KKT error was at most acceptable_tol = 1e-6 in acceptable_iter = 15
                                         consecutive iterations while mpc_config.tol is tighter.  casadi reports it as success
                                         like "Solve Succeeded".  Never returned when tol >= 1e-6 (the reference's own tol). */
#define MPC_STATUS_ACCEPTABLE_ON_KINK 7 /* the same with a vehicle held at the d = 1 discontinuity (see status 5) */
#define MPC_STATUS_KINK_UNSOLVED 8     /* MPC_FLAG_STRICT_DISCONTINUITY only: would be status 5 or 7; counted as not solved */
/* solved, as a caller should count it: status 0, 5, 6, 7 */
#define MPC_STATUS_IS_SOLVED(st) ((st) == 0 || ((st) >= 5 && (st) <= 7))

typedef struct mpc_handle mpc_handle;

typedef struct mpc_config {
    int32_t struct_size; /* sizeof(mpc_config), for ABI evolution */
    int32_t horizon;     /* N; reference cfg key `horizon` (config/cfg.yaml:90 default 16, BASELINE uses 20) */
    double dt;           /* 1 / policy_frequency (agents/base_agent.py:43), 0.1 */
    int32_t max_iter;    /* interior-point iteration cap (reference: ipopt.max_iter 1000, agents/pure_mpc.py:294) */
    int32_t device;      /* HIP device ordinal */
    double tol;          /* scaled KKT tolerance (reference: ipopt.tol 1e-6, agents/pure_mpc.py:295); default 1e-8 */
    double w_distance;   /* cfg key weight_distance  (config/cfg.yaml:105), used with MPC_FLAG_COLLISION_COST */
    double w_collision;  /* cfg key weight_collision (config/cfg.yaml:106), used with MPC_FLAG_COLLISION_COST */
    int32_t ltv_passes;  /* iterative-linear agent: linearisation passes per call, the trip count of the loop at
                            agents/pure_mpc_linear.py:189 (`for _ in range(1)`); default 1 */
    int32_t stall_window; /* 0 (default) = off.  W > 0: a solve whose scaled KKT error has not halved within W consecutive
                            iterations ends with MPC_STATUS_STALLED instead of running on to max_iter - what makes the
                            reference's max_iter 1000 affordable in a batch, whose time is that of its slowest instance
                            (profiles/r03_tail.txt: one instance in a thousand never converges and costs 40 ms).  64
                            gives up on 2 - 6 of 4096 BASELINE-config-3 instances that would converge after 100 - 440
                            iterations.  Not an IPOPT option; the reference runs such instances to max_iter. */
} mpc_config;

/* ABI version of the loaded library (MPC_ABI_VERSION it was built with). */
int mpc_version(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char *mpc_last_error(void);

/* Fill *cfg with the defaults (horizon 20, dt 0.1, max_iter 100, tol 1e-8, weights 10 / 1, device 0).
 * Writes sizeof(mpc_config) bytes of THIS header's layout: only for callers compiled against this header. */
void mpc_default_config(mpc_config *cfg);

/* The same for bindings that declare the struct themselves (ctypes, cgo, ...): `size` is the size of the caller's
 * object.  Nothing is written and MPC_ERR_INVALID_ARG is returned unless it equals the library's sizeof(mpc_config),
 * so a binding built for an older layout fails loudly instead of being overrun. */
int mpc_default_config_sized(mpc_config *cfg, int32_t size);

/* Create an engine bound to cfg->device.  Replaces PureMPC_Agent.__init__ (agents/pure_mpc.py:24-63). */
int mpc_create(const mpc_config *cfg, mpc_handle **out);

void mpc_destroy(mpc_handle *h);

/* Upload the global reference path: ref[M][4] = x, y, v, heading (host pointer).
 * Replaces the `reference_states` property (agents/base_agent.py:118-154), M = 85 there. */
int mpc_set_reference(mpc_handle *h, const double *ref, int32_t M);

/*
 * Solve B independent MPC instances (replaces B calls of PureMPC_Agent._solve, agents/pure_mpc.py:80-318).
 *   state      [B][4]    x, y, theta, v of the ego vehicle               (agents/pure_mpc.py:233-238)
 *   ego_index  [B]       index of the nearest reference point            (agents/pure_mpc.py:106-109)
 *   vref       [B][N+1]  reference speed of stage k = ref[min(ego_index+k, M-1)][2] after
 *                        update_reference_states (agents/pure_mpc.py:678-724); NULL = speeds of the table
 *   weights    [B][3]    weight_speed, weight_control, weight_input_diff (agents/pure_mpc.py:96-104)
 *   is_collide [B]       1: speed weight forced to 100 (agents/pure_mpc.py:143-147) and, with
 *                        MPC_FLAG_COLLISION_COST, the 3000 v^2 term is active
 *   others     [B][V][4] x, y, speed, heading of the other vehicles (constant-velocity model,
 *                        agents/base_agent.py:172-174); read only with MPC_FLAG_COLLISION_COST; may be NULL
 * outputs (u0 required, the rest optional = NULL):
 *   u0     [B][2]      first control (acceleration, steer) = MPC_Action    (agents/pure_mpc.py:311-318)
 *   U      [B][N][2]   full control sequence;  X [B][N+1][4] state trajectory
 *   status [B], iters [B]
 *   stream: hipStream_t to enqueue on (NULL = the null stream)
 */
int mpc_solve_batch(mpc_handle *h, int32_t B, const double *state, const int32_t *ego_index, const double *vref,
                    const double *weights, const uint8_t *is_collide, const double *others, int32_t V,
                    uint32_t flags, double *u0, double *U, double *X, int32_t *status, int32_t *iters,
                    void *stream);

/*
 * Observation-level entry: everything PureMPC_Agent.predict() does (agents/pure_mpc.py:68-78) for B parallel
 * environments, on the device: observation parsing (agents/base_agent.py:81-116), the path-crossing collision
 * detector with its 10-step memory (agents/pure_mpc.py:552-676), the rewrite of the reference speed profile
 * (agents/pure_mpc.py:678-724) and the solve.  Environment b of the call owns state record b inside the handle
 * (records are created fresh on first use and persist across calls; see mpc_reset_env_state).
 *   obs            [B][vehicles_count][8] float32: presence, x, y, vx, vy, heading, sin_h, cos_h, absolute values,
 *                  row 0 = ego (config/config.py:4-26); present rows are contiguous (agents/base_agent.py:106-114)
 *   vehicles_count rows per observation, 1..MPC_MAX_OTHERS+1 (cfg `observation.vehicles_count`, 10)
 *   weights        [B][3] weight_speed, weight_control, weight_input_diff (RL action or the cfg defaults)
 *   ref_speed      [B] RL reference-speed override (agents/pure_mpc.py:683-688) or NULL
 *   flags          MPC_FLAG_COLLISION_COST: distance/collision terms over the observed vehicles; MPC_FLAG_DEVICE_PTRS,
 *                  MPC_FLAG_NO_SYNC as for mpc_solve_batch; MPC_FLAG_DETECT_ONLY / MPC_FLAG_DETECTED (above)
 *   act            [B][2] acceleration, steer;  status/iters [B] optional
 */
int mpc_predict_batch(mpc_handle *h, int32_t B, const float *obs, int32_t vehicles_count, const double *weights,
                      const double *ref_speed, uint32_t flags, double *act, int32_t *status, int32_t *iters,
                      void *stream);

/*
 * Iterative-linear MPC: B calls of IterativeLinearMPC_Agent._solve (agents/pure_mpc_linear.py:153-203): nearest
 * reference point (:38-60), forward simulation of the stored control profile (predict_motion, :84-110), linearisation
 * about it (linear_model_matrix, :62-82) and the QP of _linear_mpc_control (:205-257), which the reference gives to
 * cvxpy/ECOS and this engine solves with a Riccati-based primal-dual interior-point method.
 *   state  [B][4]     x, y, v, yaw of the ego vehicle - the state order of that agent (:23, :198)
 *   U      [B][N][2]  in: the stored profile (oa, od), zeros for a first call (:190-192); out: the new profile where
 *                     status is 0, unchanged elsewhere (:193-196).  Required.
 *   u0     [B][2]     acceleration, steer = U[b][0] where status is 0, else (0, 0) (:195)
 *   X      [B][N+1][4] optional: states of the linear model under the new controls (x, y, v, yaw)
 *   status [B]        MPC_STATUS_CONVERGED / MAX_ITER / FACTORIZATION, or MPC_STATUS_INFEASIBLE_START when the ego
 *                     speed is outside [0, 40/3.6] (the QP then has no feasible point: x[2,0] == v0, :252-256)
 *   target_index [B]  optional: index of the nearest reference point
 *   flags: MPC_FLAG_DEVICE_PTRS, MPC_FLAG_NO_SYNC.  cfg.horizon, dt, max_iter apply; the weights and bounds are the
 *   module constants of the reference (:27-37).
 */
int mpc_ltv_solve_batch(mpc_handle *h, int32_t B, const double *state, uint32_t flags, double *u0, double *U, double *X,
                        int32_t *status, int32_t *iters, int32_t *target_index, void *stream);

/* Observation-level entry of the same agent: Agent.predict (agents/base_agent.py:54-73) = _parse_obs + _solve for B
 * parallel environments.  Environment b owns a stored profile inside the handle (zeros until its first solve, kept
 * across calls, forgotten by mpc_reset_env_state / mpc_reset_env_mask).  obs, vehicles_count, act, status, iters as
 * for mpc_predict_batch. */
int mpc_ltv_predict_batch(mpc_handle *h, int32_t B, const float *obs, int32_t vehicles_count, uint32_t flags, double *act,
                          int32_t *status, int32_t *iters, void *stream);

/* Episode boundaries: forget collision memory / stop point of the listed environments (host array of n ids);
 * env_ids == NULL or n < 0 resets all.  Replaces constructing a new PureMPC_Agent (agents/pure_mpc.py:38-43,63). */
int mpc_reset_env_state(mpc_handle *h, const int32_t *env_ids, int32_t n, void *stream);

/* Same, from a done-mask: environment b is reset where done[b] != 0 (device pointer with MPC_FLAG_DEVICE_PTRS;
 * with MPC_FLAG_NO_SYNC the reset is only enqueued). */
int mpc_reset_env_mask(mpc_handle *h, int32_t B, const uint8_t *done, uint32_t flags, void *stream);

/* Detector state of environments 0..B-1 after the last mpc_predict_batch (host arrays, any may be NULL; synchronises):
 * is_collide, ego_index, collision_memory, stop_index (-1 = none) [B]; conflict_index [B][MPC_MAX_OTHERS] (-1 = none);
 * conflict_points [B][MPC_MAX_OTHERS][2] (NaN = none): where the predicted paths cross.
 * Serves the attributes callers / plots read from the agent (agents/pure_mpc.py:38-43, 589-593: is_collide,
 * conflict_points, conflict_index, agent_collide, stop_point). */
int mpc_get_env_state(mpc_handle *h, int32_t B, int32_t *is_collide, int32_t *ego_index, int32_t *collision_memory,
                      int32_t *stop_index, int32_t *conflict_index, double *conflict_points);

/* Checkpoint / resume of the per-environment detector records (the reference never saves them: its agent state is lost
 * with the process, agents/pure_mpc.py:38-43).  mpc_env_state_bytes(): size of one opaque record;
 * mpc_save_env_state copies records 0..B-1 to a host buffer of B * that size; mpc_set_env_state puts them back (into
 * this or another handle of the same library version), growing the handle's capacity if needed (not while a captured
 * hipGraph still addresses the old buffers: size the handle with mpc_reserve_envs first).  Every record is validated
 * (counts, indices against the reference table) and a bad one refuses the whole call; the restored environments lose
 * their warm-start controls and stored LTV profile (those belong to the episode that was running).  Both synchronise. */
int64_t mpc_env_state_bytes(void);
int mpc_save_env_state(mpc_handle *h, int32_t B, void *records);
int mpc_set_env_state(mpc_handle *h, int32_t B, const void *records);

/* Size the per-environment buffers (detector records, warm-start controls) for B environments now.  They otherwise grow
 * on demand, which frees the old buffers: not allowed inside a stream capture (a captured graph holds the addresses)
 * and it waits for the device to go idle.  Call this once before capturing a step into a hipGraph. */
int mpc_reserve_envs(mpc_handle *h, int32_t B);

/* Problem data the last mpc_predict_batch derived from the observations (host arrays, any may be NULL; synchronises):
 * state [B][4], ego_index [B], vref [B][N+1], is_collide [B], others [B][max(vehicles_count-1,1)][4], nveh [B].
 * Diagnostics / tests: these are exactly the arguments mpc_solve_batch would take. */
int mpc_get_last_inputs(mpc_handle *h, int32_t B, double *state, int32_t *ego_index, double *vref,
                        uint8_t *is_collide, double *others, int32_t *nveh);

/* Diagnostics for parity tests: with on != 0 every following mpc_predict_batch also keeps the polylines its detector
 * worked on.  mpc_get_last_paths then returns (host arrays, any may be NULL; synchronises), for environments 0..B-1 of that
 * call: ego_path [B][31][2] and ego_len [B] - what predict_ego_future_positions returns (agents/pure_mpc.py:459-527; 0
 * points where the environment replayed its collision memory, :558-563, and predicted nothing) - and agent_paths
 * [B][max(vehicles_count-1,1)][31][2] float32 - predict_future_positions per observed vehicle (:529-550; zeros for
 * absent vehicles).  Costs one memset and B * (500 + 248 V) bytes per call; off by default. */
int mpc_set_diagnostics(mpc_handle *h, int32_t on);
int mpc_get_last_paths(mpc_handle *h, int32_t B, double *ego_path, int32_t *ego_len, float *agent_paths);

/*
 * Synthetic intersection environment for MPC-in-the-loop rollouts (BASELINE configs 4-5; highway-env, which the reference
 * steps at agents/ppo_mpc.py:430-432, is not available offline): ONE launch per policy step for B environments - vehicle
 * models, respawn, reward / termination shape of envs/intersection_env_Feb2025_v1.py:80-155, terminal observation,
 * auto-reset and the next observation in the layout of config/config.py:10-26 (csrc/mpc_synth_env.hpp).  All pointers are
 * DEVICE memory on `device`; the call only enqueues on `stream` (capturable in a hipGraph).  State arrays, updated in
 * place: ego [B][4] x, y, heading, speed; opos [B][K'][2], ospeed / ohead [B][K'] f64, oactive [B][K'] u8 with
 * K' = max(K, 1); t [B] steps of the episode; rng_counter [B] (zero-initialised; with `seed` and env_offset + b it keys a
 * counter-based generator, so shards of one job draw distinct streams).  action [B][2] acceleration, steer.
 * Outputs: obs [B][10][8] f32 what the policy sees next (after the auto-reset), terminal_obs the same before it, reward [B]
 * f32, done / truncated / crashed / arrived [B] u8.  reset_all != 0: start fresh episodes everywhere and write obs only.
 * Limits (sixteen lanes per environment): K <= 15 other vehicles, M <= 128 route points; MPC_ERR_INVALID_ARG beyond.
 */
int mpc_synth_env_step(int32_t device, int32_t B, int32_t K, double dt, double spawn_probability, uint64_t seed,
                       int32_t env_offset, const double *ref_xy, int32_t M, const double *action, double *ego, double *opos,
                       double *ospeed, double *ohead, uint8_t *oactive, int32_t *t, int64_t *rng_counter, float *obs,
                       float *terminal_obs, float *reward, uint8_t *done, uint8_t *truncated, uint8_t *crashed,
                       uint8_t *arrived, int32_t reset_all, void *stream);

/*
 * Rollout glue for the same configurations (csrc/mpc_rollout_glue.hpp): what a step of the reference's collect_rollouts does
 * besides the MPC call and env.step, as two launches per step instead of ~40 torch kernels, and one at the rollout's end.  Device pointers, enqueue only.
 *
 * mpc_policy_act: the SB3 MlpPolicy-shaped actor-critic of PPO_MPC / A2C_MPC (agents/ppo_mpc.py:390-394; 80 -> H -> H tanh
 * twice, Gaussian head) for B observations obs [B][80] f32: both towers as one 80 -> 2H -> 2H -> (A + 1) network with the
 * weights laid out as w1 [80][2H], b1 [2H], w2 [2H][2H] (block diagonal), b2 [2H], wh [2H][A + 1], bh [A + 1], std [A] =
 * exp(log_std), c0 [1] = sum(log_std) + A / 2 log(2 pi); noise [B][A] the sample's standard normal draws: with noise_step
 * == NULL read from the caller (its own generator), otherwise DRAWN here - counter-based, keyed by (noise_seed, env_offset + b,
 * noise_step[0], component), no generator state, replayable in a hipGraph - and left in `noise`.
 * Outputs: actions [B][A] = mean + std * noise, values [B], log_probs [B], and the MPC's inputs as the reference maps the
 * action: version_v1 == 0 -> mpc_ref_speed [B] f64 = action 0 (agents/ppo_mpc.py:410-414), else mpc_weights [B][3] f64 =
 * actions 0..2 (:416-420); clip != 0 clips to the Box(-1, 1) action space first (PPO, :399-407; A2C does not, a2c_mpc.py:138-144).
 *
 * mpc_rollout_record: row pos_dev[0] of the rollout buffer row [T][B][cols] = [obs 80 | action A | reward | episode_start |
 * value | log_prob | (terminal_obs 80 | truncated)] (rollout_buffer.add, agents/ppo_mpc.py:462-469) and of mpc_actions_buf
 * [T][B][2]; last_obs <- new_obs, last_starts <- done; counts [4] += finished / crashed / arrived episodes and solves whose
 * status is not solved (MPC_STATUS_IS_SOLVED); dones_out <- done; pos_dev[0] += 1 (ticket: one zero-initialised int32 of
 * scratch) and, if given, step_counter[0] += 1 (the policy steps taken so far: mpc_policy_act's noise_step).  T is the number
 * of rows of the buffer: a step taken with pos_dev[0] >= T (or < 0) writes NOTHING to row / mpc_actions_buf - the torch path it
 * replaces raises an index error there - and is counted in counts[4] instead (counts is [5]: finished, crashed, arrived,
 * unsolved, refused steps); the carry-over, the counters and the position still advance.
 *
 * mpc_rollout_finish: the end of a rollout of T steps (1 <= T <= 8192) over the same buffer (agents/ppo_mpc.py:471-476
 * `rollout_buffer.compute_returns_and_advantage`, stable-baselines3's arithmetic in float32, operation by operation): if
 * keep_terminal and terminal_values [T][B] (V of every step's terminal observation) is given, the reward column becomes
 * reward + gamma * terminal_values * truncated first (the PPO agents' truncation bootstrap, :451-461); then, with
 * last_values [B] = V(observation after the last step) and dones [B] = the last step ended an episode, per environment
 *   delta_t = r_t + gamma V_{t+1} (1 - start_{t+1}) - V_t,  A_t = delta_t + gamma lambda (1 - start_{t+1}) A_{t+1}
 * backwards from t = T - 1 (V_T = last_values, start_T = dones); advantages [T][B] = A, returns [T][B] = A + V.
 */
int mpc_policy_act(int32_t device, int32_t B, int32_t A, int32_t H2, const float *obs, const float *w1, const float *b1,
                   const float *w2, const float *b2, const float *wh, const float *bh, const float *std_, const float *c0,
                   float *noise, uint64_t noise_seed, int32_t env_offset, const int64_t *noise_step, int32_t version_v1,
                   int32_t clip, float *actions, float *values, float *log_probs, double *mpc_weights, double *mpc_ref_speed,
                   void *stream);
int mpc_rollout_record(int32_t device, int32_t T, int32_t B, int32_t A, int32_t cols, int32_t keep_terminal, float *row,
                       double *mpc_actions_buf, int64_t *pos_dev, int32_t *ticket, float *last_obs, float *last_starts,
                       const float *actions, const float *values, const float *log_probs, const double *mpc_act,
                       const int32_t *mpc_status, const float *new_obs, const float *reward, const uint8_t *done,
                       const float *terminal_obs, const uint8_t *truncated, const uint8_t *crashed, const uint8_t *arrived,
                       int64_t *counts, uint8_t *dones_out, int64_t *step_counter, void *stream);
int mpc_rollout_finish(int32_t device, int32_t T, int32_t B, int32_t A, int32_t cols, int32_t keep_terminal, float *row,
                       const float *last_values, const uint8_t *dones, const float *terminal_values, double gamma,
                       double gae_lambda, float *advantages, float *returns, void *stream);

/*
 * Diagnostics: the NLP's functions at GIVEN points, evaluated by the solve kernel's own code (csrc/mpc_wave.hpp:
 * Solver::evaluate - stage_terms / track / dist, which judge every line-search trial, and the model step of the rollouts),
 * so that f(z) and g(z) computed by the reference's statements (agents/pure_mpc.py:128-283; tests/golden/
 * reference_sequences.npz) pin the device directly and not only through solutions.  Host pointers, synchronous.
 * Problem data as for mpc_solve_batch (the ego state is X[b][0]); X [B][N+1][4], U [B][N][2] the point;
 * f [B] <- the objective, unscaled, terms of k = 0 .. N-1 (:128-212; with MPC_FLAG_COLLISION_COST plus the terms of
 * agents/archive/pure_mpc.py:189-206); x_next [B][N][4] <- the model's successor of (X[k], U[k]) (:220-257): the reference's
 * dynamics constraint is X[k+1] - x_next[k] = 0.
 */
int mpc_eval_nlp(mpc_handle *h, int32_t B, const int32_t *ego_index, const double *vref, const double *weights,
                 const uint8_t *is_collide, const double *others, int32_t V, uint32_t flags, const double *X, const double *U,
                 double *f, double *x_next);

/* LDS bytes one workgroup (= one wave = one instance) of the solve kernel uses with V other vehicles in the
 * collision-cost term (V = 0: term off): 13.5 KB at horizon 20 with 8 vehicles, the same in every build since round 5 (B is
 * accepted for compatibility and ignored).  The engine keeps no per-instance solver state in HBM.  (diagnostics / capacity
 * planning) */
int64_t mpc_workspace_bytes(const mpc_handle *h, int32_t B, int32_t V);

/* Do kernels on two streams of `device` run side by side?  *overlap <- 1 if a ~0.3 ms one-wave timer kernel enqueued on each
 * finishes in the time of one, 0 if they take the time of two: the HIP runtime multiplexes its streams onto a few hardware
 * queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share one SERIALISE, whatever the kernels - which of a
 * process's streams share is decided when they are created (measured round 6: of torch's pool streams 0-7 two shared a
 * queue, 8-15 none; batches in flight on them gave 2.6 against 3.45 M solves/s).  A caller that keeps several independent
 * batches in flight (MPC_FLAG_THROUGHPUT; the reference's vectorised environments in groups, agents/a2c_mpc.py:111-180) picks
 * its streams with this probe (engine.concurrent_streams).  Synchronises both streams; not capturable. */
int mpc_streams_overlap(int32_t device, void *stream_a, void *stream_b, int32_t *overlap);

#ifdef __cplusplus
}
#endif
#endif /* MPC_MI355X_H */

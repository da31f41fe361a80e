// Host build of mpc-rl_for_avs_amd/csrc/mpc_synth_env.hpp for tests only (-m "not gpu"): the fused environment step, one
// loop iteration per environment, with the argument list of mpc_synth_env_step minus device and stream.
#include <cmath>
#include <cstdint>

#include "../mpc-rl_for_avs_amd/csrc/mpc_synth_env.hpp"

extern "C" int synth_env_step(int B, int K, double dt, double spawn_probability, uint64_t seed, int env_offset,
                              const double *ref_xy, int M, const double *action, double *ego, double *opos, double *ospeed,
                              double *ohead, uint8_t *oactive, int32_t *t, int64_t *ctr, float *obs, float *terminal_obs,
                              float *reward, uint8_t *done, uint8_t *truncated, uint8_t *crashed, uint8_t *arrived,
                              int reset_all) {
    namespace env = mpc::env;
    if (K < 0 || K > env::kMaxOthers) return -1;
    const int Ks = K > 0 ? K : 1;
    for (int b = 0; b < B; ++b) {
        const env::View v{ego + (size_t)b * 4, opos + (size_t)b * Ks * 2, ospeed + (size_t)b * Ks, ohead + (size_t)b * Ks,
                          oactive + (size_t)b * Ks, t + b, ctr + b};
        float *o = obs + (size_t)b * env::kRows * env::kCols;
        if (reset_all) {
            const env::Rng r(seed, env_offset + b, *v.ctr);
            *v.ctr += 1;
            env::reset_env(v, K, r);
            env::observe(v, K, o);
            continue;
        }
        const env::StepOut so = env::step_env(v, K, dt, spawn_probability, seed, env_offset + b, ref_xy, M, action + (size_t)b * 2,
                                              terminal_obs + (size_t)b * env::kRows * env::kCols, o);
        reward[b] = so.reward;
        done[b] = so.done;
        truncated[b] = so.truncated;
        crashed[b] = so.crashed;
        arrived[b] = so.arrived;
    }
    return 0;
}

// preamble_sections.hip - where does the observation preamble (mpc_preamble_wave.hpp) spend its cycles?
// Development tool (not part of the product): the same preamble_env_wave with a context whose `tick(section)` accumulates cycle
// counter deltas per section in LDS, run for B environments (one wave each) on COPIES of their detector records; cycles per
// environment and section are returned.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libpreamble_sections.so tools/ubench/preamble_sections.hip
// driven by tools/gpu_preamble_sections.py.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../mpc-rl_for_avs_amd/csrc/mpc_wave_dev.hpp"
#include "../../mpc-rl_for_avs_amd/csrc/mpc_preamble.hpp"
#include "../../mpc-rl_for_avs_amd/csrc/mpc_preamble_wave.hpp"

namespace {
namespace pre = mpc::pre;

struct PreProfCtx : mpc::wave::WaveOpsT<3> {
    static constexpr int kN = 0;
    const double *table;
    int e0, M;
    unsigned long long *acc;   // [PT_COUNT] accumulators of this environment in LDS (lane 0 adds)
    unsigned long long last;
    __device__ __forceinline__ PreProfCtx(mpc::wave::lds_double_t *l, const double *t, int m, unsigned long long *a)
        : mpc::wave::WaveOpsT<3>{l}, table(t), e0(0), M(m), acc(a), last(__builtin_readcyclecounter()) {}
    __device__ __forceinline__ void tick(int s) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (threadIdx.x == 0) acc[s] += now - last;
        last = now;
    }
};

__global__ __launch_bounds__(64, 2) void prof_kernel(int B, const float *obs, int rows, const double *ref5, int M, int N, double dt,
                                                     pre::EnvState *env, double *state, int32_t *ego_index, double *vref,
                                                     uint8_t *is_collide, double *others, int Vslots, int32_t *nveh, int advance,
                                                     unsigned long long *cycles) {
    __shared__ double s_words[pre::preamble_wave_lds_doubles()];
    __shared__ int32_t s_conf[pre::kMaxOthers];
    __shared__ pre::P2 s_cpt[pre::kMaxOthers];
    __shared__ unsigned long long s_acc[pre::PT_COUNT];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    if (lane < pre::PT_COUNT) s_acc[lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PreProfCtx ctx((mpc::wave::lds_double_t *)s_words, ref5, M, s_acc);
    const pre::RefTable R{ref5, M};
    const pre::PreDiag diag{nullptr, nullptr, nullptr, Vslots};
    pre::preamble_env_wave(ctx, obs + (size_t)b * rows * pre::kObsCols, rows, R, N, dt, nullptr, env[b], state + (size_t)b * 4,
                           ego_index[b], vref + (size_t)b * (N + 1), is_collide[b], others + (size_t)b * Vslots * 4, Vslots, nveh[b],
                           advance != 0, s_conf, s_cpt, diag);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < pre::PT_COUNT) cycles[(size_t)b * pre::PT_COUNT + lane] = s_acc[lane];
}

void *g_scratch = nullptr;
size_t g_scratch_bytes = 0;

}  // namespace

extern "C" int preamble_sections_count() { return pre::PT_COUNT; }
extern "C" long long preamble_sections_record_bytes() { return (long long)sizeof(pre::EnvState); }

// obs [B][rows][8] f32, ref5 the engine's table layout ([M][5] x y heading sin cos, then [M] speeds), env [B] records (device,
// UPDATED like the product kernel would), cycles [B][PT_COUNT] u64: all device pointers.  Outputs go to a scratch buffer.
extern "C" int preamble_sections(int B, const float *obs, int rows, const double *ref5, int M, int N, double dt, void *env,
                                 int Vslots, int advance, unsigned long long *cycles) {
    if (B <= 0 || rows < 1 || rows > pre::kMaxOthers + 1 || Vslots < 1) return -1;
    const size_t per = 4 * 8 + 4 + (size_t)(N + 1) * 8 + 8 + (size_t)Vslots * 4 * 8 + 4;
    const size_t need = (size_t)B * (per + 64);
    if (need > g_scratch_bytes) {
        if (g_scratch) (void)hipFree(g_scratch);
        if (hipMalloc(&g_scratch, need) != hipSuccess) return -2;
        g_scratch_bytes = need;
    }
    char *p = static_cast<char *>(g_scratch);
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 63) / 64 * 64; return q; };
    double *state = reinterpret_cast<double *>(take((size_t)B * 4 * 8));
    double *vref = reinterpret_cast<double *>(take((size_t)B * (N + 1) * 8));
    double *others = reinterpret_cast<double *>(take((size_t)B * Vslots * 4 * 8));
    int32_t *ego_index = reinterpret_cast<int32_t *>(take((size_t)B * 4));
    int32_t *nveh = reinterpret_cast<int32_t *>(take((size_t)B * 4));
    uint8_t *is_collide = reinterpret_cast<uint8_t *>(take((size_t)B));
    if ((size_t)(p - static_cast<char *>(g_scratch)) > g_scratch_bytes) return -3;
    hipLaunchKernelGGL(prof_kernel, dim3((unsigned)B), dim3(64), 0, 0, B, obs, rows, ref5, M, N, dt,
                       static_cast<pre::EnvState *>(env), state, ego_index, vref, is_collide, others, Vslots, nveh, advance, cycles);
    if (hipGetLastError() != hipSuccess) return -4;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -5;
}

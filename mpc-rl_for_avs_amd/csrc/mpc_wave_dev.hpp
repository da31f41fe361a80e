// mpc_wave_dev.hpp - the device side of the CTX interface of mpc_wave.hpp (one wave64 = one workgroup = one MPC
// instance): LDS access, per-lane sections, the FP64 matrix core, lane permutations and wave reductions.
// WaveOps is the base of the product kernel's context (mpc_engine.hip) and of the profiling build
// (tools/ubench/wave_sections.hip); tests/cpu_wave_harness.cpp is the host model of the same interface.
#pragma once

#include <hip/hip_runtime.h>

#include "mpc_wave.hpp"

namespace mpc {
namespace wave {

typedef __attribute__((address_space(3))) double lds_double_t;

// DPP row controls (GFX9): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
// row_shl:n - lane i of a 16-lane row reads lane i + n of the same row, 0 beyond the row (bound_ctrl)
constexpr int kDppRowShl1 = 0x101, kDppRowShl2 = 0x102, kDppRowShl4 = 0x104, kDppRowShl8 = 0x108;

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_first_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Reduction inside each 16-lane row with symmetric partner exchanges (xor 1, xor 2, mirror within 8, mirror within
// 16): both partners combine the same two values, so all lanes of a row end bit-identical.
template <class OP>
__device__ __forceinline__ double row_reduce(double v, OP op) {
    v = op(v, dpp_f64<kDppXor1>(v));
    v = op(v, dpp_f64<kDppXor2>(v));
    v = op(v, dpp_f64<kDppHalfMirror>(v));
    v = op(v, dpp_f64<kDppMirror>(v));
    return v;
}

// Broadcast of lane J of every 16-lane row to the whole row: ONE v_mov_b64_dpp row_newbcast:J (the only DPP control the
// double-precision ALU of gfx90a+ supports, and exactly what a row-cooperative matrix-vector product needs: lane i of a row
// holds row i of the matrix and accumulates M[i][j] * bcast_j(x) - no LDS, no pair of 32-bit moves per operand).
template <int J>
__device__ __forceinline__ double row_bcast_f64(double v) {
    static_assert(J >= 0 && J < 16, "a row has 16 lanes");
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + J, 0xf, 0xf, true));
}

// Two broadcasts into one register: lanes 0..7 of every row take lane J, lanes 8..15 lane J + 8 of their row (the second
// move is bank-masked: banks 2, 3 = lanes 8..15).  The trial (lanes 0..7) and the linearised step (lanes 8..15) of a rollout
// row run the same matrix-vector products on their own halves: one multiply-add serves both.
template <int J>
__device__ __forceinline__ double row_bcast2_f64(double v) {
    static_assert(J >= 0 && J < 8, "lane J of the lower, J + 8 of the upper half row");
    const long long lo = __builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + J, 0xf, 0xf, true);
    return __longlong_as_double(__builtin_amdgcn_update_dpp(lo, __double_as_longlong(v), 0x150 + J + 8, 0xf, 0xc, false));
}

// RELAX = 0 is the build for four resident waves per SIMD (128 registers): fresh() / opaque() hide where a constant or a
// lane-derived table comes from, so that it is recomputed where it is used instead of living in registers across the
// iteration loop.  The relaxed builds let the compiler hoist: with both relaxed ~200 registers (two waves per SIMD) and
// 7 % less time per iteration for a wave that has its SIMD to itself (38.1 against 40.7 us, profiles/r03_latency.txt) -
// the builds the engine launches when the batch leaves SIMDs that empty anyway (mpc_engine.hip: dispatch_solve).
template <int RELAX>   // bit 0: fresh() is the identity, bit 1: opaque() is the identity, bit 2: the build has its SIMD to itself (opaque_shared)
struct WaveOpsT {
    static constexpr int kRelax = RELAX;                    // bit 3 in mpc_ltv.hpp: relax_bits
    lds_double_t *L;  // this instance's LDS words
    __device__ __forceinline__ double ld(int i) const { return L[i]; }
    __device__ __forceinline__ void st(int i, double v) { L[i] = v; }
    // A phase ends where other lanes may read what this one wrote to LDS.  The workgroup is one wave, whose LDS
    // instructions execute in issue order, so no hardware barrier or wait is needed - only the compiler must keep the
    // program order of the LDS accesses: an acquire-release fence at wavefront scope (stores of the phase stay above
    // it, loads of the next one below it; it emits no instruction) plus the wave barrier.
    // (the lane id is handed out through opaque(): whatever a phase derives from it - LDS addresses of its stage, role
    // flags - is then recomputed in that phase, a handful of integer operations, instead of being hoisted out of the
    // iteration loop and kept in registers the kernel does not have)
    template <class F>
    __device__ __forceinline__ void phase(F &&f) {
        f(opaque((int)threadIdx.x));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    template <class F>
    __device__ __forceinline__ void lanes(F &&f) {
        f((int)threadIdx.x);
    }
    __device__ __forceinline__ void mfma(PerLane<double> &a, PerLane<double> &b, PerLane<double> &cd) const {
        cd.v = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v, b.v, cd.v, 0, 0, 0);
    }
    __device__ __forceinline__ double lane_get(PerLane<double> &p, int lane) const { return readlane_f64(p.v, lane); }
    // dst (every lane) = src of lane J of the same 16-lane row
    template <int J>
    __device__ __forceinline__ void row_bcast(PerLane<double> &dst, PerLane<double> &src) const {
        dst.v = row_bcast_f64<J>(src.v);
    }
    template <int J>
    __device__ __forceinline__ void row_bcast2(PerLane<double> &dst, PerLane<double> &src) const {
        dst.v = row_bcast2_f64<J>(src.v);
    }
    // A wave-uniform double that the VALU computed sits in two vector registers like any per-lane value; moved to a scalar
    // register pair it costs none (and when scalar registers run out the compiler parks them in lanes of a vector
    // register, 32 doubles per register, instead of spilling vector registers to scratch memory).
    __device__ __forceinline__ double uni(double v) const { return readlane_first_f64(v); }
    __device__ __forceinline__ void sched_fence() const { __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ double keep(double v) const {
        asm volatile("" : "+v"(v));
        return v;
    }
    // A wave-uniform double (or a literal) as a value the compiler cannot trace back: whatever is computed from it is
    // computed where it is used.  Otherwise every loop-invariant product of solve constants (0.01 sf wc, 1/dt, ...) and
    // every 64-bit literal used twice is hoisted out of the iteration loop into a vector register pair for the whole
    // solve - 40 registers of the 128 a wave may hold if four waves are to share a SIMD.
    __device__ __forceinline__ double fresh(double v) const {
        if (RELAX & 1) return v;
        v = readlane_first_f64(v);       // folds away for a value that already lives in scalar registers
        asm volatile("" : "+s"(v));
        return v;
    }
    // an integer the compiler must keep in a vector register as a value it knows nothing about, in every build: a per-lane
    // bit mask (0 / ~0) applied with v_bfi_b32 stays a vector operand - known to be a function of the lane id it is turned into
    // a scalar-register lane mask + v_cndmask, and the scalar file being full, into two v_readlane per use
    __device__ __forceinline__ int hide(int v) const {
        asm volatile("" : "+v"(v));
        return v;
    }
    // m ? a : b bit by bit, m = 0 or ~0 per lane, the mask a vector operand (hide()).  Written in C the compiler hoists ~m out
    // of the loop and issues and / and / or per half; forced into two v_bfi_b32 by inline assembly it is two instructions fewer
    // and - same box, alternating - 0.6 % SLOWER for a lone wave (29.2 against 29.0 us per iteration, straggler 36.7 against
    // 36.1): the assembly blocks pin the schedule.  The C form stays.
    __device__ __forceinline__ double bit_select(int m, double a, double b) const {
        const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
        const unsigned long long mm = ((unsigned long long)(unsigned)m << 32) | (unsigned)m;
        return __longlong_as_double((long long)((ua & mm) | (ub & ~mm)));
    }
    // issue priority of this wave on its SIMD (s_setprio 0..3): a wave that has run many iterations is a straggler of its batch
    template <int PRIO>      // (s_setprio takes an immediate)
    __device__ __forceinline__ void set_priority() const { __builtin_amdgcn_s_setprio(PRIO); }
    __device__ __forceinline__ int opaque(int v) const {
        if (RELAX & 2) return v;
        asm volatile("" : "+v"(v));
        return v;
    }
    // the same, relaxed only in the build that has its SIMD to itself (bit 2): the 0 / 1 weights the Riccati stage builds its
    // operands with are recomputed per stage in the builds that share a SIMD (ten registers they do not have)
    __device__ __forceinline__ int opaque_shared(int v) const {
        if (RELAX & 4) return v;
        asm volatile("" : "+v"(v));
        return v;
    }
    __device__ __forceinline__ int wave_bcast(PerLane<int> &p, int lane) const {
        return __builtin_amdgcn_readlane(p.v, lane);
    }
    // bit l = lane l's value is non-zero
    __device__ __forceinline__ unsigned long long ballot(PerLane<int> &p) const { return __ballot(p.v != 0); }
    // wave reductions: DPP inside the 16-lane rows, then the four row results (read with v_readlane) are combined
    // as (r0 op r1) op (r2 op r3) identically in every lane
    template <class OP>
    __device__ __forceinline__ double reduce(double v, OP op) const {
        v = row_reduce(v, op);
        return op(op(readlane_f64(v, 0), readlane_f64(v, 16)), op(readlane_f64(v, 32), readlane_f64(v, 48)));
    }
    __device__ __forceinline__ double wave_sum(PerLane<double> &p) const {
        return reduce(p.v, [](double a, double b) { return a + b; });
    }
    __device__ __forceinline__ double wave_max(PerLane<double> &p) const {
        return reduce(p.v, [](double a, double b) { return fmax2(a, b); });
    }
    __device__ __forceinline__ double wave_min(PerLane<double> &p) const {
        return reduce(p.v, [](double a, double b) { return fmin2(a, b); });
    }
    // inclusive suffix sum over the lanes: log-step shifts inside each 16-lane row (DPP), then the totals of the rows
    // above (row r's total sits in its lane 16 r after the row scan) are added
    __device__ __forceinline__ void wave_suffix_sum(PerLane<double> &p) const {
        double v = p.v;
        v += dpp_f64<kDppRowShl1>(v);
        v += dpp_f64<kDppRowShl2>(v);
        v += dpp_f64<kDppRowShl4>(v);
        v += dpp_f64<kDppRowShl8>(v);
        const double t1 = readlane_f64(v, 16), t2 = readlane_f64(v, 32), t3 = readlane_f64(v, 48);
        const int row = (int)threadIdx.x >> 4;
        const double above = row == 0 ? (t1 + (t2 + t3)) : (row == 1 ? (t2 + t3) : (row == 2 ? t3 : 0.0));
        p.v = v + above;
    }
    __device__ __forceinline__ void wave_sum2(PerLane<double> &p, double &lo, double &hi) const {
        const double v = row_reduce(p.v, [](double a, double b) { return a + b; });
        lo = readlane_f64(v, 0) + readlane_f64(v, 16);
        hi = readlane_f64(v, 32) + readlane_f64(v, 48);
    }
    __device__ __forceinline__ void wave_max_ratio(PerLane<double> &pn, PerLane<double> &pd, double &rn, double &rd) const {
        double n = pn.v, d = pd.v;
#define MPC_RATIO_STEP(CTRL)                                                      \
    {                                                                             \
        const double n2 = dpp_f64<CTRL>(n), d2 = dpp_f64<CTRL>(d);                \
        const bool t = ratio_greater(n2, d2, n, d);                               \
        n = t ? n2 : n;                                                           \
        d = t ? d2 : d;                                                           \
    }
        MPC_RATIO_STEP(kDppXor1)
        MPC_RATIO_STEP(kDppXor2)
        MPC_RATIO_STEP(kDppHalfMirror)
        MPC_RATIO_STEP(kDppMirror)
#undef MPC_RATIO_STEP
        double bn = readlane_f64(n, 0), bd = readlane_f64(d, 0);
#pragma unroll
        for (int r = 16; r < 64; r += 16) {
            const double n2 = readlane_f64(n, r), d2 = readlane_f64(d, r);
            const bool t = ratio_greater(n2, d2, bn, bd);
            bn = t ? n2 : bn;
            bd = t ? d2 : bd;
        }
        rn = bn;
        rd = bd;
    }
};
using WaveOps = WaveOpsT<0>;

}  // namespace wave
}  // namespace mpc

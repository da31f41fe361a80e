// mpc_engine.hip - gfx950 kernels and the C ABI (include/mpc_mi355x.h) of the batched MPC solve engine.
//
// Data layout in HBM (per handle, sized for the largest batch seen):
//   work   [STAGE_SLOTS][N+1][Bp] double   per-stage solver state, instance index fastest (Bp = B rounded
//                                          up to 64): a wave's 64 lanes read/write 512 contiguous bytes
//   oth    [V][4][Bp]            double    other vehicles: x, y, per-stage displacement dx, dy
//   ref    [M][6]                double    reference path x, y, v, heading, sin(heading), cos(heading);
//                                          staged into LDS once per workgroup (gathered by ego_index + k)
// One wave64 lane solves one instance start to finish (mpc_core.hpp); a workgroup is one wave so that the
// 256 CUs / 8 XCDs are covered as soon as B >= 16384 and nothing is ever exchanged between lanes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mpc_mi355x.h"
#include "mpc_core.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct GlobalWS {
    double *base;         // work + lane
    const double *obase;  // oth + lane
    size_t Bp;
    int N1;
    __device__ __forceinline__ double ld(int slot, int k) const { return base[((size_t)slot * N1 + k) * Bp]; }
    __device__ __forceinline__ void st(int slot, int k, double v) { base[((size_t)slot * N1 + k) * Bp] = v; }
    __device__ __forceinline__ double oth(int j, int c) const { return obase[((size_t)j * 4 + c) * Bp]; }
};

constexpr int kBlock = 64;  // one wave64 per workgroup

__global__ __launch_bounds__(kBlock) void mpc_solve_kernel(
    mpc::SolveParams P, int B, size_t Bp, double *__restrict__ work, double *__restrict__ oth,
    const double *__restrict__ ref6, int M, const double *__restrict__ state, const int32_t *__restrict__ ego_index,
    const double *__restrict__ vref, const double *__restrict__ weights, const uint8_t *__restrict__ is_collide,
    const double *__restrict__ others, int Vin, double w_collision, double *__restrict__ u0_out,
    double *__restrict__ U_out, double *__restrict__ X_out, int32_t *__restrict__ status_out,
    int32_t *__restrict__ iters_out) {
    extern __shared__ double s_ref[];  // [M][6]
    for (int i = threadIdx.x; i < M * 6; i += kBlock) s_ref[i] = ref6[i];
    __syncthreads();
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;

    const int N = P.N;
    GlobalWS w{work + b, oth + b, Bp, N + 1};
    double x0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x0[i] = state[(size_t)b * 4 + i];
    const int e0 = ego_index[b];
    for (int k = 0; k <= N; ++k) {
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        const double *r = s_ref + idx * 6;
        w.st(mpc::S_REF + 0, k, r[0]);
        w.st(mpc::S_REF + 1, k, r[1]);
        w.st(mpc::S_REF + 2, k, vref ? vref[(size_t)b * (N + 1) + k] : r[2]);
        w.st(mpc::S_REF + 3, k, r[3]);
        w.st(mpc::S_REF + 4, k, r[4]);
        w.st(mpc::S_REF + 5, k, r[5]);
    }
    const bool collide = is_collide[b] != 0;
    if (P.collision_cost) {
        for (int j = 0; j < P.V; ++j) {
            const double *ov = others + ((size_t)b * Vin + j) * 4;
            const double sp = ov[2] * P.dt, hh = ov[3];
            oth[((size_t)j * 4 + 0) * Bp + b] = ov[0];
            oth[((size_t)j * 4 + 1) * Bp + b] = ov[1];
            oth[((size_t)j * 4 + 2) * Bp + b] = sp * cos(hh);
            oth[((size_t)j * 4 + 3) * Bp + b] = sp * sin(hh);
        }
    }
    const double ws_ = collide ? 100.0 : weights[(size_t)b * 3 + 0];  // agents/pure_mpc.py:143-147
    const double wc_ = weights[(size_t)b * 3 + 1], wd_ = weights[(size_t)b * 3 + 2];
    const double wcoll = (P.collision_cost && collide) ? 3000.0 * w_collision : 0.0;

    int status, iters, cur;
    double kkt;
    mpc::solve_instance(P, w, x0, ws_, wc_, wd_, wcoll, status, iters, cur, kkt);

    const int CB = cur * mpc::BUF_SLOTS;
    u0_out[(size_t)b * 2 + 0] = w.ld(CB + mpc::B_U + 0, 0);
    u0_out[(size_t)b * 2 + 1] = w.ld(CB + mpc::B_U + 1, 0);
    if (U_out)
        for (int k = 0; k < N; ++k) {
            U_out[((size_t)b * N + k) * 2 + 0] = w.ld(CB + mpc::B_U + 0, k);
            U_out[((size_t)b * N + k) * 2 + 1] = w.ld(CB + mpc::B_U + 1, k);
        }
    if (X_out)
        for (int k = 0; k <= N; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) X_out[((size_t)b * (N + 1) + k) * 4 + i] = w.ld(CB + mpc::B_X + i, k);
    if (status_out) status_out[b] = status;
    if (iters_out) iters_out[b] = iters;
}

}  // namespace

struct mpc_handle {
    mpc_config cfg;
    int device = 0;
    double *d_ref = nullptr;  // [M][6]
    int M = 0;
    double *d_work = nullptr;
    size_t work_doubles = 0;
    double *d_oth = nullptr;
    size_t oth_doubles = 0;
    // staging buffers for host-pointer calls
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
};

namespace {

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

int ensure_buffer(double **p, size_t *have, size_t need) {
    if (*have >= need) return MPC_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), need * sizeof(double)));
    *have = need;
    return MPC_OK;
}

}  // namespace

extern "C" {

int mpc_version(void) { return MPC_ABI_VERSION; }

const char *mpc_last_error(void) { return g_last_error.c_str(); }

void mpc_default_config(mpc_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(mpc_config);
    cfg->horizon = 20;
    cfg->dt = 0.1;
    cfg->max_iter = 100;
    cfg->device = 0;
    cfg->tol = 1e-8;
    cfg->w_distance = 10.0;
    cfg->w_collision = 1.0;
}

int mpc_create(const mpc_config *cfg, mpc_handle **out) {
    if (!cfg || !out) return fail(MPC_ERR_INVALID_ARG, "mpc_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(mpc_config))
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: mpc_config.struct_size mismatch");
    if (cfg->horizon < 1 || cfg->horizon > MPC_MAX_HORIZON)
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: horizon out of range");
    if (!(cfg->dt > 0.0) || cfg->max_iter < 0 || !(cfg->tol > 0.0))
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: dt, max_iter and tol must be positive");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MPC_ERR_NO_DEVICE,
                    std::string("mpc_create: no HIP device available (the engine has no CPU path): ") +
                        (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (cfg->device < 0 || cfg->device >= count) return fail(MPC_ERR_INVALID_ARG, "mpc_create: bad device ordinal");
    HIP_TRY(hipSetDevice(cfg->device));
    mpc_handle *h = new (std::nothrow) mpc_handle();
    if (!h) return fail(MPC_ERR_HIP, "mpc_create: out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    *out = h;
    g_last_error.clear();
    return MPC_OK;
}

void mpc_destroy(mpc_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_ref) (void)hipFree(h->d_ref);
    if (h->d_work) (void)hipFree(h->d_work);
    if (h->d_oth) (void)hipFree(h->d_oth);
    if (h->d_stage) (void)hipFree(h->d_stage);
    delete h;
}

int mpc_set_reference(mpc_handle *h, const double *ref, int32_t M) {
    if (!h || !ref || M < 1 || M > 4096) return fail(MPC_ERR_INVALID_ARG, "mpc_set_reference: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    std::string buf((size_t)M * 6 * sizeof(double), '\0');
    double *r6 = reinterpret_cast<double *>(&buf[0]);
    for (int i = 0; i < M; ++i) {
        r6[i * 6 + 0] = ref[i * 4 + 0];
        r6[i * 6 + 1] = ref[i * 4 + 1];
        r6[i * 6 + 2] = ref[i * 4 + 2];
        r6[i * 6 + 3] = ref[i * 4 + 3];
        r6[i * 6 + 4] = sin(ref[i * 4 + 3]);
        r6[i * 6 + 5] = cos(ref[i * 4 + 3]);
    }
    if (h->d_ref) HIP_TRY(hipFree(h->d_ref));
    h->d_ref = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_ref), buf.size()));
    HIP_TRY(hipMemcpy(h->d_ref, r6, buf.size(), hipMemcpyHostToDevice));
    h->M = M;
    return MPC_OK;
}

int64_t mpc_workspace_bytes(const mpc_handle *h, int32_t B, int32_t V) {
    if (!h || B < 0 || V < 0) return -1;
    const size_t Bp = round_up((size_t)(B > 0 ? B : 1), kBlock);
    const size_t N1 = (size_t)h->cfg.horizon + 1;
    return (int64_t)(((size_t)mpc::STAGE_SLOTS * N1 + (size_t)V * 4) * Bp * sizeof(double));
}

int mpc_solve_batch(mpc_handle *h, int32_t B, const double *state, const int32_t *ego_index, const double *vref,
                    const double *weights, const uint8_t *is_collide, const double *others, int32_t V,
                    uint32_t flags, double *u0, double *U, double *X, int32_t *status, int32_t *iters,
                    void *stream_) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: null handle");
    if (B < 0 || !state || !ego_index || !weights || !is_collide || !u0)
        return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: null required pointer or negative batch");
    if (V < 0 || V > MPC_MAX_OTHERS) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: V out of range");
    const bool cc = (flags & MPC_FLAG_COLLISION_COST) != 0;
    if (cc && V > 0 && !others) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: collision cost needs `others`");
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_solve_batch: call mpc_set_reference first");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int N = h->cfg.horizon;
    const size_t N1 = (size_t)N + 1;
    const size_t Bp = round_up((size_t)B, kBlock);
    const int Vuse = cc ? V : 0;
    int rc = ensure_buffer(&h->d_work, &h->work_doubles, (size_t)mpc::STAGE_SLOTS * N1 * Bp);
    if (rc) return rc;
    rc = ensure_buffer(&h->d_oth, &h->oth_doubles, (size_t)(Vuse > 0 ? Vuse : 1) * 4 * Bp);
    if (rc) return rc;

    // ---- device views of the arguments
    const double *d_state = state, *d_vref = vref, *d_weights = weights, *d_others = others;
    const int32_t *d_ego = ego_index;
    const uint8_t *d_coll = is_collide;
    double *d_u0 = u0, *d_U = U, *d_X = X;
    int32_t *d_status = status, *d_iters = iters;
    const bool dev = (flags & MPC_FLAG_DEVICE_PTRS) != 0;
    size_t off_u0 = 0, off_U = 0, off_X = 0, off_st = 0, off_it = 0;
    if (!dev) {
        // pack everything into one staging allocation (8-byte aligned segments)
        size_t off = 0;
        auto seg = [&](size_t bytes) {
            size_t o = off;
            off += round_up(bytes, 16);
            return o;
        };
        const size_t o_state = seg((size_t)B * 4 * 8), o_ego = seg((size_t)B * 4), o_w = seg((size_t)B * 3 * 8);
        const size_t o_c = seg((size_t)B), o_vref = vref ? seg((size_t)B * N1 * 8) : 0;
        const size_t o_oth = (cc && V > 0) ? seg((size_t)B * V * 4 * 8) : 0;
        off_u0 = seg((size_t)B * 2 * 8);
        off_U = U ? seg((size_t)B * N * 2 * 8) : 0;
        off_X = X ? seg((size_t)B * N1 * 4 * 8) : 0;
        off_st = seg((size_t)B * 4);
        off_it = seg((size_t)B * 4);
        if (h->stage_bytes < off) {
            if (h->d_stage) HIP_TRY(hipFree(h->d_stage));
            h->d_stage = nullptr;
            h->stage_bytes = 0;
            HIP_TRY(hipMalloc(&h->d_stage, off));
            h->stage_bytes = off;
        }
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(sb + o_state, state, (size_t)B * 4 * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_ego, ego_index, (size_t)B * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_w, weights, (size_t)B * 3 * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_c, is_collide, (size_t)B, hipMemcpyHostToDevice, stream));
        d_state = reinterpret_cast<double *>(sb + o_state);
        d_ego = reinterpret_cast<int32_t *>(sb + o_ego);
        d_weights = reinterpret_cast<double *>(sb + o_w);
        d_coll = reinterpret_cast<uint8_t *>(sb + o_c);
        if (vref) {
            HIP_TRY(hipMemcpyAsync(sb + o_vref, vref, (size_t)B * N1 * 8, hipMemcpyHostToDevice, stream));
            d_vref = reinterpret_cast<double *>(sb + o_vref);
        }
        if (cc && V > 0) {
            HIP_TRY(hipMemcpyAsync(sb + o_oth, others, (size_t)B * V * 4 * 8, hipMemcpyHostToDevice, stream));
            d_others = reinterpret_cast<double *>(sb + o_oth);
        }
        d_u0 = reinterpret_cast<double *>(sb + off_u0);
        d_U = U ? reinterpret_cast<double *>(sb + off_U) : nullptr;
        d_X = X ? reinterpret_cast<double *>(sb + off_X) : nullptr;
        d_status = reinterpret_cast<int32_t *>(sb + off_st);
        d_iters = reinterpret_cast<int32_t *>(sb + off_it);
    }

    mpc::SolveParams P;
    P.N = N;
    P.V = Vuse;
    P.max_iter = h->cfg.max_iter;
    P.collision_cost = cc ? 1 : 0;
    P.dt = h->cfg.dt;
    P.tol = h->cfg.tol;
    P.mu_init = 0.1;
    P.w_distance = h->cfg.w_distance;

    const unsigned grid = (unsigned)(Bp / kBlock);
    const size_t lds = (size_t)h->M * 6 * sizeof(double);
    hipLaunchKernelGGL(mpc_solve_kernel, dim3(grid), dim3(kBlock), lds, stream, P, (int)B, Bp, h->d_work, h->d_oth,
                       h->d_ref, h->M, d_state, d_ego, d_vref, d_weights, d_coll, d_others, (int)V,
                       h->cfg.w_collision, d_u0, d_U, d_X, d_status, d_iters);
    HIP_TRY(hipGetLastError());

    if (!dev) {
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(u0, sb + off_u0, (size_t)B * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (U) HIP_TRY(hipMemcpyAsync(U, sb + off_U, (size_t)B * N * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (X) HIP_TRY(hipMemcpyAsync(X, sb + off_X, (size_t)B * N1 * 4 * 8, hipMemcpyDeviceToHost, stream));
        if (status) HIP_TRY(hipMemcpyAsync(status, sb + off_st, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        if (iters) HIP_TRY(hipMemcpyAsync(iters, sb + off_it, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    } else if (!(flags & MPC_FLAG_NO_SYNC)) {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return MPC_OK;
}

}  // extern "C"

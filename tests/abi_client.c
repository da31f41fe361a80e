/* A plain C client of the C ABI (include/mpc_mi355x.h), for tests only: no Python, no torch - what a binding in any
 * host language sees.  Usage: abi_client <libmpc_mi355x.so> [solve]
 *   without "solve": loads the library, checks version / default configuration / argument errors and what mpc_create
 *                    answers on this machine (0 with a GPU, MPC_ERR_NO_DEVICE without), prints "create rc=<rc>"
 *   with "solve"   : needs a GPU: the known-answer instances of tests/test_parity_gpu.py::test_known_answers through
 *                    mpc_solve_batch with host pointers, and the iterative-linear QP of an ego on the path.
 * Exit code 0 = every check passed. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mpc_mi355x.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #cond); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); CHECK(p_##name != NULL)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    SYM(mpc_version); SYM(mpc_last_error); SYM(mpc_default_config); SYM(mpc_create); SYM(mpc_destroy);
    SYM(mpc_set_reference); SYM(mpc_solve_batch); SYM(mpc_ltv_solve_batch); SYM(mpc_workspace_bytes);
    CHECK(p_mpc_version() == MPC_ABI_VERSION);
    mpc_config cfg;
    p_mpc_default_config(&cfg);
    CHECK(cfg.struct_size == (int32_t)sizeof(mpc_config) && cfg.horizon == 20 && cfg.max_iter == 100);
    mpc_handle *h = NULL;
    cfg.horizon = MPC_MAX_HORIZON + 1;
    CHECK(p_mpc_create(&cfg, &h) == MPC_ERR_INVALID_ARG && strlen(p_mpc_last_error()) > 0);
    cfg.horizon = 20;
    const int rc = p_mpc_create(&cfg, &h);
    printf("create rc=%d\n", rc);
    if (rc != MPC_OK) {
        CHECK(rc == MPC_ERR_NO_DEVICE && h == NULL && strstr(p_mpc_last_error(), "no HIP device") != NULL);
        return (argc > 2) ? 1 : 0;          /* "solve" needs the GPU */
    }
    if (argc > 2 && strcmp(argv[2], "solve") == 0) {
        /* reference path of agents/base_agent.py:118-154 (x, y, v, heading), 85 points */
        enum { M = 85, N = 20 };
        static double ref[M][4];
        {
            double x = 2.0, y = 50.0, heading = -M_PI / 2;
            const double v = 10.0, dt = 0.1;
            int n = 0;
            for (int i = 0; i < 40; ++i, ++n) {
                y += v * dt * sin(heading);
                ref[n][0] = x; ref[n][1] = y; ref[n][2] = v; ref[n][3] = heading;
            }
            for (int i = 0; i < 20; ++i, ++n) {
                heading -= (M_PI / 2) / 20;
                x += v * dt * cos(heading);
                y += v * dt * sin(heading);
                ref[n][0] = x; ref[n][1] = y; ref[n][2] = v; ref[n][3] = heading;
            }
            for (int i = 0; i < 25; ++i, ++n) {
                x += v * dt * cos(heading);
                ref[n][0] = x; ref[n][1] = y; ref[n][2] = v; ref[n][3] = heading;
            }
        }
        double state[2][4] = {{2.0, 45.0, -M_PI / 2, 10.0}, {2.0, 45.0, -M_PI / 2, 31.0}};
        int32_t ego[2] = {4, 4}, status[2], iters[2];
        double weights[2][3] = {{1, 1, 1}, {1, 1, 1}}, u0[2][2], U[2][N][2];
        uint8_t collide[2] = {0, 0};
        CHECK(p_mpc_solve_batch(h, 2, &state[0][0], ego, NULL, &weights[0][0], collide, NULL, 0, 0, &u0[0][0], &U[0][0][0],
                                NULL, status, iters, NULL) == MPC_ERR_NO_REFERENCE);
        CHECK(p_mpc_set_reference(h, &ref[0][0], M) == MPC_OK);
        CHECK(p_mpc_solve_batch(h, 2, &state[0][0], ego, NULL, &weights[0][0], collide, NULL, 0, 0, &u0[0][0], &U[0][0][0],
                                NULL, status, iters, NULL) == MPC_OK);
        /* on the straight part of the path at the reference speed: the optimal controls are zero (the straight
         * part does not depend on how the arc is discretised); a speed of 31 m/s violates the state bounds */
        CHECK(status[0] == MPC_STATUS_CONVERGED && fabs(u0[0][0]) < 1e-6 && fabs(u0[0][1]) < 1e-6);
        CHECK(status[1] == MPC_STATUS_INFEASIBLE_START);
        /* iterative-linear agent: same ego, state order (x, y, v, yaw); zero stored profile */
        double st2[1][4] = {{2.0, 45.0, 10.0, -M_PI / 2}}, u2[1][2], U2[1][N][2];
        int32_t s2[1], i2[1], t2[1];
        memset(U2, 0, sizeof(U2));
        CHECK(p_mpc_ltv_solve_batch(h, 1, &st2[0][0], 0, &u2[0][0], &U2[0][0][0], NULL, s2, i2, t2, NULL) == MPC_OK);
        /* (the reference's linear model has no affine term, so even on the path the QP steers: no zero answer here) */
        CHECK(s2[0] == MPC_STATUS_CONVERGED && t2[0] == 4 && u2[0][0] == U2[0][0][0] && u2[0][1] == U2[0][0][1]);
        CHECK(u2[0][0] >= -5.0 - 1e-9 && u2[0][0] <= 2.0 + 1e-9 && fabs(u2[0][1]) <= M_PI / 6 + 1e-9 && i2[0] > 3 && i2[0] < 40);
        CHECK(p_mpc_workspace_bytes(h, 1, 0) > 0);
        printf("solve ok: u0 = (%.3e, %.3e), ltv u0 = (%.3e, %.3e) in %d iterations\n", u0[0][0], u0[0][1], u2[0][0], u2[0][1],
               i2[0]);
    }
    p_mpc_destroy(h);
    return 0;
}

/* c_api_demo.c -- the C ABI (include/nmpc_solver.h) from plain C: no Python, no torch, no HIP headers.
 *
 *   gcc -O2 -I include examples/c_api_demo.c -o c_api_demo \
 *       -L mpc_trajectory_generator_amd/csrc -lnmpc_hip -Wl,-rpath,$PWD/mpc_trajectory_generator_amd/csrc -lm
 *   ./c_api_demo [p.bin]        (p.bin: B x 430 doubles; default: one hand-made default.yaml-shaped problem)
 *
 * Solves a small batch through nmpc_solve_batch_host (what replaces mng.call(parameters),
 * reference src/mpc/mpc_generator.py:206) and prints OpEn-style status lines. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nmpc_solver.h"

static const char *EXIT[] = {"Converged", "NotConvergedIterations", "NotConvergedOutOfTime", "NotConvergedCost",
                             "NotConvergedNotFiniteComputation"};

int main(int argc, char **argv)
{
    /* what MpcModule.build() bakes into the generated solver (configs/default.yaml:7-13,18,34-40) */
    nmpc_problem pb = {.N = 20, .nobs = 10, .ndyn = 3, .reserved = 0, .ts = 0.2, .vmin = -0.5, .vmax = 1.5, .wmax = 0.5,
                       .amin = -1.0, .amax = 1.0, .awmax = 3.0};
    nmpc_opts op;
    nmpc_default_opts(&op);
    const int n_p = nmpc_n_p(&pb), n_u = nmpc_n_u(&pb), n1 = nmpc_n1(&pb);
    int B = 1;
    double *p = NULL;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { perror(argv[1]); return 2; }
        fseek(f, 0, SEEK_END);
        const long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        B = (int)(bytes / (8L * n_p));
        p = malloc((size_t)B * n_p * 8);
        if (fread(p, 8, (size_t)B * n_p, f) != (size_t)B * n_p) { fprintf(stderr, "short read\n"); return 2; }
        fclose(f);
    } else {
        /* a straight reference along x at y = 1 walked at 1 m/s, robot 5 cm off it and already moving at
         * 1 m/s, no obstacles near (parameter layout: reference src/path_generator.py:378-379) */
        p = calloc((size_t)n_p, 8);
        const double w[10] = {0, 10, 0, 0, 0, 0, 0, 200, 10, 5};      /* configs/default.yaml:22-31 */
        p[0] = 0.0; p[1] = 1.05; p[2] = 0.0;                          /* x, y, theta */
        p[3] = 1.0; p[8] = 1.0;                                       /* last applied v (twice, :378) */
        p[5] = 20 * 0.33; p[6] = 1.0; p[7] = 0.0;                     /* x_finish */
        memcpy(p + 10, w, sizeof w);
        for (int t = 0; t < 20; ++t) p[20 + t] = 1.0;                 /* vel_ref */
        for (int k = 0; k < 3 * 20; ++k) { double *e = p + 70 + 5 * k; e[2] = 1.0; e[3] = 1.0; }   /* dyn padding */
        for (int t = 0; t < 20; ++t) { p[370 + 3 * t] = 0.22 * t; p[370 + 3 * t + 1] = 1.0; }      /* reference samples */
    }
    nmpc_handle *h = NULL;
    int rc = nmpc_new(&pb, &op, 0, B, &h);
    if (rc) { fprintf(stderr, "nmpc_new failed: %d (this library needs a HIP device)\n", rc); return 1; }
    printf("ABI %d, kernel %s, B = %d, n_p = %d, n_u = %d\n", nmpc_abi_version(), nmpc_kernel_name(h), B, n_p, n_u);
    double *u = calloc((size_t)B * n_u, 8), *y = calloc((size_t)B * n1, 8);
    nmpc_status *st = calloc((size_t)B, sizeof *st);
    rc = nmpc_solve_batch_host(h, B, p, u, NULL, NULL, y, st);
    if (rc) { fprintf(stderr, "solve failed: %d %s\n", rc, nmpc_last_error(h)); return 1; }
    for (int b = 0; b < B && b < 8; ++b)
        printf("instance %d: %s, %u outer / %u inner iterations, fpr %.2e, ||F2|| %.2e, u[0:2] = (%.6f, %.6f)\n", b,
               EXIT[st[b].exit_status], st[b].num_outer_iterations, st[b].num_inner_iterations,
               st[b].last_problem_norm_fpr, st[b].f2_norm, u[(size_t)b * n_u], u[(size_t)b * n_u + 1]);
    printf("kernel time of the batch %.2f ms, instance 0 alone %.3f ms\n", nmpc_last_batch_ms(h), st[0].solve_time_ms);
    /* warm start from the solution and multipliers: the second call of a receding-horizon loop */
    rc = nmpc_solve_batch_host(h, B, p, u, y, NULL, y, st);
    if (rc) return 1;
    printf("warm restart: instance 0 %s after %u inner iterations\n", EXIT[st[0].exit_status], st[0].num_inner_iterations);
    nmpc_free(h);
    free(p); free(u); free(y); free(st);
    return 0;
}

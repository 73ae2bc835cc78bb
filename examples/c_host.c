/* A C host of the C-ABI (include/ev2g.h), no Python involved: draw scenarios with the library's generator, load them as a
 * resident pool, run whole episodes of the fused step kernel with a constant action (the reference's ChargeAsFastAsPossible,
 * heuristics.py:152-166), read the episode statistics.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_host.c -Lev2gym_amd -lev2g_hip -Wl,-rpath,$PWD/ev2gym_amd -o c_host
 *   ./c_host [--generate-only] [n_envs] [n_episodes]
 * --generate-only stops after the (host-side) scenario generation: it runs on a box without a GPU. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ev2g.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != EV2G_OK) {                                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ev2g_last_error(h));                       \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char **argv) {
    ev2g_handle *h = NULL;
    int generate_only = 0, n_envs = 256, n_episodes = 3, pool_factor = 4, a = 1;
    if (a < argc && strcmp(argv[a], "--generate-only") == 0) { generate_only = 1; a++; }
    if (a < argc) n_envs = atoi(argv[a++]);
    if (a < argc) n_episodes = atoi(argv[a++]);

    /* 1. scenarios: V2GProfitPlusLoads.yaml's values, 50 chargers, pool_factor x n_envs scenarios */
    ev2g_gen_config gc;
    CHECK(ev2g_gen_default_config(0, &gc));
    gc.number_of_charging_stations = 50;
    ev2g_gen_result *gen = NULL;
    CHECK(ev2g_generate(&gc, n_envs * pool_factor, 2024u, 0, &gen));
    const ev2g_scenario_batch *b = ev2g_gen_batch(gen);
    printf("generated %d scenarios: %d steps, %d chargers, %lld EV sessions (%.2f per port)\n", b->n_envs, b->n_steps, b->n_chargers,
           (long long)b->n_sessions, (double)b->n_sessions / ((double)b->n_envs * b->n_chargers));
    if (generate_only) { ev2g_gen_free(gen); return 0; }

    /* 2. engine: n_envs envs stepped concurrently over the resident pool */
    ev2g_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0;
    cfg.reward_kind = EV2G_REWARD_PROFITMAX_TRPENALTY_USERINCENTIVES;
    cfg.state_kind = EV2G_STATE_V2G_PROFIT_MAX_LOADS;
    cfg.flags = EV2G_FLAG_LOG_SOC;
    cfg.n_active_envs = n_envs;
    CHECK(ev2g_create(&cfg, &h));
    CHECK(ev2g_load_scenarios(h, b));
    ev2g_gen_free(gen); /* the engine copied what it needs */
    const int E = ev2g_n_envs(h), P = ev2g_n_ports(h), D = ev2g_obs_dim(h), T = ev2g_n_steps(h);
    printf("kernel: %s, E=%d P=%d D=%d T=%d\n", ev2g_kernel_name(h), E, P, D, T);

    /* 3. device buffers: one [E,P] action block reused every step (stride 0), outputs overwritten in place */
    double *act = (double *)ev2g_malloc(h, sizeof(double) * E * P), *obs = (double *)ev2g_malloc(h, sizeof(double) * E * D);
    double *rew = (double *)ev2g_malloc(h, sizeof(double) * E), *d_stats = (double *)ev2g_malloc(h, sizeof(double) * E * EV2G_N_STATS);
    uint8_t *done = (uint8_t *)ev2g_malloc(h, E), *mask = (uint8_t *)ev2g_malloc(h, (size_t)E * P);
    double *ones = (double *)malloc(sizeof(double) * E * P), *stats = (double *)malloc(sizeof(double) * E * EV2G_N_STATS);
    if (!act || !obs || !rew || !d_stats || !done || !mask || !ones || !stats) { fprintf(stderr, "allocation failed\n"); return 1; }
    for (int i = 0; i < E * P; i++) ones[i] = 1.0;
    CHECK(ev2g_memcpy_h2d(h, act, ones, sizeof(double) * E * P));

    /* 4. episodes: every reset moves on to the next E scenarios of the pool (EV2Gym.reset() drawing a new scenario) */
    for (int ep = 0; ep < n_episodes; ep++) {
        CHECK(ev2g_reset_ex(h, obs, (int64_t)ep * E));
        CHECK(ev2g_step_n(h, T, EV2G_STEPN_PERSISTENT, act, 0, obs, 0, rew, 0, done, 0, mask, 0, 0));
        CHECK(ev2g_check_faults(h, NULL));
        CHECK(ev2g_get_stats(h, d_stats));
        CHECK(ev2g_memcpy_d2h(h, stats, d_stats, sizeof(double) * E * EV2G_N_STATS));
        double served = 0, profit = 0, sat = 0;
        for (int e = 0; e < E; e++) { served += stats[e * EV2G_N_STATS + 0]; profit += stats[e * EV2G_N_STATS + 1]; sat += stats[e * EV2G_N_STATS + 4]; }
        printf("episode %d: %.1f EVs served per env, mean %s %.2f, mean %s %.3f, step kernel %.3f ms\n", ep, served / E, ev2g_stat_name(1), profit / E,
               ev2g_stat_name(4), sat / E, ev2g_last_step_n_kernel_ms(h));
    }
    ev2g_free(h, act); ev2g_free(h, obs); ev2g_free(h, rew); ev2g_free(h, d_stats); ev2g_free(h, done); ev2g_free(h, mask);
    free(ones); free(stats);
    ev2g_destroy(h);
    return 0;
}

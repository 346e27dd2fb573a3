/* ll_work_stats.c — diagnostic: work distribution of LunarLander's contact path, measured on the CPU oracle
 * built with -DORA_STATS (counters in oracle/b2mini.c).  Drives a population like bench.py does (random actions,
 * wind on, masked reset of finished episodes) and prints per-step histograms for the envs that hold contacts:
 * existing / touching contacts, position iterations, time-of-impact calls, TOI sub-steps.
 *   gcc -O2 -std=c99 -DORA_STATS -ffp-contract=off -I oracle tools/ll_work_stats.c oracle/{b2mini,lunar_lander,classic_control,rng,vec}.c -lm -o /tmp/ll_work_stats
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct {
    long steps, steps_with_contacts;
    long hist_contacts[16], hist_touching[16], hist_pos_iters[64];
    long toi_calls, toi_outer_iters, toi_root_iters, toi_pushback, gjk_calls, gjk_iters;
    long hist_toi_calls[64], hist_substeps[32], hist_rejected[32];
    long toi_island_contacts[16], toi_pos_iters;
    int cur_toi_calls, cur_substeps, cur_rejected;
} ora_b2_stats;
extern ora_b2_stats g_b2_stats;

static void hist(const char *name, const long *h, int n) {
    long tot = 0;
    for (int i = 0; i < n; ++i) tot += h[i];
    printf("%-22s total=%ld :", name, tot);
    for (int i = 0; i < n; ++i) if (h[i]) printf(" %d:%.4f", i, (double)h[i] / (double)(tot ? tot : 1));
    printf("\n");
}

int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 2048, steps = argc > 2 ? atol(argv[2]) : 600, warm = argc > 3 ? atol(argv[3]) : 400;
    ora_vec_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.kind = 3; cfg.n_envs = (uint64_t)n; cfg.seed = 0x5EED0004; cfg.gravity = -10.0f; cfg.enable_wind = 1;
    cfg.wind_power = 15.0f; cfg.turbulence_power = 1.5f; cfg.is_euler = 1;
    int st;
    ora_vec *v = ora_vec_new(&cfg, &st);
    if (!v) { fprintf(stderr, "new failed %d\n", st); return 1; }
    ora_vec_reset(v, NULL, NULL, 1);
    uint32_t *acts = (uint32_t *)malloc((size_t)16 * n * 4);
    srand(1);
    for (long i = 0; i < 16 * n; ++i) acts[i] = (uint32_t)(rand() & 3);
    ora_vec_run(v, acts, 16, warm, 1);
    memset(&g_b2_stats, 0, sizeof g_b2_stats);
    long fin = ora_vec_run(v, acts, 16, steps, 1);
    const ora_b2_stats *s = &g_b2_stats;
    printf("envs=%ld steps=%ld world_steps=%ld (incl. implicit step(0) of resets) with_contacts=%ld (%.3f) episodes_finished=%ld\n", n, steps, s->steps,
           s->steps_with_contacts, (double)s->steps_with_contacts / (double)s->steps, fin);
    hist("existing contacts", s->hist_contacts, 16);
    hist("touching (island)", s->hist_touching, 16);
    hist("island pos iters", s->hist_pos_iters, 64);
    hist("toi calls / step", s->hist_toi_calls, 64);
    hist("toi substeps / step", s->hist_substeps, 32);
    hist("toi rejected / step", s->hist_rejected, 32);
    hist("toi island contacts", s->toi_island_contacts, 16);
    printf("toi_calls=%ld outer_iters/call=%.2f root_iters/call=%.2f pushback/call=%.2f toi_pos_iters/substep=%.2f\n", s->toi_calls,
           (double)s->toi_outer_iters / (double)(s->toi_calls ? s->toi_calls : 1), (double)s->toi_root_iters / (double)(s->toi_calls ? s->toi_calls : 1),
           (double)s->toi_pushback / (double)(s->toi_calls ? s->toi_calls : 1),
           (double)s->toi_pos_iters / (double)(s->hist_substeps[0] >= 0 ? 1 : 1) );
    return 0;
}

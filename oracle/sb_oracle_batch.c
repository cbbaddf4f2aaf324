/*
 * sb_oracle_batch.c -- runs the single-building oracle over many independent buildings
 * (TEST INFRASTRUCTURE: used for in-run parity of the first few buildings and as the
 * "port" CPU baseline timed by bench.py; OpenMP over buildings, one state per building).
 */
#include "sb_oracle.h"

#include <omp.h>

int32_t sbo_max_threads(void) { return omp_get_max_threads(); }

/* Steps buildings [0, nb) once.  states/outs are arrays of per-building structs; `ins`
 * is either one shared sbo_step_in (per_building_in == 0) or an array of nb of them. */
void sbo_step_batch(const sbo_plan *p, const sbo_params *prm, sbo_state *states,
                    const sbo_step_in *ins, int32_t per_building_in, sbo_step_out *outs,
                    int32_t nb, int32_t n_threads) {
  if (n_threads <= 1) {
    for (int32_t b = 0; b < nb; b++)
      sbo_step(p, prm, &states[b], per_building_in ? &ins[b] : ins, &outs[b]);
    return;
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int32_t b = 0; b < nb; b++)
    sbo_step(p, prm, &states[b], per_building_in ? &ins[b] : ins, &outs[b]);
}

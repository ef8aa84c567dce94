// Host side of training dropout: turns the public bd_dropout into the kernel argument block.
#pragma once
#include "bd_rng.h"
#include "bidiff.h"

// Null or p == 0 switches dropout off (thr = 0, scale = 1). False when p is outside [0, 1).
static inline bool bd_drop_make(const bd_dropout* d, unsigned site, bd_drop_args* out) {
  out->thr = 0u; out->k0 = 0u; out->k1 = 0u; out->site = site; out->step = 0u; out->scale = 1.0f;
  if (d == nullptr || d->p == 0.0f) return true;
  if (!(d->p > 0.0f && d->p < 1.0f)) return false;
  out->thr = (uint32_t)((double)d->p * 65536.0);
  if (out->thr == 0u) return true;      // p below the 16-bit resolution: off
  out->scale = 1.0f / (1.0f - d->p);
  out->k0 = (uint32_t)d->seed;
  out->k1 = (uint32_t)(d->seed >> 32);
  out->step = d->step;
  return true;
}

// Optional HIP-event bracketing of kernel launches (see bd_prof_enable in bidiff.h).
#pragma once
#include <hip/hip_runtime.h>
// kernel families the profiler keeps separate totals for (bd_prof_read_kind)
enum { BD_PROF_GEMM = 0, BD_PROF_PANEL = 1, BD_PROF_FFN = 2, BD_PROF_ATTN_BLOCK = 3, BD_PROF_MEGA = 4, BD_PROF_KINDS = 5 };
bool bd_prof_on();
void* bd_prof_begin(hipStream_t s);
void bd_prof_end(void* token, hipStream_t s, double flops, int kind = BD_PROF_GEMM);

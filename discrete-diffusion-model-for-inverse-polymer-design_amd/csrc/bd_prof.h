// Optional HIP-event bracketing of kernel launches (see bd_prof_enable in bidiff.h).
#pragma once
#include <hip/hip_runtime.h>
bool bd_prof_on();
void* bd_prof_begin(hipStream_t s);
void bd_prof_end(void* token, hipStream_t s, double flops);

// Optional HIP-event bracketing of kernel launches (see bd_prof_enable in bidiff.h).
#pragma once
#include <hip/hip_runtime.h>
// kernel families the profiler keeps separate totals for (bd_prof_read_kind)
enum { BD_PROF_GEMM = 0, BD_PROF_PANEL = 1, BD_PROF_FFN = 2, BD_PROF_ATTN_BLOCK = 3, BD_PROF_MEGA = 4,
       BD_PROF_WGRAD = 5,      // weight-gradient GEMMs (flops)
       BD_PROF_ATTN_FWD = 6,   // stand-alone attention forward (flops)
       BD_PROF_ATTN_BWD = 7,   // attention backward, both kernels (flops)
       BD_PROF_LN = 8,         // LayerNorm forward / backward (bytes)
       BD_PROF_CE = 9,         // masked cross-entropy forward / backward (bytes)
       BD_PROF_EMBED = 10,     // embedding sum / scatter, corruption, dropout, casts (bytes)
       BD_PROF_OPT = 11,       // gradient norm, AdamW, Muon, weight-copy refresh (bytes)
       BD_PROF_SAMPLE = 12,    // reverse step and post-loop repairs (bytes)
       BD_PROF_KINDS = 13 };
bool bd_prof_on();
void* bd_prof_begin(hipStream_t s);
void bd_prof_end(void* token, hipStream_t s, double flops, int kind = BD_PROF_GEMM);

// Brackets everything launched between its construction and destruction (one entry point's launches).
// `work` is flops for the MFMA families and algorithmic bytes for the bandwidth families.
struct bd_prof_scope {
  void* tok; hipStream_t s; double work; int kind;
  bd_prof_scope(void* stream, double work_, int kind_)
      : tok(bd_prof_on() ? bd_prof_begin((hipStream_t)stream) : nullptr), s((hipStream_t)stream), work(work_), kind(kind_) {}
  ~bd_prof_scope() { if (tok) bd_prof_end(tok, s, work, kind); }
  bd_prof_scope(const bd_prof_scope&) = delete;
  bd_prof_scope& operator=(const bd_prof_scope&) = delete;
};

// Token ids -> strings on the device: the join of the non-special tokens' texts, one wavefront per sequence.
// Replaces the per-id dictionary loop of the reference tokenizers' decode (Bi_Diffusion_SMILES
// src/data/tokenizer.py:207-224, detokenize :110-121; the SMILES_BPE and SELFIES tokenizers decode the same way),
// which the samplers call on the whole batch after the last step (src/sampling/sampler.py:724).
#include "bd_common.h"
#include "bd_prof.h"
#include "bidiff.h"

namespace {

__global__ __launch_bounds__(64) void decode_kernel(const long long* __restrict__ ids, int L,
                                                    const unsigned char* __restrict__ text,
                                                    const int* __restrict__ offs, int V,
                                                    unsigned char* __restrict__ out, int* __restrict__ out_len,
                                                    int stride) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const long long* row = ids + (size_t)b * L;
  const int CH = (L + 63) / 64;         // contiguous positions per lane: the output keeps the position order
  const int l0 = lane * CH;
  int mine = 0;
  for (int j = 0; j < CH; ++j) {
    const int l = l0 + j;
    if (l < L) {
      const long long id = row[l];
      if (id >= 0 && id < V) mine += offs[id + 1] - offs[id];   // specials and unknown ids have empty text
    }
  }
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  int at = incl - mine;
  unsigned char* dst = out + (size_t)b * stride;
  for (int j = 0; j < CH; ++j) {
    const int l = l0 + j;
    if (l < L) {
      const long long id = row[l];
      if (id >= 0 && id < V) {
        const int s = offs[id], n = offs[id + 1] - s;
        for (int c = 0; c < n; ++c)
          if (at + c < stride) dst[at + c] = text[s + c];
        at += n;
      }
    }
  }
  if (lane == 0) out_len[b] = total;    // > stride tells the caller that the row was cut
}

}  // namespace

extern "C" int bd_decode_ids(const long long* ids, int B, int L, const unsigned char* text, const int* offsets,
                             int V, unsigned char* out, int* out_len, int stride, void* stream) {
  BD_REQUIRE(ids && text && offsets && out && out_len, "bd_decode_ids: null operand");
  BD_REQUIRE(B >= 0 && L > 0 && V > 0 && stride > 0, "bd_decode_ids: bad shape B=%d L=%d V=%d stride=%d", B, L, V, stride);
  if (B == 0) return BD_OK;
  bd_prof_scope prof(stream, (double)B * (8.0 * L + stride), BD_PROF_SAMPLE);
  hipLaunchKernelGGL(decode_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ids, L, text, offsets, V, out, out_len, stride);
  return bd_check_launch("bd_decode_ids");
}

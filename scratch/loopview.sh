#!/bin/bash
# usage: loopview.sh <asm> <kernel-substring> [from to]: compact instruction sequence
S=$1; K=$2
awk -v k="$K" 'index($0, k) && /^_Z.*:/ {p=1} p {print} p && /s_endpgm/ {exit}' $S > /tmp/kern.s
if [ -z "$3" ]; then
  echo "lines: $(wc -l < /tmp/kern.s)"
  grep -n "^.LBB\|s_barrier\|scratch_" /tmp/kern.s | awk '{print $1,$2}' | tr '\n' ' ' | fold -w 200; echo
else
  sed -n "$3,$4p" /tmp/kern.s | grep -v "^\s*;" | awk '{print $1 ($1=="s_waitcnt" ? "_"$2 : "")}' | sed 's/v_mfma_f32_16x16x32_bf16/MFMA/; s/ds_read_b128/DSR/; s/v_pk_fma_f32/pkfma/; s/s_waitcnt_/W:/; s/v_add_u32_e32/vadd/; s/global_load_lds_dwordx4/GLDS/; s/v_readfirstlane_b32/rfl/' | uniq -c | awk '{ if ($1>1) printf "%dx%s ", $1, $2; else printf "%s ", $2 } END {print ""}' | fold -w 200
fi

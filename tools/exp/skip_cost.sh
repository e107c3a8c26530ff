# What an entry point costs inside the replayed step: the step with it skipped (results garbage) vs the full step, same box.
# Needs the timing-experiment build: make -C mirror_amd/csrc EXP=1 (tools/exp/step_time.py; bench.py refuses MH_EXP_*).
# CAVEAT (measured): only meaningful for entry points whose output does not feed the MFMA kernels — skipping a producer leaves
# zeros / garbage downstream, and GEMMs / attention on zero operands run at higher clocks.
run() { printf '%-90s %s ms\n' "$1" "$(MH_EXP_SKIP=$1 python3 tools/exp/step_time.py 20 2>/dev/null | tail -1)"; }
run none
RNA=mh_rna_block_fwd,mh_rna_block_bwd,mh_skinny_fwd,mh_skinny_wgrad,mh_headattn_fwd,mh_headattn_bwd
for k in $RNA mh_rna_block_fwd,mh_rna_block_bwd mh_skinny_fwd,mh_skinny_wgrad mh_pinv_chain_fwd,mh_pinv_chain_bwd mh_pinv_chain_fwd mh_pinv_chain_bwd mh_gemm mh_nys_attn1_fwd,mh_nys_attn1_bwd mh_nys_attn3_fwd,mh_nys_attn3_bwd mh_adam mh_layernorm_bwd,mh_layernorm_bwd_lm mh_resconv_fwd,mh_resconv_wgrad mh_ppeg_fwd,mh_ppeg_wgrad mh_dropout_lite mh_rank_mask mh_loss_terms_fwd,mh_loss_terms_bwd "$RNA,mh_pinv_chain_fwd,mh_pinv_chain_bwd"; do run $k; done
run none

# What an entry point costs inside the replayed step: bench with it skipped (results garbage) vs the full step, same box.
# CAVEAT (measured): only meaningful for entry points whose output does not feed the MFMA kernels — skipping a producer leaves
# zeros / garbage downstream, and GEMMs / attention on zero operands run at higher clocks (Adam 0.18 ms, colsum 0.11 ms and the
# transposes 0.04 ms are clean readings; "mse_masked costs 0.49 ms" is the whole backward running on zero gradients).
run() { MH_EXP_SKIP=$1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-46s %8.3f ms' % ('$1', d['ms_per_step']))"; }
run none
for k in mh_adam mh_transpose_bf16_many mh_layernorm_fwd mh_layernorm_bwd mh_resconv_fwd mh_resconv_wgrad mh_ppeg_fwd,mh_ppeg_wgrad,mh_ppeg_merge mh_dropout,mh_dropout_add mh_landmark_fwd,mh_landmark_bwd mh_rank_mask mh_nys_attn1_fwd mh_nys_attn3_fwd mh_nys_attn1_bwd mh_nys_attn3_bwd mh_pinv_chain_fwd mh_pinv_chain_bwd mh_colsum mh_fanout_bwd mh_mse_masked_fwd,mh_mse_masked_bwd mh_rna_block_fwd,mh_rna_block_bwd mh_skinny_fwd,mh_skinny_wgrad mh_cast mh_pinv_z0_bwd,mh_pinv_absmax,mh_pinv_chain_prep,mh_pinv_chain_pack mh_softmax_fwd,mh_softmax_bwd mh_relu_bwd mh_mask_apply_fwd,mh_mask_apply_bwd mh_gemm; do run $k; done
run none

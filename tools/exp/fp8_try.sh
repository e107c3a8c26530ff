for v in "MIRROR_HEADS_SIDE=0 MIRROR_RNA_LATE=0" "MIRROR_HEADS_SIDE=1 MIRROR_RNA_LATE=0" "MIRROR_HEADS_SIDE=0 MIRROR_RNA_LATE=1" "MIRROR_HEADS_SIDE=1 MIRROR_RNA_LATE=1"; do
  echo "== $v"
  env $v python -m pytest tests/test_model_gpu.py -q -m gpu -k fp8_forward_policy --timeout 600 2>&1 | grep -E "passed|failed|ACTUAL|DESIRED|^E    " | head -12
done

import sys, os, collections, torch
sys.path.insert(0, os.getcwd())
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
from mirror_amd import functional as Fn
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                                    style_loss_weight=0.1, cluster_loss_weight=0.1), lr=2e-5, precision="bf16")
names = {p.data_ptr(): n for n, p in model.named_parameters()}
orig = Fn.flush_skinny_wgrads
def spy():
    q = Fn._wgrad_queue or []
    cur = torch.cuda.current_stream()
    main = torch.cuda.default_stream()
    for e in q:
        st = e[6]
        tag = "flush-stream" if st == cur else ("MAIN" if st == main else f"other {st}")
        print(f"{tag:14s} {names.get(e[4].data_ptr(), '?'):55s} dy {tuple(e[0].shape)}")
    return orig()
Fn.flush_skinny_wgrads = spy
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(16, 4096, 1024, device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(16, 2048, device=dev, generator=g)
eng.step(wsi, rna)
torch.cuda.synchronize()

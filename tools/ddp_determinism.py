"""Two gloo ranks on one GPU run the eager bucketed step twice (second engine optionally with the graphed RNA branch):\nper-parameter differences of the reduced gradients and checksums — they must agree from run to run (SECOND_ON=0/1, GATHER=0/1)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp
import test_engine_gpu as T
def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eo, init, lo = T._run_eager(False, 4, gather=os.environ.get("GATHER", "1") == "1")
        en, _, ln = T._run_eager(os.environ.get("SECOND_ON", "1") == "1", 4, gather=os.environ.get("GATHER", "1") == "1")
        if rank == 0:
            names = {id(p): n for n, p in en.model.named_parameters()}
            for s in (1, 2, 3):
                bad = []
                for p, o in zip(en.params, en.offsets):
                    a, b = eo.grad_snaps[s][o:o + p.numel()], en.grad_snaps[s][o:o + p.numel()]
                    d, r = float((a - b).norm()), float(a.norm())
                    if d > 2e-2 * r + 1e-7: bad.append((names[id(p)], round(r, 5), round(float(b.norm()), 5), round(d, 5)))
                print("step", s, "mismatching params:", len(bad), bad[:12], flush=True)
            for s in (0, 1, 2):
                print("checksum step", s, "first", float(eo.grad_snaps[s].double().norm()), "second", float(en.grad_snaps[s].double().norm()), flush=True)
            print("buckets", len(en.buckets), "state", en._rna_branch_state, flush=True)
    finally:
        dist.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29955)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]

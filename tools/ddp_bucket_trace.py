"""Two gloo ranks on one GPU: how often does each parameter report to its gradient bucket per step, which buckets are\nleft for _finish_reduce, which went negative (each parameter must report exactly once)."""
import sys, os, collections
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp
import test_engine_gpu as T
def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mirror_amd.engine import TrainEngine
        cnt = collections.Counter(); neg = []
        orig = TrainEngine._on_grad
        def spy(self, p):
            cnt[id(p)] += 1
            if rank == 0 and cnt[id(p)] <= 2 and p is self.params[1] and not self._counting:
                import traceback
                print("CALL", cnt[id(p)], "".join(traceback.format_stack(limit=6)[:-1]).replace("\n", " | ")[:900], flush=True)
            orig(self, p)
            b = self._bucket_of[id(p)]
            if self._pending[b] < 0: neg.append((b, self._pending[b]))
        TrainEngine._on_grad = spy
        origf = TrainEngine._finish_reduce
        def fin(self):
            left = [b for b, c in enumerate(self._pending) if c != 0]
            origf(self)
            if rank == 0:
                names = {id(p): n for n, p in self.model.named_parameters()}
                multi = [(names[k], v) for k, v in cnt.items() if v != 1]
                never = [names[id(p)] for p in self.params if cnt[id(p)] == 0]
                idx = {id(p): i for i, p in enumerate(self.params)}
                print("step: leftover", left, "n_multi", len(multi), "of", len(self.params), "multi uses", [(names[k], v, self._uses[idx[k]]) for k, v in cnt.items() if v != 1][:6], "single", [(names[k], self._uses[idx[k]]) for k, v in cnt.items() if v == 1][:4], flush=True)
            cnt.clear(); neg.clear()
        TrainEngine._finish_reduce = fin
        T._run_eager(False, 3, gather=False)
    finally:
        dist.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29966)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]

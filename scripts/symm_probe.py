"""Does torch.distributed._symmetric_memory work here (ROCm, dmabuf IPC)?  2 ranks on ONE GPU over gloo: each rank writes
its shard straight into the peer's buffer (the one-shot all-gather pattern), then both check the assembled result.
usage: python scripts/symm_probe.py   (scripts/, experiment)"""
import os
import sys
import traceback

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.distributed._symmetric_memory as symm

        n = 2048 * 768
        buf = symm.empty((world * n,), dtype=torch.float32, device="cuda:0")
        hdl = symm.rendezvous(buf, dist.group.WORLD)
        local = torch.full((n,), float(rank + 1), device="cuda:0")
        hdl.barrier()
        for p in range(world):
            peer = hdl.get_buffer(p, (world * n,), torch.float32)
            peer[rank * n:(rank + 1) * n].copy_(local)
        torch.cuda.synchronize()
        hdl.barrier()
        want = torch.cat([torch.full((n,), float(r + 1)) for r in range(world)]).to("cuda:0")
        print(f"rank {rank}: symmetric memory one-shot all-gather ok = {torch.equal(buf, want)}", flush=True)
    except Exception:
        print(f"rank {rank}: FAILED\n{traceback.format_exc()}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)

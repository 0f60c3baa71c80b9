"""Two ranks on ONE GPU, gloo carrying a small device tensor, no libicpx: does the
all-reduce stall by itself when kernels are in flight?  Launch with torch.distributed.run."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    faulthandler.dump_traceback_later(60, exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = torch.cuda.Stream()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    with torch.cuda.stream(st):
        x = torch.rand(n, device="cuda", dtype=torch.float64)
        sums = torch.zeros(40, device="cuda", dtype=torch.float64)
        t0 = time.time()
        for i in range(iters):
            y = x
            for _ in range(6):
                y = y * 1.0000001 + 1e-9
            sums += y[:40]
            dist.all_reduce(sums)
            sums *= 0.5
            if i % 200 == 0:
                print("[rank %d] iteration %d  %.1f s" % (rank, i, time.time() - t0), flush=True)
    st.synchronize()
    print("[rank %d] done: %g" % (rank, float(sums[0])), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Micro-benchmark: why is a replayed search 3-5 % slower than the same launches issued directly?
N dependent, launch-sized kernels (a 64-element in-place add: ~2 us of device time each, every one waiting for the one before),
issued (a) directly on the stream, (b) as one captured hipGraph -- device time from the first start to the last end (HIP events
around the burst), per kernel.  What differs between the two is only how ROCm 7.2 chains dependent kernel nodes: in-order queue
packets with the barrier bit (stream) against the graph executor's own dependency handling."""
import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)


def burst(n):
    for _ in range(n):
        x.add_(1.0)


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3          # us per call of fn


def run_table(title, sizes):
    print(title)
    for n in sizes:
        direct = timed(lambda: burst(n))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            burst(n)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            burst(n)
        replay = timed(g.replay)
        print("n = %2d   direct %8.2f us (%.2f per kernel)   graph replay %8.2f us (%.2f per kernel)   replay - direct = %+.2f us per kernel"
              % (n, direct, direct / n, replay, replay / n, (replay - direct) / n))


# (1) launch-sized kernels: the direct column is HOST-bound here (the host cannot issue faster than ~4.5 us per launch), the replay
# column shows what a dependent graph node costs on the device side
run_table("dependent launch-sized kernels (64 elements) per burst: device us per burst, per kernel", (4, 6, 8, 16))
# (2) kernels long enough for the host to run ahead (the searches' case: 20-170 us kernels): direct launches sit back to back on
# the stream; the replay pays its per-node dependency handling on top of the same kernels
x = torch.zeros(24 * 1024 * 1024, device=dev)
run_table("dependent ~30 us kernels (24 M elements) per burst", (4, 6, 8))

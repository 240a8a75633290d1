#!/usr/bin/env python3
"""Cost of many fork/join regions inside one captured HIP graph: R regions x W branches x L spin kernels.
(On ROCm 7.2 building a fifth graph of 16 regions x 4 x 8 in the same process segfaulted; left out.)"""
import time

import torch


def build(regions, width, length, cycles):
    main = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(width - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            for _ in range(regions):
                if width > 1:
                    ev = torch.cuda.Event()
                    ev.record(main)
                for s in sides:
                    s.wait_event(ev)
                    with torch.cuda.stream(s):
                        for _ in range(length):
                            torch.cuda._sleep(cycles)
                for _ in range(length):
                    torch.cuda._sleep(cycles)
                for s in sides:
                    main.wait_stream(s)
    return g


def timeit(g, reps=20):
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if __name__ == '__main__':
    torch.zeros(1, device='cuda')
    for cycles in (5000, 20000, 100000):
        for regions, width, length in ((64, 1, 8), (64, 2, 4), (64, 4, 2), (64, 8, 1)):
            t = timeit(build(regions, width, length, cycles))
            print('spin %6d  regions %3d x width %d x length %d (%4d kernels): graph %.3f ms  -> %.2f us per kernel' %
                  (cycles, regions, width, length, regions * width * length, t, t * 1e3 / (regions * width * length)), flush=True)

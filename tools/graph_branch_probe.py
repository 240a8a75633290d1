#!/usr/bin/env python3
"""Does a HIP-graph replay run independent captured branches concurrently on this ROCm?  Captures W
streams x L spin kernels (each ~20 us, one workgroup) as fork/join branches and compares the replay time
with the same work captured on one stream."""
import time

import torch


def build(width, length, cycles):
    main = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(width - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
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


def eager(width, length, cycles, reps=10):
    main = torch.cuda.current_stream()
    sides = [torch.cuda.Stream() for _ in range(width - 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
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
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if __name__ == '__main__':
    torch.zeros(1, device='cuda')
    for cycles in (2000, 40000):
        for width in (1, 2, 4, 8):
            length = 64 // width
            print('spin %6d cycles  width %d x length %3d : graph %.3f ms   eager %.3f ms' %
                  (cycles, width, length, timeit(build(width, length, cycles)), eager(width, length, cycles)), flush=True)

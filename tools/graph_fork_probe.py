#!/usr/bin/env python3
"""How does this HIP runtime's graph executor run the branches of a captured multi-stream graph?  Synthetic captures: spin
kernels of ~50 us on a main stream and four side streams in several fork / join topologies; replay time tells how many
branches really ran at once (1.0 = all branches in parallel, 5.0 = one after the other).

    python tools/graph_fork_probe.py
"""
import sys
import time

import torch

SPIN = None


def spin(n=1):
    for _ in range(n):
        torch.cuda._sleep(SPIN)


def calibrate():
    global SPIN
    SPIN = 100000
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    spin(20)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    SPIN = int(SPIN * 50.0 / us)
    return us


def measure(name, body, lanes, per_branch):
    main = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main, capture_error_mode='thread_local'):
            body(main, lanes)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print('%-34s %7.3f ms  = %.2f x one branch (%d spins of 50 us)' % (name, ms, ms / (per_branch * 0.05), per_branch), flush=True)


def fork(main, lane):
    lane.wait_stream(main)


def join(main, lanes):
    for s in lanes:
        main.wait_stream(s)


K = 10


def same_point(main, lanes):
    spin(1)
    for s in lanes:
        fork(main, s)
    for s in lanes:
        with torch.cuda.stream(s):
            spin(K)
    spin(K)
    join(main, lanes)
    spin(1)


def staggered(main, lanes):
    # main: 2 spins, fork lane 0, 2 spins, fork lane 1, ... each lane K spins; main runs K spins in all
    for s in lanes:
        spin(2)
        fork(main, s)
        with torch.cuda.stream(s):
            spin(K)
    spin(K - 2 * len(lanes))
    join(main, lanes)
    spin(1)


def prefork_then_staggered(main, lanes):
    # all lanes forked at ONE point by a one-spin stub, their real work gated later by main events
    spin(1)
    for s in lanes:
        fork(main, s)
        with torch.cuda.stream(s):
            spin(1)
    for s in lanes:
        spin(2)
        fork(main, s)
        with torch.cuda.stream(s):
            spin(K)
    spin(K - 2 * len(lanes))
    join(main, lanes)
    spin(1)


def chain(main, lanes):
    # main forks lane 0, lane 0 forks lane 1, ...
    spin(1)
    prev = main
    for s in lanes:
        s.wait_stream(prev)
        with torch.cuda.stream(s):
            spin(1)
        prev = s
    for s in lanes:
        with torch.cuda.stream(s):
            spin(K)
    spin(K)
    join(main, lanes)
    spin(1)


def interleaved_launch(main, lanes):
    # same point fork, but the host launches the branches' kernels round-robin (as autograd does with several cells in flight)
    spin(1)
    for s in lanes:
        fork(main, s)
    for _ in range(K):
        for s in lanes:
            with torch.cuda.stream(s):
                spin(1)
        spin(1)
    join(main, lanes)
    spin(1)


def refork_each_level(main, lanes):
    # 3 levels: at every level all lanes fork from main at one point, run K // 3 spins, join
    for _ in range(3):
        spin(1)
        for s in lanes:
            fork(main, s)
        for s in lanes:
            with torch.cuda.stream(s):
                spin(K // 3)
        spin(K // 3)
        join(main, lanes)


def relay_star(main, lanes):
    # same-point fork; then a hand-over lane 1 -> lane 0 through main (main waits lane 1, lane 0 waits main), twice
    spin(1)
    for s in lanes:
        fork(main, s)
    for s in lanes:
        with torch.cuda.stream(s):
            spin(K // 2)
    for a, b in ((1, 0), (2, 1), (3, 2)):
        main.wait_stream(lanes[a])
        lanes[b].wait_stream(main)
    for s in lanes:
        with torch.cuda.stream(s):
            spin(K // 2)
    spin(K)
    join(main, lanes)
    spin(1)


def main():
    us = calibrate()
    print('spin calibrated: 100000 cycles = %.1f us -> %d cycles per 50 us' % (us, SPIN))
    lanes = [torch.cuda.Stream() for _ in range(4)]
    measure('serial (all on main), 5 x K', lambda m, l: spin(5 * K + 2), lanes, K)
    measure('fork 4 lanes at one point', same_point, lanes, K)
    measure('fork 4 lanes, staggered', staggered, lanes, K)
    measure('stub fork at one point + staggered', prefork_then_staggered, lanes, K)
    measure('chain of forks', chain, lanes, K)
    measure('one point, round-robin launches', interleaved_launch, lanes, K)
    measure('re-fork at each of 3 levels', refork_each_level, lanes, 3 * (K // 3))
    measure('one point + relays through main', relay_star, lanes, K)
    measure('fork 2 lanes at one point', same_point, lanes[:2], K)
    measure('fork 3 lanes at one point', same_point, lanes[:3], K)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Only the supernet search step (bench.py's search leg), for rocprofv3 --kernel-trace:

    rocprofv3 --kernel-trace -d gpurun_out/search_trace -- python3 tools/search_profile.py [steps]
    python tools/trace_by_grid.py gpurun_out/search_trace
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    """search_profile.py [steps] [--eager]: --eager runs the two passes launch by launch instead of replaying the captured
    graphs (rocprofv3 --pmc crashed inside the tool on the replayed search graphs, ~2 800 nodes each, in round 3; the
    per-launch counters of the same kernels on the same tensors do not depend on how they were launched)."""
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    steps = int(args[0]) if args else 2
    print(bench.bench_search(torch.device('cuda:0'), steps, 0, 1, use_graph='--eager' not in sys.argv), flush=True)


if __name__ == '__main__':
    main()

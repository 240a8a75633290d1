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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    print(bench.bench_search(torch.device('cuda:0'), steps, 0, 1), flush=True)


if __name__ == '__main__':
    main()

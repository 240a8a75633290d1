"""The lane scheduler (csrc/sched.hip) behind a captured pass: the captured multi-stream graph is never instantiated; its
nodes are re-issued as single-branch graphs on a handful of streams with an event per cross-lane dependency."""
import ctypes as C
import os

import torch

from . import _lib

# lanes per captured pass: one per hardware queue (GPU_MAX_HW_QUEUES defaults to 4); the scheduler takes 1 .. 16
MAX_LANES = min(16, max(1, int(os.environ.get('SENAS_MAX_LANES', 4))))


_PENDING = []          # scheduler handles whose destruction had to wait (see LaneSchedule.close)


class LaneSchedule(object):
    def __init__(self, graph, max_lanes=None):
        """``graph``: a ``torch.cuda.CUDAGraph(keep_graph=True)`` whose capture has ended; it is kept alive here (it owns the
        captured hipGraph_t and the memory pool the pass lives in)."""
        self.graph = graph
        self.handle = C.c_void_p()
        # the device the pass was captured on: the scheduler's lane pool is per device and its segments are launched on that
        # device's streams, whatever the current device is when launch() / close() are called
        self.device = torch.cuda.current_device()
        lanes = min(16, max(1, int(max_lanes or MAX_LANES)))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().senas_sched_create(C.c_void_p(graph.raw_cuda_graph()), lanes, C.byref(self.handle)), 'senas_sched_create')
        if os.environ.get('SENAS_SCHED_VERBOSE'):
            import sys
            sys.stderr.write('[lanesched] %s\n' % self.info())

    def launch(self):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().senas_sched_launch(self.handle, torch.cuda.current_stream().cuda_stream), 'senas_sched_launch')

    def info(self):
        out = (C.c_int32 * 8)()
        _lib.check(_lib.lib().senas_sched_info(self.handle, out), 'senas_sched_info')
        keys = ('nodes', 'lanes', 'segments', 'cross_lane_dependencies', 'kernel_nodes', 'memset_nodes', 'memcpy_nodes', 'empty_nodes')
        return dict(zip(keys, [int(v) for v in out]))

    def close(self):
        """Destroy the scheduler (its segment graphs and events) once the device is idle.  Called from ``__del__`` too, i.e.
        possibly by the garbage collector in the middle of somebody's stream capture, where a device synchronisation is not
        allowed (it would invalidate the capture): the handle then waits in ``_PENDING`` for the next close outside a capture."""
        if not self.handle:
            return
        handle, self.handle = self.handle, C.c_void_p()
        if torch.cuda.is_current_stream_capturing():
            _PENDING.append(handle)
            return
        torch.cuda.synchronize(self.device)
        for h in [handle] + _PENDING:
            _lib.lib().senas_sched_destroy(h)
        del _PENDING[:]

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown
            pass

"""The UNet++-style macro grid shared by the supernet and the derived network.

Level 0 is the down path (stem1 + depth-1 down cells); level i >= 1 holds the up cells
``(i, j)``, j = 0 .. depth-i-1.  Up cell (i, j) takes as in0 the concatenation of the outputs that
currently sit at positions j .. i+j-1 of the running output list (all at resolution level j) and
as in1 the output at position i+j, and overwrites position i+j.
Reference: search/senas_search.py:16-112 and models/senas_model.py:78-179 build and walk the same
grid; here the bookkeeping lives in one place.
"""
import contextlib
import os

import torch
import torch.nn as nn

from .operations import ConvBn, Stem1


class FanPlan(object):
    """Readers of the tensors a macro-grid forward pass produces.  A tensor with several consumers (a cell output that
    is a skip input of later cells, the in1 of the next cell and a blend operand) would have its gradient accumulated by
    autograd in n - 1 binary adds; with the number of readers known up front every reader gets its own alias and the
    gradient is ONE n-ary sum (functional.fan_out).  The schedule is static, so the counts come from a dry run of the
    same forward code: ``put`` / ``get`` count in dry mode and hand out aliases in live mode."""

    def __init__(self):
        self.counts, self.live, self.dry = {}, {}, True

    def put(self, key, value=None):
        if self.dry:
            self.counts[key] = 0
        else:
            from . import functional as F
            n = self.counts[key]
            self.live[key] = iter(F.fan_out(value, n)) if n > 1 else iter([value] * max(n, 1))
        return key

    def get(self, key):
        if self.dry:
            self.counts[key] += 1
            return None
        return next(self.live[key])

    def start(self):
        """Switch to live mode for one forward pass (the counts stay)."""
        self.dry = False
        self.live = {}
        return self

    def __deepcopy__(self, memo):
        fresh = FanPlan()                 # (the aliases of the last pass are not part of a copied model)
        fresh.counts, fresh.dry = dict(self.counts), self.dry
        return fresh


class NoPlan(object):
    """The FanPlan interface without aliases: every reader takes the tensor itself and autograd accumulates (the two-part
    backward of a multi-rank step re-leafs tensors in the middle of the schedule, which aliases made earlier would miss)."""
    dry = False

    def put(self, key, value=None):
        return value

    def get(self, key):
        return key


class Lanes(object):
    """The macro grid on several HIP streams: one lane per column of up cells.

    Up cell (i, j) reads the outputs of the cells below it in ITS column (levels < i) and the output of cell (i-1, j+1):
    the columns are chains that meet along the diagonals, and the six small-map cells of a depth-5 grid (64 x 64 ... 16 x 16
    maps: kernels of a few blocks each, bound by launch latency) can run under the four 128 x 128 cells of column 0 instead of
    in front of them.  Lane j is an ordinary stream; ``on(j)`` makes it torch's current stream, so every kernel of the cell --
    and, through autograd's stream rule, every kernel of the cell's backward pass -- is launched on it.  Inside a HIP-graph
    capture the waits become the fork / join edges of the graph.  The reference walks the same cells one after the other
    (search/senas_search.py:96-107, models/senas_model.py:160-175); the outputs are functions of the inputs, so any schedule
    that respects the data flow gives its results.

    **Star topology, and why.**  Lanes only ever wait for events recorded on the caller's stream ("main": stems, down path,
    head -- under capture the ORIGIN stream of the capture), and only main waits for lane events.  A tensor that goes from
    lane a to lane b is handed over THROUGH main: main waits for a's event, the tensor passes an identity autograd node made
    on main (functional.hop), b waits for main's event -- and autograd, which replays every node on its forward stream and
    synchronises producer and consumer streams itself, then does the same in the other direction on the way back.  The HIP
    runtime under this torch build (ROCm 7.0 libamdhip64) keeps, for every stream of a capture, the list of streams that
    waited on one of its events, and ``hip::Stream::EndCapture()`` walks those lists recursively before clearing them; only
    the origin stream is never entered in a list.  Lane j waiting on lane j+1 in the forward pass and lane j+1 on lane j in
    the backward pass makes the lists cyclic, and hipStreamEndCapture recurses until the stack is gone (measured:
    profiles/r4_endcapture_recursion.txt -- 174 575 frames of hip::Stream::EndCapture).  Main records no kernel between two
    hand-overs, so a hand-over also waits for the producers of the earlier ones (the order of the schedule below keeps that
    harmless)."""

    enabled = True          # class-wide switch (False: every cell on the caller's stream, as the reference's loop)
    down_lane = os.environ.get('SENAS_DOWN_LANE', '1') != '0'        # the down cells on a lane of their own (False: on the caller's stream, round 4's schedule)
    # Inside a stream capture the columns only fork when whoever captures has said that the captured graph goes to the lane
    # scheduler (``with Lanes.scheduled():`` -- the step drivers and the Evaluator): anybody else's ``torch.cuda.graph`` would
    # instantiate the multi-branch graph on the runtime's own executor (SIGSEGV in hip::Graph::UpdateStreams,
    # profiles/r4_graph_executor.txt) -- their capture gets the serial schedule.
    _scheduled = 0

    @staticmethod
    @contextlib.contextmanager
    def scheduled():
        Lanes._scheduled += 1
        try:
            yield
        finally:
            Lanes._scheduled -= 1

    @staticmethod
    def allowed():
        return Lanes.enabled and (Lanes._scheduled > 0 or not torch.cuda.is_current_stream_capturing())

    def __init__(self, device, columns):
        from . import functional as F
        self.F = F
        self.main = torch.cuda.current_stream(device)
        # (streams of this library's own, not torch.cuda.Stream(): see functional.own_stream)
        self.streams = [F.own_stream(device, 'lane%d' % j) for j in range(columns)]
        # The down cells on a lane of their own (``down_lane``): on the caller's stream their backward passes sit behind every
        # hand-over the host issued before them -- the chain of relay markers on that stream carries all of those waits on --
        # and ran as a serial tail of the pass (2.5 ms of the search step's weight pass: profiles/r4_search_by_level.txt, down4
        # waiting for cells of column 0 it has nothing to do with).  On a lane a down cell waits for exactly the hand-overs it
        # takes part in (csrc/sched.hip: the typed markers).
        self.down = F.own_stream(device, 'lane_down') if Lanes.down_lane else None
        self.events = {}
        self.used = []

    def mark(self, key):
        """``key`` is produced by what the current stream holds so far.  On a lane the event sits behind a PRODUCER marker: the
        lane scheduler then knows which parent of a relay marker on the origin stream is the hand-over's source."""
        s = torch.cuda.current_stream()
        if s != self.main:
            self.F.marker(self.F.MARK_PRODUCER)
        ev = torch.cuda.Event()
        ev.record(s)
        self.events[key] = (ev, s)

    def down_stream(self):
        """The stream the down cells run on: their own lane, or the caller's stream."""
        if self.down is None:
            return self.main
        if self.down not in self.used:
            self.used.append(self.down)
            self.F.LANES.add(self.down)
        return self.down

    def after_down(self, keys):
        """The down lane waits for the producers of ``keys`` that ran on the caller's stream (stems)."""
        s = self.down_stream()
        for key in keys:
            ev, src = self.events[key]
            if src == self.main and s != self.main:
                s.wait_event(ev)

    def lane(self, j):
        s = self.streams[j]
        if s not in self.used:
            self.used.append(s)
            self.F.LANES.add(s)
        return s

    def after(self, j, keys):
        """Lane j waits for the producers of ``keys`` that ran on main (for those of other lanes: ``hand``)."""
        s = self.lane(j)
        for key in keys:
            ev, src = self.events[key]
            if src == self.main:
                s.wait_event(ev)
            elif src != s:
                raise RuntimeError('lane %d may not wait for another lane directly (grid.Lanes)' % j)

    def hand(self, t, key, j):
        """Tensor ``t``, produced under ``key`` on another stream, for a reader on lane j (``'down'``: the down lane): through
        main (see the class text).  Three identity nodes mark the hand-over for the lane scheduler, forward and backward: one made
        on the producer's lane (backward: a CONSUMER marker there -- the gradient arrives), the hop on main (a RELAY marker both
        ways), one on the reader's lane (forward: a CONSUMER marker; backward: a PRODUCER marker -- the gradient leaves).  With
        them the scheduler gives the reader exactly the producer as its dependency instead of everything main had waited for
        before (csrc/sched.hip)."""
        ev, src = self.events[key]
        s = self.down_stream() if j == 'down' else self.lane(j)
        if src == s:
            return t
        if src != self.main:
            with torch.cuda.stream(src):
                t = self.F.lane_out(t)
            self.main.wait_event(ev)
            with torch.cuda.stream(self.main):
                t = self.F.hop(t)
            ev = torch.cuda.Event()
            ev.record(self.main)
            s.wait_event(ev)
            if s != self.main:
                with torch.cuda.stream(s):
                    t = self.F.lane_in(t)
            return t
        s.wait_event(ev)
        return t

    @contextlib.contextmanager
    def on(self, j):
        with torch.cuda.stream(self.lane(j)):
            yield

    @staticmethod
    def take(t):
        """A tensor made on another stream is read on the current one: tell the caching allocator, which otherwise hands the
        block to the next allocation of the PRODUCER's stream as soon as the last reference dies, ordered or not."""
        if t is not None and t.is_cuda:
            s = torch.cuda.current_stream()
            t.record_stream(s)
            st = getattr(t, '_senas_stats', None)
            if st is not None:
                st.record_stream(s)
        return t

    def join(self):
        """The caller's stream waits for every lane."""
        for s in self.used:
            self.main.wait_stream(s)


def gamma_index(i, j):
    """Index into the flat gamma table of the skip feeding row i+j from column j."""
    return sum(range(i + j)) + j


class MacroGrid(nn.Module):
    """Owns ``stem0``, ``stem1``, ``blocks`` and ``head_block`` with the reference's names.

    make_cell(cell_type, c_in0, c_in1, c_out, i, j) -> nn.Module or None (None = pruned up cell)
    make_head(c_in0, c_in1, nclass) -> nn.Module
    """

    def __init__(self, in_channels, c, nclass, depth, double_down_channel, make_cell, make_head):
        super().__init__()
        assert depth >= 2, 'depth must >= 2'
        self._depth = depth
        self._double_down_channel = double_down_channel
        double = 2 if double_down_channel else 1
        self.blocks = nn.ModuleList()
        self.stem0 = ConvBn(in_channels, c, kernel_size=7)
        self.stem1 = Stem1(c, c)
        widths = [[c]]                       # widths[level][j] = channels produced at grid slot (level, j)
        row = nn.ModuleList([self.stem1])
        c_in0, c_in1, c_cur = c, c, c
        for _ in range(1, depth):
            c_cur = int(double * c_cur)
            row.append(make_cell('down', c_in0, c_in1, c_cur, 0, len(row)))
            widths[0].append(c_cur)
            c_in0, c_in1 = c_in1, c_cur
        self.blocks.append(row)
        for i in range(1, depth):
            row, wrow = nn.ModuleList(), []
            for j in range(depth - i):
                c_skip = sum(widths[k][j] for k in range(i))
                cell = make_cell('up', c_skip, widths[i - 1][j + 1], widths[0][j], i, j)
                row.append(cell)
                wrow.append(widths[0][j] if cell is not None else 0)
            self.blocks.append(row)
            widths.append(wrow)
        self.head_block = nn.ModuleList([make_head(c, widths[-1][0], nclass)])
        # step drivers may cut the autograd graph at the outputs of the down path (senas_amd.step: backward is then run
        # -- and captured -- in two parts, so that the up-path gradients are all-reduced while the rest still runs):
        # cut(tensor) -> the leaf the up path reads in its place
        self.cut = None
        self.lanes = True        # this grid's columns may run on lanes (grid.Lanes; a step driver turns it off where it must)

    def down_parameters(self):
        """Parameters of the down path (stems, down cells): the ones whose gradients backward produces last."""
        out, seen = [], set()
        for m in [self.stem0, self.stem1] + list(self.blocks[0]):
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    def _walk_grid(self, plan, x, run, skips):
        """The forward schedule of both networks against a FanPlan / NoPlan: dry (``x is None``: count the readers of every
        tensor) or live.  ``run(module, kind, a, b)`` applies a cell or the head; ``skips(plan, G, i, j, live)`` returns the
        list of tensors whose concatenation is in0 of up cell (i, j) (and takes its readers from ``plan``).  G[i][j]: output
        of grid slot (i, j) -- level 0 is the down path.

        Order: down cell j + 1 is followed at once by up cell (1, j), which needs nothing else; then the levels 2, 3, ... with
        j ascending.  With ``Lanes.enabled`` column j of the up cells runs on lane j.  The order matters beyond the data flow:
        autograd replays the nodes in reverse creation order and makes a consumer's STREAM wait when a gradient is handed
        over, so whatever the host launches later on that stream waits too -- with this order the backward pass of down cell
        j + 1 is launched after that of up cell (1, j) and before that of (1, j - 1), and waits for exactly what it needs."""
        live = x is not None
        depth = self._depth
        lanes = Lanes(x.device, depth - 1) if (live and self.lanes and x.is_cuda and depth > 2 and Lanes.allowed()) else None
        cut = self.cut if live else None
        G = [[None] * (depth - i) for i in range(depth)]

        def call(name, module, kind, a, b):
            """run(...), between time stamps when a timeline is being taken (tools/lane_timeline.py)."""
            from . import functional as F
            a, b = F.cell_in(a), F.cell_in(b)      # (the end of the cell's backward pass: its queued weight gradients go to their lane)
            if F.STAMPS is None:
                return run(module, kind, a, b)
            return F.stamp(run(module, kind, F.stamp(a, name + '.in0'), F.stamp(b, name + '.in1')), name + '.out')

        def up_cell(i, j):
            cell = self.blocks[i][j]
            if cell is None:                                   # pruned skip cell of a derived network
                return
            if G[i - 1][j + 1] is None:
                raise TypeError('up cell (%d, %d) is alive but its input cell (%d, %d) is pruned' % (i, j, i - 1, j + 1))
            in1 = plan.get(G[i - 1][j + 1])
            if lanes is None:
                ins = skips(plan, G, i, j, live)
                y = call('up%d%d' % (i, j), cell, 'up', ins[0] if len(ins) == 1 else torch.cat(ins, dim=1), in1) if live else None
                G[i][j] = plan.put(('o', i, j), y)
                return
            in1 = lanes.hand(in1, (i - 1, j + 1), j)

            def fetch(k, t):
                # a tensor of the column as lane j reads it: the cells (k, j), k >= 1, ran on this lane; the column's down-path
                # output comes from the down lane (a hand-over) or from the caller's stream (a wait)
                return Lanes.take(lanes.hand(t, (k, j), j))

            with lanes.on(j):
                ins = skips(plan, G, i, j, live, fetch)
                y = call('up%d%d' % (i, j), cell, 'up', ins[0] if len(ins) == 1 else torch.cat(ins, dim=1), Lanes.take(in1))
                G[i][j] = plan.put(('o', i, j), y)
                lanes.mark((i, j))

        def cut_here(t, key=None):
            # the two-part backward of a multi-rank step (senas_amd.step) re-leafs the down-path outputs: what the up path
            # reads is the leaf, behind an identity node made on THIS stream (the leaf's gradient is then accumulated here,
            # whichever lanes its readers run on).  A tensor the down lane made is waited for here and from then on counts as
            # made on this stream (its readers wait for this stream, no hand-over)
            if cut is None:
                return t
            from . import functional as F
            if lanes is not None and key is not None and lanes.events[key][1] != lanes.main:
                lanes.main.wait_event(lanes.events[key][0])
                Lanes.take(t)
            out = F.hop(cut(t))
            if lanes is not None and key is not None:
                lanes.mark(key)
            return out

        s0 = plan.put('s0', self.stem0(x) if live else None)
        first = plan.get(s0)
        D = [plan.put(('o', 0, 0), self.stem1(first) if live else None)]             # the down path as the down path reads it
        if lanes is not None:
            lanes.mark((0, 0))
        G[0][0] = cut_here(D[0], (0, 0))
        # the down cells run on the down lane (grid.Lanes.down): it waits for the stems once, and its tensors reach the columns
        # of up cells through hand-overs
        on_down = (lambda: torch.cuda.stream(lanes.down_stream())) if lanes is not None else contextlib.nullcontext
        if lanes is not None:
            lanes.after_down([(0, 0)])
        for j in range(1, depth):
            with on_down():
                a, b = plan.get(s0 if j == 1 else D[j - 2]), plan.get(D[j - 1])
                if lanes is not None and j <= 2:                 # (what the stems made is read on the down lane)
                    a = Lanes.take(a)
                    b = Lanes.take(b) if j == 1 else b
                D.append(plan.put(('o', 0, j), call('down%d' % j, self.blocks[0][j], 'down', a, b) if live else None))
                if lanes is not None:
                    lanes.mark((0, j))
            G[0][j] = cut_here(D[j], (0, j))
            up_cell(1, j - 1)
        s0 = cut_here(s0)
        for i in range(2, depth):
            for j in range(depth - i):
                up_cell(i, j)
        if lanes is not None:
            lanes.join()
        head = self.head_block[-1]
        tails = [G[m][0] for m in range(depth)] if self._supervision else [G[depth - 1][0]]
        self._check_tails(tails)
        res = []
        for o in tails:
            a, b = plan.get(s0), plan.get(o)
            if lanes is not None:
                Lanes.take(b)                                  # made on lane 0, read here
            res.append(call('head', head, 'head', a, b) if live else None)
        return res

    def _check_tails(self, tails):
        if any(o is None for o in tails):
            raise TypeError('deep supervision needs every grid column alive; this genotype prunes some '
                            '(the reference fails the same way, models/senas_model.py:177)')

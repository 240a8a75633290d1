"""Search-phase cell: MixedOp (alpha-weighted sum of the candidate ops of one edge) and Cell
(beta-weighted DAG of MixedOps).  Same constructors, forward signatures and parameter names as the
reference's ``search/cell.py`` (``MixedOp`` :5-43, ``Cell`` :46-110).

Execution differs: a node is ONE ``bn_combine`` launch over the raw outputs of all candidates of
all incoming edges -- batch-norm, SE gate, alpha and beta weights, the edge sum and the node ReLU
are per-channel coefficients of that single pass instead of ~13 elementwise passes per edge.
"""
import torch
import torch.nn as nn

from . import functional as F
from .operations import OPS, OpType, RectifyBlock, ShrinkBlock, build_activation, build_rectify


class MixedOp(nn.Module):
    def __init__(self, c_in, c_out, op_type):
        super().__init__()
        self._op_type = op_type
        self.k = 1                       # PC-DARTS style partial channels are off in SENAS (k == 1)
        self.c_out = c_out
        self.c_part = c_out // self.k
        self._ops = nn.ModuleList(OPS[name](c_in, self.c_part, op_type, dp=0) for name in op_type.value['ops'])

    def pick(self, alpha_normal, alpha_up_dn):
        return alpha_normal if self._op_type == OpType.NORM else alpha_up_dn

    def terms(self, x):
        """x: the edge's input, or one alias of it per candidate (functional.fan_out)."""
        xs = x if isinstance(x, (list, tuple)) else [x] * len(self._ops)
        return [op.raw(xi) for op, xi in zip(self._ops, xs)]

    def forward(self, x, alpha_normal, alpha_up_dn):
        return F.bn_combine(self.terms(x), mix=self.pick(alpha_normal, alpha_up_dn))


class Cell(nn.Module):
    def __init__(self, meta_node_num, double_down, c_in0, c_in1, c_out, cell_type):
        super().__init__()
        self.k = 4                       # "shrink": every edge works on c_out / k channels while searching
        self._meta_node_num = meta_node_num
        self._input_num = 2
        if cell_type == 'down':
            self.preprocess0 = build_rectify(c_in0, c_in1, cell_type)
            c_part = (c_out // double_down) // self.k
        else:
            self.preprocess0 = ShrinkBlock(c_in0, c_in1)
            c_part = c_out // self.k
        self.preprocess1 = build_activation(False)
        self.node_activation = build_activation()
        self.post_process = RectifyBlock(c_part * meta_node_num, c_out, cell_type=cell_type)

        def edge_type(j):
            if j >= self._input_num:
                return OpType.NORM
            if cell_type == 'down':
                return OpType.DOWN
            return OpType.UP if j > 0 else OpType.NORM

        self._ops = nn.ModuleList()
        for i in range(meta_node_num):
            for j in range(self._input_num + i):
                self._ops.append(MixedOp(c_in1 if j < self._input_num else c_part, c_part, edge_type(j)))

    _norm_masks = {}                  # (edge kinds, device) -> bool [edges, 1] constant

    def _node_mixes(self, weights_norm, weights_chg, betas):
        """This cell's slot of the mixing table (cells of different lanes never share one)."""
        return self._mix_slots(weights_norm, weights_chg, betas).take()

    def _mix_slots(self, weights_norm, weights_chg, betas):
        """Per node, the concatenated ``beta_j * alpha_row_j`` of its incoming edges (search/cell.py:100-106), as rows
        of ONE matrix per cell kind (functional._EdgeMix).  They depend on the architecture tensors and the cell KIND
        only, so all cells of a kind share them within a forward pass: the matrix is parked on the ``betas`` tensor
        (fresh softmax outputs every pass, so nothing goes stale) and the nodes of all those cells accumulate their
        d loss / d mix into one table (a slot per cell) -- a handful of tiny launches per kind and pass instead of ~10 per cell."""
        kinds = tuple(edge._op_type == OpType.NORM for edge in self._ops)
        key = (id(weights_norm), id(weights_chg), kinds)
        cache = betas.__dict__.setdefault('_senas_mix', {}) if hasattr(betas, '__dict__') else {}
        if key not in cache:
            mkey = (kinds, str(betas.device))
            if mkey not in Cell._norm_masks:
                Cell._norm_masks[mkey] = torch.tensor(kinds, device=betas.device).unsqueeze(1)
            nops = weights_norm.shape[1]
            dM = torch.zeros((F.MIX_SLOTS, len(kinds), nops), device=betas.device, dtype=torch.float32)
            M = F._EdgeMix.apply(weights_norm, weights_chg, betas, Cell._norm_masks[mkey], dM)
            cache[key] = F.MixSlots(M, dM, [self._input_num + i for i in range(self._meta_node_num)])
        return cache[key]

    # ------------------------------------------------------------------ state-major execution
    stacked = True      # run same-named candidates of the edges LEAVING one state as one convolution (class-wide switch)
    fused_tail = True   # DepSepConv candidates: batch-norm + ReLU + 1x1 convolution as one pass (functional.dstail)
    pair_tails = True   # ... and the tails of BOTH input states' candidates as one launch (Cell.forward)
    pair_dilated = True  # dil_3_conv_5 + dil_2_conv_5 of the same edges: one forward and one data-gradient launch

    def _out_edges(self, j):
        """Flat indices of the edges that read state j (one per later node)."""
        out, offset = [], 0
        for i in range(self._meta_node_num):
            if j < self._input_num + i:
                out.append(offset + j)
            offset += self._input_num + i
        return out

    def _stack(self, convs):
        """The persistent stacked weight of these convolutions (functional.StackedWeight), made on first use."""
        stacks = self.__dict__.setdefault('_stacks', {})
        key = tuple(id(c) for c in convs)
        if key not in stacks:
            c0 = convs[0]
            for c in convs[1:]:
                if (type(c), c.weight.shape, c.stride, c.padding, c.dilation, c.groups) != (type(c0), c0.weight.shape, c0.stride,
                                                                                         c0.padding, c0.dilation, c0.groups):
                    raise F.SenasHipError('stacked candidates disagree in geometry')
            tr = isinstance(c0, nn.ConvTranspose2d)
            # three transposed 32 -> 8 candidates: a zero-weight fourth part makes the stack a full 32-channel tile, which
            # puts its weight gradient and data gradient on the LDS kernels (24 fine-grid channels fall off them)
            # (the same for stride-1 Conv2d stacks: their data gradient gathers over the 24 stacked channels)
            pad = 1 if (len(convs) == 3 and c0.out_channels * 4 == 32 and (tr or c0.stride[0] == 1)) else 0
            stacks[key] = F.StackedWeight([c.weight for c in convs], 1 if tr else 0, pad_parts=pad)
        return stacks[key]

    def stacked_weights(self):
        """Every StackedWeight this cell uses (built without running a forward pass), for the weight packer."""
        for j in range(self._input_num + self._meta_node_num):
            self._plan(j)
        post = self.post_process
        c_in = post.conv.in_channels
        if self.stacked and c_in % 16 != 0 and c_in <= 32 and c_in % 4 == 0:
            self.__dict__.setdefault('_stacks', {}).setdefault(('post', id(post.conv)), F.StackedWeight([post.conv.weight], 1, pad_to=32))
        return [v for v in self.__dict__.get('_stacks', {}).values() if isinstance(v, F.StackedWeight)]

    def _parts(self, convs):
        """Parts of the stacked output of these edges' convolutions (with the zero-weight padding part, if any)."""
        return len(convs) + self._stack(convs).pad_parts

    def _stacked_conv(self, convs, x, want_stats):
        """ONE convolution for the same-geometry convolutions of k edges that read the same tensor: weights stacked along
        c_out (every edge receives its slice of the stacked weight gradient).  Returns the stacked output and its
        producer-side statistics; the caller hands every edge its channel slice of both (functional.unstack: aliases, no
        launch)."""
        c0 = convs[0]
        tr = isinstance(c0, nn.ConvTranspose2d)
        w = self._stack(convs).tensor()
        return F.conv2d(x, w, stride=c0.stride[0], pad=c0.padding[0], dil=c0.dilation[0], transposed=tr,
                        out_pad=c0.output_padding[0] if tr else 0, groups=1, want_stats=want_stats, stacked=self._parts(convs))

    def _depsep_job(self, items):
        """DepSepConv candidates (edge, position, module) that read one state: their depthwise convolutions, ONE batched
        BatchNorm + ReLU over all the depthwise outputs (functional.bnrelu_multi), their pointwise convolutions."""
        from .operations import run_conv
        # the depthwise convolutions of all of them read the same state and differ in nothing but the kernel size: one batched
        # launch for dep_sep_conv_3 and dep_sep_conv_5 of every edge together (functional.dwconv_multi), one alias of the state
        groups = [list(range(len(items)))]

        def stage1(xs):                                      # the depthwise halves: [(z1, producer statistics)] per item
            zs, sts = [None] * len(items), [None] * len(items)
            for idxs, x in zip(groups, xs):
                mods = [items[i][2] for i in idxs]
                parts = F.dwconv_multi(x, [m[0] for m in mods], mods[0][1].training) if len(idxs) > 1 else None
                if parts is None:                            # off the batched path: one by one (x is read-only: sharing it is fine)
                    parts = [run_conv(m[0], x, want_stats=m[1].training) for m in mods]
                for i, (z, st) in zip(idxs, parts):
                    zs[i], sts[i] = z, st
            return zs, sts

        def job(xs):
            return self._depsep_tail(items, *stage1(xs))
        job.stages = (stage1, items)                          # (Cell.forward runs the tails of both input states as one launch)
        return job, len(groups)

    def _depsep_tail(self, items, zs, sts):
        """BatchNorm2d + ReLU + pointwise convolution of DepSepConv candidates whose depthwise halves are done: ONE fused
        launch (functional.dstail) when the shapes allow, else in chunks of batched batch-norm + pointwise launches."""
        from .operations import run_conv
        out = []
        fused = self.fused_tail and len(items) <= F.MAX_DSTAIL
        pws = F.dstail(zs, [m[1] for _, _, m in items], sts, [m[3] for _, _, m in items], items[0][2][4].training) if fused else None
        if pws is not None:
            return [(e, p, F.Term(z2, m[4], stats=st2)) for (e, p, m), (z2, st2) in zip(items, pws)]
        for i in range(0, len(items), F.MAX_BNRELU):
            it, z, st = items[i:i + F.MAX_BNRELU], zs[i:i + F.MAX_BNRELU], sts[i:i + F.MAX_BNRELU]
            bns = [m[1] for _, _, m in it]
            pws = F.dstail(z, bns, st, [m[3] for _, _, m in it], it[0][2][4].training) if self.fused_tail else None
            if pws is None:
                if F.bnrelu_multi_ok(z, bns):
                    mids = F.bnrelu_multi(z, bns, st)
                else:
                    mids = [F.bn_combine([F.Term(zz, bn, stats=s_)], relu=True) for zz, bn, s_ in zip(z, bns, st)]
                pws = F.pw_multi(mids, [m[3] for _, _, m in it], it[0][2][4].training)      # the pointwise halves, batched
                if pws is None:
                    pws = [run_conv(m[3], mid, want_stats=m[4].training) for (_, _, m), mid in zip(it, mids)]
            out += [(e, p, F.Term(z2, m[4], stats=st2)) for (e, p, m), (z2, st2) in zip(it, pws)]
        return out

    def _plan(self, j):
        """What has to run on state j: ``jobs`` -- pairs ``(fn, a)`` with ``fn(list of a aliases of the state) ->
        [(edge, op position, Term)]`` -- and ``fixed`` terms that read nothing ('none')."""
        from .operations import AdapterBlock, ConvBn, ConvBnSe, DepSepConv, ZeroOp
        edges = self._out_edges(j)
        jobs, fixed = [], []
        if not edges:
            return jobs, fixed
        k = len(edges)
        nops = len(self._ops[edges[0]]._ops)
        depsep = []
        dil5 = {}                                            # position -> modules of a 5x5 dilated ConvBn candidate (Conv2d form)

        def conv_job(mods, p):
            """The (stacked) convolution of one ConvBn / ConvBnSe candidate over the k edges."""
            def job(xs, mods=mods, p=p):
                convs = [m[0] for m in mods]
                se = isinstance(mods[0], ConvBnSe)
                want = mods[0][1].training or se
                z, st = self._stacked_conv(convs, xs[0], want)
                sw = self._stack(convs)
                parts = F.unstack(z, len(mods) + sw.pad_parts, want_stats=want, used=len(mods), owner=sw, stats=st)
                return [(e, p, F.Term(zz, m[1], se=m[2] if se else None, stats=st, grad_slot=slot))
                        for e, m, (zz, st, slot) in zip(edges, mods, parts)]
            return job

        for p in range(nops):
            mods = [self._ops[e]._ops[p] for e in edges]
            m0 = mods[0]
            if isinstance(m0, AdapterBlock) and isinstance(m0.module, ZeroOp):
                fixed += [(e, p, m.raw(None)) for e, m in zip(edges, mods)]
                continue
            stack = self.stacked and 1 < k <= F.MAX_STACK
            c0 = m0[0] if isinstance(m0, (ConvBn, ConvBnSe)) else None
            if (self.stacked and self.pair_dilated and type(m0) is ConvBn and isinstance(c0, nn.Conv2d) and c0.kernel_size == (5, 5) and
                    m0.drop is None and (stack or k == 1)):
                if stack:
                    self._stack([m[0] for m in mods])
                dil5[p] = mods
            elif stack and isinstance(m0, (ConvBn, ConvBnSe)):
                self._stack([m[0] for m in mods])
                jobs.append((conv_job(mods, p), 1))
            elif stack and isinstance(m0, AdapterBlock) and m0.c_in != m0.c_ot:
                self._stack([m.conv for m in mods])

                def job(xs, mods=mods, p=p):
                    convs = [m.conv for m in mods]
                    want = mods[0].norm.training
                    z, st = self._stacked_conv(convs, mods[0]._resample(xs[0]), want)     # resampled ONCE for the k edges
                    sw = self._stack(convs)
                    parts = F.unstack(z, len(mods) + sw.pad_parts, want_stats=want, used=len(mods), owner=sw, stats=st)
                    return [(e, p, F.Term(zz, m.norm, stats=st, grad_slot=slot)) for e, m, (zz, st, slot) in zip(edges, mods, parts)]
                jobs.append((job, 1))
            elif self.stacked and isinstance(m0, DepSepConv):
                depsep += [(e, p, m) for e, m in zip(edges, mods)]
            else:
                for e, m in zip(edges, mods):
                    jobs.append((lambda xs, e=e, m=m, p=p: [(e, p, m.raw(xs[0]))], 1))
        if len(dil5) == 2:
            # dil_3_conv_5 and dil_2_conv_5 of these edges: same tensor, same shapes, different dilation -- ONE launch forward
            # and one for the two data gradients (functional.conv2d_pair); two aliases of the state
            (pa, ma), (pb, mb) = sorted(dil5.items())

            def pair_job(xs, pa=pa, ma=ma, pb=pb, mb=mb):
                ca, cb = [m[0] for m in ma], [m[0] for m in mb]
                want = ma[0][1].training
                if k > 1:
                    swa, swb = self._stack(ca), self._stack(cb)
                    wa, wb = swa.tensor(), swb.tensor()
                else:
                    wa, wb = ca[0].weight, cb[0].weight
                (za, sta), (zb, stb) = F.conv2d_pair(xs[0], xs[1], wa, wb, ca[0].stride[0], ca[0].padding[0], ca[0].dilation[0],
                                                    cb[0].padding[0], cb[0].dilation[0], want_stats=want,
                                                    stacked=self._parts(ca) if k > 1 else 0)
                out = []
                for p, mods, z, st, convs in ((pa, ma, za, sta, ca), (pb, mb, zb, stb, cb)):
                    if k > 1:
                        sw = self._stack(convs)
                        parts = F.unstack(z, len(mods) + sw.pad_parts, want_stats=want, used=len(mods), owner=sw, stats=st)
                        out += [(e, p, F.Term(zz, m[1], stats=st_, grad_slot=slot)) for e, m, (zz, st_, slot) in zip(edges, mods, parts)]
                    else:
                        out.append((edges[0], p, F.Term(z, mods[0][1], stats=st)))
                return out
            jobs.append((pair_job, 2))
        else:
            for p, mods in sorted(dil5.items()):
                if k > 1:
                    jobs.append((conv_job(mods, p), 1))
                else:
                    jobs.append((lambda xs, e=edges[0], m=mods[0], p=p: [(e, p, m.raw(xs[0]))], 1))
        for i in range(0, len(depsep), F.MAX_BNRELU):
            chunk = depsep[i:i + F.MAX_BNRELU]
            jobs.append(self._depsep_job(chunk))
        return jobs, fixed

    def forward(self, in0, in1, weights_norm, weights_chg, betas):
        mixes = self._node_mixes(weights_norm, weights_chg, betas)
        nodes, nin = self._meta_node_num, self._input_num
        terms = {}                                  # flat edge index -> [Term per candidate]
        states = []

        held = []                                   # DepSepConv candidates of the FIRST input state: they wait for the second's

        def add_state(h):
            # everything that reads this state runs now; every reader (and the output concat) gets its own alias of
            # it, so that the state's gradient is ONE n-ary sum (functional.fan_out)
            j = len(states)
            jobs, fixed = self._plan(j)
            aliases = F.fan_out(h, sum(a for _, a in jobs) + (1 if j >= nin else 0))
            for e, p, t in fixed:
                terms.setdefault(e, {})[p] = t
            taken = 0
            for job, a in jobs:
                xs = aliases[taken:taken + a]
                taken += a
                stages = getattr(job, 'stages', None)
                if stages is not None and self.pair_tails and nin == 2 and j < 2:
                    # the two input states' DepSepConv candidates have one shape: their depthwise halves run as ONE
                    # forward and one weight-gradient launch over both states (functional.dwconv_multi2), their fused tails
                    # (and the two backward passes of those) as ONE launch over both states' candidates
                    stage1, items = stages
                    if j == 0:
                        held.append((items, stage1, xs))
                        continue
                    if not held:
                        out = self._depsep_tail(items, *stage1(xs))
                    else:
                        i0, stage1_0, xs0 = held.pop(0)
                        both = None
                        if len(xs0) == 1 and len(xs) == 1 and xs0[0].shape == xs[0].shape:      # (NORM / DOWN cells: one geometry)
                            both = F.dwconv_multi2(xs0[0], [m[0] for _, _, m in i0], xs[0], [m[0] for _, _, m in items],
                                                   items[0][2][1].training)
                        if both is not None:
                            (z0, s0), (z1, s1) = [[list(col) for col in zip(*side)] for side in both]
                        else:
                            (z0, s0), (z1, s1) = stage1_0(xs0), stage1(xs)
                        if len(i0) + len(items) <= F.MAX_DSTAIL and z0[0].shape == z1[0].shape:
                            out = self._depsep_tail(i0 + items, z0 + z1, s0 + s1)
                        else:
                            out = self._depsep_tail(i0, z0, s0) + self._depsep_tail(items, z1, s1)
                else:
                    out = job(xs)
                for e, p, t in out:
                    terms.setdefault(e, {})[p] = t
            if j == 1:
                for items, stage1, xs0 in held:           # (no partner: on their own)
                    for e, p, t in self._depsep_tail(items, *stage1(xs0)):
                        terms.setdefault(e, {})[p] = t
                del held[:]
            states.append(aliases[-1] if j >= nin else None)

        add_state(self.preprocess0(in0))
        add_state(self.preprocess1(in1))
        offset = 0
        # the node outputs go straight into the (zero-padded) buffer the post-process convolution reads: every node writes its
        # channel slice from its own kernel, the last one also the zero channels behind it (node.bn_combine cat=...) -- no
        # torch.cat launch, and a last node that nothing else reads is stored nowhere else
        catbuf, c_cat = None, 0
        for i in range(nodes):
            node_terms = []
            for j in range(nin + i):
                by_pos = terms.pop(offset + j)
                node_terms += [by_pos[p] for p in sorted(by_pos)]
            offset += nin + i
            cat = None
            ref = next((t.z for t in node_terms if t.z is not None), None)
            if ref is not None and ref.shape[1] % 4 == 0 and self._post_padded(ref.shape[1] * nodes) and (catbuf is not None or i == 0):
                n, c, h, w = ref.shape
                if catbuf is None:
                    catbuf, c_cat = F.new_nhwc(n, 32, h, w, ref), c
                last = i == nodes - 1
                cat = (catbuf, i * c, not last, 32 - nodes * c if last else 0)
            elif catbuf is not None:
                raise F.SenasHipError('search cell: a node without tensor terms behind a concatenation buffer')
            # a node that feeds later nodes is read by their 'identity' candidates, whose BatchNorm2d needs its channel sums
            add_state(F.bn_combine(node_terms, mix=mixes[i], relu=True, out_stats=self.training and i < nodes - 1, cat=cat))
        return self._post(states[nin:nin + nodes], catbuf, c_cat)

    def _post_padded(self, c_in):
        """Does the post-process convolution run on a zero-padded 32-channel concatenation?  (Three 8-channel nodes make 24
        input channels, which fall off the LDS convolution / weight-gradient kernels.)"""
        return self.stacked and c_in % 16 != 0 and c_in <= 32 and c_in % 4 == 0

    def _post(self, outs, catbuf=None, c_cat=0):
        """post_process on the concatenated node outputs.  Three 8-channel nodes make 24 input channels, which fall off
        the LDS convolution / weight-gradient kernels (16-channel passes, 32-row tiles): the concatenation carries 8 zero
        channels and the weight 8 zero input planes (a persistent padded buffer, functional.StackedWeight) -- same
        result, 32 -> 32 kernels.  ``catbuf``: the padded concatenation the nodes have already written (Cell.forward)."""
        post = self.post_process
        c_in = sum(o.shape[1] for o in outs)
        if not self._post_padded(c_in):
            return post(torch.cat(outs, dim=1))
        stacks = self.__dict__.setdefault('_stacks', {})
        key = ('post', id(post.conv))
        if key not in stacks:
            stacks[key] = F.StackedWeight([post.conv.weight], 1, pad_to=32)
        if catbuf is not None:
            x = F.cat_slices(catbuf, c_cat, outs)
        else:
            o0 = outs[0]
            zkey = ('zeros', o0.shape[0], 32 - c_in, o0.shape[2], o0.shape[3], str(o0.device))
            if zkey not in stacks:                           # a constant: made once, only ever read
                stacks[zkey] = torch.zeros((o0.shape[0], 32 - c_in, o0.shape[2], o0.shape[3]), device=o0.device).contiguous(memory_format=torch.channels_last)
            x = torch.cat(list(outs) + [stacks[zkey]], dim=1)
        conv = post.conv
        z, st = F.conv2d(x, stacks[key].tensor(), stride=1, pad=conv.padding[0], dil=1, want_stats=post.norm.training)
        return F.bn_combine([F.Term(z, post.norm, stats=st)])

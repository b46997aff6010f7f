"""Multi-GPU execution of the denoising path: one process per GPU, torch.distributed over RCCL (xGMI).

The reference has no multi-GPU inference path; this is new design:

* **token (sequence) shards** -- every per-token op of the DiT (norms, modulation, all GEMMs, text cross-attention
  with replicated K/V, patch-embed, output head) runs on this rank's contiguous slice of the L tokens; the only
  exchange is an **all-gather of the local K|V** before each self-attention (2.L.D.2 bytes per layer-call in total,
  151 MB at L=12320) and one tiny all-gather of the projected output.  The all-gather is launched asynchronously right
  after the K|V projection so that it overlaps the Q projection.  xGMI is point-to-point (7 links/GPU): RCCL's
  all-gather over a fully connected in-node group uses all links of a GPU concurrently.
  The self-attention itself is split to hide more of it: local keys first (needs nothing from the wire), then the
  gathered keys before / after the own chunk, merged through (O, m, l) partials (`TokenShard.overlap_local`).
* **CFG branches** -- the cond and uncond forwards of a step are independent; with an even number of ranks they run on
  two rank groups that exchange `noise_pred` once per step (no per-layer traffic).

* **interleaved branches** (`mode="interleave"`) -- every rank holds BOTH CFG branches on a 1/world token shard and
  advances them alternately, block by block, on two HIP streams with two communicators: the branches are the only
  independent work of a denoising step, so this is what lets one branch's K|V all-gather fly under the other branch's
  GEMMs and attention instead of standing in front of its own (with one branch per rank nothing can overlap it: the
  gather is ~1.1 ms per layer on one xGMI link pair at 4 GPUs against ~2.8 ms of compute).

* **heads exchange** (`exchange="heads"`, the sequence-parallel all-to-all) -- instead of gathering every rank's K|V,
  the ranks of a token group trade token shards for head shards around the self-attention: each rank projects q | k | v
  of ITS tokens (one fused GEMM, norm + RoPE as on one GPU), an **all-to-all** hands rank j the q | k | v of ALL tokens
  for heads [j.H/P, (j+1).H/P), it attends those heads over the whole sequence (one launch, no partials), and a second
  all-to-all returns the outputs to the token owners.  Bytes sent per rank and layer: (P-1)/P . n . 4D . 2 against
  (P-1) . n . 2D . 2 received by the K|V all-gather -- equal at P = 2, half at 4, a quarter at 8 -- and on the fully
  connected xGMI mesh every pair's link carries exactly its own 1/P share.  Needs heads % P == 0 (24 heads: 2, 4, 8).

`shard_pipeline(pipe, rank, world)` picks cfg x token = 2 x (world/2) for even `world`, else 1 x world; `mode=` selects
the interleaved plan (bench.py probes both on the node and keeps the faster).  Latents and the sampler state are
replicated (2.2 M floats); every rank performs the same CFG+Euler update.
"""
import torch
import torch.distributed as dist


class TokenShard:
    def __init__(self, rank, ways, group=None, force=False, exchange="kv"):
        # force: take the sharded code path (separate K|V / Q projections, all-gather through the communicator) even
        # with one shard -- how a single GPU exercises the RCCL call sequence (tests/test_parallel_gpu.py)
        self.rank, self.ways, self.group, self.force = rank, ways, group, force
        self.exchange = exchange          # "kv": all-gather of K|V; "heads": all-to-all token shards <-> head shards
        self._buf = {}
        import os
        env = os.environ.get("FINO_KV_HEAD_GROUPS")
        if env:
            self.kv_head_groups = max(1, int(env))
        # attend to the LOCAL K/V chunk while the other ranks' chunks are still on the wire, then to what arrived, and
        # merge the partials (fino_attn_partial / fino_attn_merge): hides up to 1/ways of the attention under the gather.

    # attend to the own K/V chunk while the gather is in flight and merge (O, m, l) partials; False = one attention
    # launch over the gathered keys after the wait (bit-identical arithmetic to the unsharded forward); "auto": only with
    # 2 shards -- the partials + merge cost ~0.19 ms per layer-call over the single pass (profiles/
    # r02_heads_exchange_parts.txt) and hide 1/ways of the attention: 0.42 ms of 0.84 at 2 shards, 0.11 of 0.45 at 4
    overlap_local = "auto"

    def local_first(self):
        return self.ways <= 2 if self.overlap_local == "auto" else bool(self.overlap_local)

    # K|V exchange: project q | k | v in ONE GEMM (k | v written straight into the gather's send buffer) instead of K|V
    # first and Q under the gather.  "auto": on for the interleaved plan's shards (ParallelPlan sets it) -- there the other
    # CFG branch's kernels are what the gather overlaps, and two GEMMs of 2 D and D columns cost 185 us where the fused
    # one costs 139 (3080 rows; profiles/r02_heads_exchange_parts.txt).  A split-plan rank keeps the Q projection as the
    # only work it can put under its gather.
    fused_qkv = False

    # tile height this shard's GEMM calls ask for (ops.gemm(..., tile_m=): a per-call argument of the C ABI, never process
    # state): 0 = the library's planner (built for a GEMM alone on the chip); the interleaved plan sets 8 = 256-row tiles
    # only -- its second kernel stream fills the CUs a partial round of tiles leaves idle, and lower tiles only add operand
    # traffic there (tools/plan_sim.py)
    gemm_tile_m = 0

    def tile_m_for(self, rows):
        """the tile height this shard's GEMM calls pass for `rows` rows.  A shard that asked for 256-row tiles (8: the interleaved
        plan) gets 224-row ones (7) when they multiply >= 5 % fewer rows that do not exist.  Since round 6 a RAGGED last tile row of
        256-row tiles runs as a 64- or 128-row tile inside the same launch (csrc/fino_gemm.hip: `GP_RAGGED`), so 3080 rows (4-way shard
        of L = 12320) count as 12 x 256 + 64 and 1540 rows (8-way) as 6 x 256 + 64: 256-row tiles win both (interleave:4 69.0 -> 67.7 ms
        per step and rank against 69.7 - 71.1 with 224-row tiles, interleave:8 40.4 - 40.7 either way; tools/plan_sim.py,
        profiles/r06_ragged_step_ab.txt).  A separate remainder LAUNCH was measured too and loses (a second K walk per GEMM:
        70.3 -> 77.1 ms at 4 ways, profiles/r06_plan_sim_tile_modes.txt)."""
        t = int(self.gemm_tile_m)
        if t != 8 or rows <= 0:
            return t
        rem = rows % 256
        pad8 = rows - rem + (0 if rem == 0 else 64 if rem <= 64 else 128 if rem <= 128 else 256)
        pad7 = -(-rows // 224) * 224
        return 7 if pad7 < 0.95 * pad8 else 8

    def fused_qkv_ok(self):
        return bool(self.fused_qkv)

    # K|V exchange in HEAD GROUPS (round 5): the local K|V leave in this many all-gathers, one per run of heads, and the
    # attention runs group by group as they arrive -- heads are independent in the self-attention, so group g is attended
    # to while group g+1 is still on the wire: of the gather's time only the first group's share (1/groups) stands in
    # front of the attention, the rest flies under it.  Costs one more small launch per layer-call (the K RMSNorm + RoPE
    # kernel scatters k by head into [group][token][k_g | v_g] send blocks, v follows as a scattering copy instead of
    # staying where the projection wrote it) and attention launches of heads / groups heads each (24 heads x 13 q-blocks of a
    # 3080-row shard are 2 rounds of the CUs as one launch and as two).  1 = one gather.  tools/plan_sim.py with a modelled wire
    # (FINO_PLAN_SIM_WIRE_GBPS) measures what it hides.  Round 6: the default is ONE gather -- the only evidence for two is a
    # MODELLED wire (split N = 8: 61.9 -> 58.0 ms at 150 GB/s), and on the interleaved plan it measured a 2.3 ms loss; until a
    # node run shows the win it is opt-in: `FINO_KV_HEAD_GROUPS=2`, or bench.py's plan probe "…-kvg2" which measures it on the
    # real links and lets it win the line.
    kv_head_groups = 1

    def kv_groups(self, heads):
        g = max(1, min(int(self.kv_head_groups), heads))
        return g if (self.exchange == "kv" and not self.local_first()) else 1

    def kv_group_layout(self, heads, dh, lpad, dtype, dev):
        """send side of the head-grouped K|V gather as ONE flat buffer [group][lpad][k_g | v_g] (views[g]: the [lpad, 2 dg]
        block all-gather number g sends; rows [n, lpad) are never written and stay zero) + the per-head tables of
        ops.rmsnorm_rope_scatter: element offset of head h's k / v columns, and the row stride there."""
        groups = self.kv_groups(heads)
        key = ("kv_group_layout", heads, dh, lpad, dtype, str(dev), groups)
        lay = self._buf.get(key)
        if lay is None:
            from types import SimpleNamespace
            cuts = [heads * i // groups for i in range(groups + 1)]
            ranges = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            flat = torch.zeros(lpad * 2 * heads * dh, dtype=dtype, device=dev)
            views, off_k, off_v, ld, base = [], [0] * heads, [0] * heads, [0] * heads, 0
            for h0, h1 in ranges:
                dg = (h1 - h0) * dh
                views.append(flat[base:base + lpad * 2 * dg].view(lpad, 2 * dg))
                for h in range(h0, h1):
                    off_k[h] = base + (h - h0) * dh
                    off_v[h] = base + dg + (h - h0) * dh
                    ld[h] = 2 * dg
                base += lpad * 2 * dg
            t64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)          # noqa: E731
            lay = self._buf[key] = SimpleNamespace(flat=flat, views=views, ranges=ranges, off_k=t64(off_k), off_v=t64(off_v),
                                                   ld=t64(ld))
        return lay

    @property
    def active(self):
        return self.ways > 1 or self.force

    def rows(self, L):
        lpad = (L + self.ways - 1) // self.ways
        lo = self.rank * lpad
        n = max(0, min(L, lo + lpad) - lo)
        if n == 0:
            raise ValueError(f"sequence of {L} tokens is too short for {self.ways} shards")
        return lo, n, lpad

    def _get(self, key, shape, dtype, dev):
        k = (key, tuple(shape), dtype, str(dev))
        b = self._buf.get(k)
        if b is None:
            b = self._buf[k] = torch.zeros(shape, dtype=dtype, device=dev)
        return b

    def kv_local(self, lpad, width, dtype, dev):
        return self._get("kv_loc", (lpad, width), dtype, dev)

    def partial_buf(self, i, floats, dev):
        """fp32 partial buffer number i of the local-first attention (None on CPU: the stand-in ops allocate)"""
        if torch.device(dev).type != "cuda":
            return None
        return self._get(f"part{i}", (floats,), torch.float32, dev)

    def out_local(self, lpad, width, dtype, dev):
        return self._get("out_loc", (lpad, width), dtype, dev)

    # The stream every collective of this shard is ISSUED on, or None = the stream that is current at the call.  The
    # interleaved plan runs its two branches on two side streams; a collective issued with a side stream current cannot be
    # captured into a hipGraph on this image (SIGSEGV inside hipStreamEndCapture), one issued with the CAPTURING stream current
    # can -- also when the kernels around it run on side streams (tools/debug/rccl_capture_probe.py, round 5:
    # profiles/r05_rccl_capture_probe.txt).  So the pipeline names the step's own stream here: that stream waits for the
    # branch's producer kernels (an event), c10d forks its communicator stream from it, and the BRANCH's stream waits for the
    # collective where the forward calls work.wait().  Round 6: the pipeline sets it only around a step that is being
    # CAPTURED and resets it in a `finally` -- an eager interleaved step issues each branch's collectives from that branch's own
    # stream again, so the two branches (and their two communicators) are not coupled through the main stream, and no stale
    # capture stream survives a capture.  Results are bit-identical either way (the kernels and their order do not change).
    issue_stream = None

    def _issue(self, fn, t, async_op, sync_in_capture=False):
        """run the collective `fn()` (-> work handle) with `issue_stream` current; the calling stream waits for it at once unless
        async_op.  -> work handle or None.
        `sync_in_capture` (the all-to-all, round 6): inside a hipGraph capture the collective is issued SYNCHRONOUSLY on the step's
        own stream -- c10d then runs it on that stream itself instead of forking its communicator stream; an asynchronous
        send / receive-based collective (ncclAllToAll, grouped ncclSend / ncclRecv, batch_isend_irecv) segfaults in
        hipStreamEndCapture on this image, the synchronous form captures and replays bit-equal (tools/debug/rccl_capture_probe.py,
        profiles/r06_rccl_capture_probe.txt).  The graph must then be destroyed BEFORE the process group (StepGraph.close)."""
        iss = self.issue_stream
        cur = torch.cuda.current_stream() if (iss is not None and t.is_cuda) else None
        if sync_in_capture and t.is_cuda and torch.cuda.is_current_stream_capturing():
            if cur is None or cur == iss:
                fn(False)
                return None
            iss.wait_stream(cur)
            with torch.cuda.stream(iss):
                fn(False)
            cur.wait_stream(iss)                 # the branch's stream goes on behind the collective
            return None
        if cur is None or cur == iss:
            return fn(async_op)
        iss.wait_stream(cur)                     # what produced `t` is enqueued on the calling (branch) stream
        with torch.cuda.stream(iss):
            work = fn(True)
        if not async_op:
            work.wait()                          # current stream = the branch's again: it waits for the collective
            return None
        return work

    def _all_gather(self, key, t, async_op):
        out = self._get(key, (self.ways * t.shape[0],) + tuple(t.shape[1:]), t.dtype, t.device)
        if dist.get_backend(self.group) == "gloo":           # tests: CPU tensors, or device tensors staged via host
            if t.is_cuda:
                parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.ways)]
                dist.all_gather(parts, t.cpu().contiguous(), group=self.group)
                out.copy_(torch.cat(parts).to(t.device))
                return out, None
            parts = list(out.chunk(self.ways))
            dist.all_gather(parts, t.contiguous(), group=self.group)
            return out, None
        work = self._issue(lambda a_: dist.all_gather_into_tensor(out, t, group=self.group, async_op=a_), t, async_op)
        return out, work

    def heads_exchange_ok(self, heads):
        return self.exchange == "heads" and heads % self.ways == 0

    # the rank's heads travel in this many groups (each its own all-to-all): group g+1 is on the wire while group g is
    # attended to, and group g's outputs return while group g+1 is attended to
    head_groups = 2

    def head_ranges(self, hp):
        """[(h0, h1), ...]: the rank's hp heads cut into min(head_groups, hp) near-equal runs"""
        g = max(1, min(int(self.head_groups), hp))
        cuts = [hp * i // g for i in range(g + 1)]
        return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

    def heads_send_layout(self, heads, dh, lpad, dtype, dev):
        """The send side of the heads exchange as ONE flat buffer: per head group g a [ways, lpad, 3, dg] block (slice j =
        q | k | v of MY tokens for the heads of rank j in that group), plus the per-head destination tables of
        ops.rmsnorm_rope_scatter (element offset of head h's 128 columns for q / k / v, and the row stride there) -- the
        RMSNorm + RoPE kernel writes q and k straight into it and v goes by the same kernel as a scattering copy, instead of
        a permute copy of all three after the in-place kernel.  Rows [n, lpad) are never written and stay zero."""
        key = ("heads_send_layout", heads, dh, lpad, dtype, str(dev), int(self.head_groups))
        lay = self._buf.get(key)
        if lay is None:
            from types import SimpleNamespace
            ways, hp = self.ways, heads // self.ways
            flat = torch.zeros(ways * lpad * 3 * hp * dh, dtype=dtype, device=dev)
            views, off, ld, base = [], [[0] * heads for _ in range(3)], [0] * heads, 0
            for h0, h1 in self.head_ranges(hp):
                dg = (h1 - h0) * dh
                views.append(flat[base:base + ways * lpad * 3 * dg].view(ways, lpad, 3, dg))
                for j in range(ways):
                    for hl in range(h0, h1):
                        for which in range(3):
                            off[which][j * hp + hl] = base + j * lpad * 3 * dg + which * dg + (hl - h0) * dh
                        ld[j * hp + hl] = 3 * dg
                base += ways * lpad * 3 * dg
            lay = self._buf[key] = SimpleNamespace(
                flat=flat, views=views, ld=torch.tensor(ld, dtype=torch.int64, device=dev),
                off=[torch.tensor(o_, dtype=torch.int64, device=dev) for o_ in off],
                off_qkv=torch.tensor(off[0] + off[1] + off[2], dtype=torch.int64, device=dev))
        return lay

    def heads_recv_layout(self, heads, dh, lpad, dtype, dev, key="o_recv"):
        """The RETURN side of the heads exchange as one flat buffer [groups, ways, lpad, dg] when the head groups are equal
        (None otherwise): group g's all-to-all receives into slice g -- registered here under the buffer key all_to_all
        looks up (f"{key}{g}") -- so that the out-projection can read all of it as K blocks (ops.gemm_blocked_a)."""
        ways, hp = self.ways, heads // self.ways
        ranges = self.head_ranges(hp)
        if len({b - a for a, b in ranges}) != 1:
            return None
        dg = (ranges[0][1] - ranges[0][0]) * dh
        k = ("heads_recv_layout", key, heads, dh, lpad, dtype, str(dev), len(ranges))
        flat = self._buf.get(k)
        if flat is None:
            flat = self._buf[k] = torch.zeros(len(ranges), ways, lpad, dg, dtype=dtype, device=dev)
            for g in range(len(ranges)):
                self._buf[(f"{key}{g}", (ways, lpad, dg), dtype, str(dev))] = flat[g]
        return flat

    def all_to_all(self, key, send, async_op=False):
        """send [ways, rows, width] (slice j goes to rank j) -> (received [ways, rows, width]: slice j came from rank j,
        work handle or None).  Without async_op the call is blocking in stream order."""
        recv = self._get(key, tuple(send.shape), send.dtype, send.device)
        if dist.get_backend(self.group) == "gloo" and send.is_cuda:      # tests on one GPU: staged through host memory
            r = torch.empty(send.shape, dtype=send.dtype)
            dist.all_to_all_single(r, send.cpu().contiguous(), group=self.group)
            recv.copy_(r.to(send.device))
            return recv, None
        work = self._issue(lambda a_: dist.all_to_all_single(recv, send, group=self.group, async_op=a_), send, async_op,
                           sync_in_capture=True)
        return recv, (work if async_op else None)

    def a2a_buffer(self, key, shape, dtype, dev):
        return self._get(key, shape, dtype, dev)

    def all_gather_kv(self, kv_loc):
        """-> ([ways*lpad, 2D] buffer, work handle to wait on before the attention launch)."""
        return self._all_gather("kv_all", kv_loc, True)

    def all_gather_out(self, po_loc):
        return self._all_gather("out_all", po_loc, False)[0]


class ParallelPlan:
    def __init__(self, rank, world, cfg_ways, token_ways, token_group, cfg_group, token_group_b=None, exchange="kv"):
        self.rank, self.world = rank, world
        self.cfg_ways, self.token_ways = cfg_ways, token_ways
        self.cfg_idx, self.tok_rank = rank // token_ways, rank % token_ways
        self.token_group, self.cfg_group = token_group, cfg_group
        force = world == 1                    # single-rank rehearsal of the N>1 call sequence
        self.exchange = exchange
        self.token_group_b = token_group_b
        self.shard = TokenShard(self.tok_rank, token_ways, token_group, force, exchange)
        # interleaved plan: a second shard object (own buffers, own communicator) for the second CFG branch
        self.interleave = token_group_b is not None
        self.shards = (self.shard, TokenShard(self.tok_rank, token_ways, token_group_b, force, exchange)) \
            if self.interleave else None
        if self.interleave:
            # the other CFG branch is what flies under a branch's all-to-alls here; cutting the heads into groups only
            # costs (two half-size attention launches: +4-6 % GPU work, tools/plan_sim.py)
            for sh in self.shards:
                sh.head_groups = 1
                sh.fused_qkv = True
                sh.gemm_tile_m = 8           # (a shard's row count may lower it: TokenShard.tile_m_for)
                # two kernel streams already put one branch's attention under the other's gather: head groups add 2.3 ms of
                # launches per step there and hide nothing more (tools/plan_sim.py with a modelled wire, profiles/r05_plan_sim*)
                sh.kv_head_groups = 1
        self._buf = None

    def with_exchange(self, exchange):
        """the same ranks, groups and communicators with the other self-attention exchange (bench.py probes both)"""
        return ParallelPlan(self.rank, self.world, self.cfg_ways, self.token_ways, self.token_group, self.cfg_group,
                            self.token_group_b, exchange)

    def with_kv_groups(self, groups):
        """the same plan with the K|V all-gather cut into `groups` head groups (TokenShard.kv_head_groups; bench.py probes it)"""
        p = self.with_exchange("kv")
        for sh in (p.shards or (p.shard,)):
            sh.kv_head_groups = int(groups)
        p.kv_groups = int(groups)
        return p

    kv_groups = 1

    @property
    def desc(self):
        tail = "-heads" if self.exchange == "heads" else ("" if self.kv_groups <= 1 else f"-kvg{self.kv_groups}")
        if self.interleave:
            return f"token{self.token_ways}x2branches-interleaved{tail}"
        return f"cfg{self.cfg_ways}xtoken{self.token_ways}{tail}"

    def exchange_cfg(self, mine):
        """all-gather of the two CFG branches' predictions inside the pair group -> (cond_pred, uncond_pred)."""
        if self._buf is None or self._buf.shape[1:] != mine.shape or self._buf.dtype != mine.dtype:
            self._buf = torch.empty((2,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        if dist.get_backend(self.cfg_group) == "gloo":
            if mine.is_cuda:                                 # tests on one GPU: stage through host memory
                parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(2)]
                dist.all_gather(parts, mine.cpu().contiguous(), group=self.cfg_group)
                self._buf.copy_(torch.stack(parts).to(mine.device))
            else:
                dist.all_gather([self._buf[0], self._buf[1]], mine.contiguous(), group=self.cfg_group)
        else:
            dist.all_gather_into_tensor(self._buf, mine.contiguous(), group=self.cfg_group)
        return self._buf[0], self._buf[1]


def make_plan(rank, world, cfg_parallel=True, mode="split", allow_single=False, exchange="kv"):
    """`allow_single`: build the plan for world == 1 too (the sharded code path forced through real communicators of
    one rank -- the rehearsal a 1-GPU box can run)."""
    if mode == "interleave":
        if world < 2 and not allow_single:
            raise ValueError("the interleaved plan needs at least 2 ranks")
        ga = dist.new_group(list(range(world)))      # one communicator per branch: their collectives are independent
        gb = dist.new_group(list(range(world)))
        return ParallelPlan(rank, world, 1, world, ga, None, token_group_b=gb, exchange=exchange)
    cfg_ways = 2 if (cfg_parallel and world % 2 == 0) else 1
    token_ways = world // cfg_ways
    token_group = cfg_group = None
    # every rank creates every group, in the same order
    for c in range(cfg_ways):
        ranks = list(range(c * token_ways, (c + 1) * token_ways))
        g = dist.new_group(ranks) if (token_ways > 1 or allow_single) else None
        if rank in ranks:
            token_group = g
    for t in range(token_ways):
        ranks = [t + c * token_ways for c in range(cfg_ways)]
        g = dist.new_group(ranks) if cfg_ways > 1 else None
        if rank in ranks:
            cfg_group = g
    return ParallelPlan(rank, world, cfg_ways, token_ways, token_group, cfg_group, exchange=exchange)


def shard_pipeline(pipe, rank, world, cfg_parallel=True, mode="split", plan=None, allow_single=False, exchange="kv"):
    plan = plan or make_plan(rank, world, cfg_parallel, mode, allow_single, exchange)
    heads = getattr(getattr(pipe.transformer, "config", None), "num_attention_heads", None)
    if plan.exchange == "heads" and heads is not None and heads % plan.token_ways != 0:
        raise ValueError(f"the heads exchange needs num_attention_heads ({heads}) divisible by the token shards "
                         f"({plan.token_ways})")
    pipe.parallel = plan
    pipe.parallel_desc = plan.desc
    pipe.token_shards = plan.token_ways
    pipe.transformer.parallel = plan.shard if plan.shard.active else None
    return plan


def sharded_vae_decode(vae, z, rank, world, group=None):
    """The Wan VAE decode on `world` ranks (round 6): every rank runs the blocks up to the last temporal upsampling (26 % of the
    decode's FLOPs at 704 x 1280; no attention after them), then the tail -- up_blocks.2 / 3 + head, 74 %, local 3 x 3 x 3
    convolutions -- on ITS horizontal slab of the frame plus the halo rows its 15 convolutions need (`AutoencoderKLWan.decode_slab`:
    recomputed, not exchanged), and ONE all-gather of the finished video slabs puts the clip on every rank.  Bit-identical to
    `vae.decode(z)`.  Reference: architecture/autoencoder_kl_wan.py:1198-1227 (decode), :783-909 (decoder); the reference has no
    multi-GPU path.  -> video [1, C, T, H, W]."""
    out, (r0, r1, per, height) = vae.decode_slab(z, rank, world)
    probe = out if out is not None else None
    if probe is None:                               # a slab past the frame's last row (world > rows): an empty contribution
        raise ValueError(f"{world} slabs for a frame of {height} rows: fewer ranks than that, please")
    _, c, t, _, w = out.shape
    send = out.new_zeros((1, c, t, per, w))
    send[:, :, :, :r1 - r0] = out
    allv = out.new_empty((world, 1, c, t, per, w))
    g = group if group is not None else dist.group.WORLD
    if dist.get_backend(g) == "gloo" and out.is_cuda:           # tests: staged through the host
        parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)]
        dist.all_gather(parts, send.cpu().contiguous(), group=g)
        allv.copy_(torch.stack(parts).to(out.device))
    elif dist.get_backend(g) == "gloo":
        dist.all_gather(list(allv.unbind(0)), send.contiguous(), group=g)
    else:
        dist.all_gather_into_tensor(allv, send.contiguous(), group=g)
    return allv.permute(1, 2, 3, 0, 4, 5).reshape(1, c, t, world * per, w)[:, :, :, :height].contiguous()


def sharded_vae_encode(vae, x, rank, world, group=None):
    """The Wan VAE encode of one video on `world` ranks (round 6): conv_in and the first two down blocks -- 73 % of the encoder's
    FLOPs at 704 x 1280, all local 3 x 3 (x 3) convolutions -- run on the rank's horizontal slab of the frame plus a recomputed halo
    (`AutoencoderKLWan.encode_slab`), ONE all-gather of the slabs' activations (a quarter of the frame's rows, 320 channels) puts
    the whole tensor on every rank, and the rest (two down blocks, the mid block with its attention, the head) runs replicated.
    Bit-identical to `vae.encode(x)`.  Reference: architecture/autoencoder_kl_wan.py:1145-1169 (encode), :505-623 (encoder); the
    reference has no multi-GPU path.  -> what `vae.encode(x)` returns (`.latent_dist`)."""
    part, (a, b, per, hk) = vae.encode_slab(x, rank, world)
    g = group if group is not None else dist.group.WORLD
    # (every rank can derive the activation's shape from its own slab; a rank whose slab is empty asks the others: rare -- more
    # ranks than rows -- and refused)
    if part is None:
        raise ValueError(f"{world} slabs for {hk} rows of encoder activation: fewer ranks than that, please")
    t, _, w, c = part.shape
    send = part.new_zeros((t, per, w, c))
    send[:, :b - a] = part
    allv = part.new_empty((world, t, per, w, c))
    if dist.get_backend(g) == "gloo" and part.is_cuda:           # tests: staged through the host
        parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)]
        dist.all_gather(parts, send.cpu().contiguous(), group=g)
        allv.copy_(torch.stack(parts).to(part.device))
    elif dist.get_backend(g) == "gloo":
        dist.all_gather(list(allv.unbind(0)), send.contiguous(), group=g)
    else:
        dist.all_gather_into_tensor(allv, send.contiguous(), group=g)
    full = allv.permute(1, 0, 2, 3, 4).reshape(t, world * per, w, c)[:, :hk].contiguous()
    return vae.encode_resume(full)

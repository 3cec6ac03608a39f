"""Sequence-parallel (Ulysses) DiT forward: one clip's forward spread over P ranks — SURVEY §8e axis 3.

The reference's own design for this is USP (pipelines/svi_video.py:119-135: chunk the token axis, all_gather after the head;
distributed/xdit_context_parallel.py: all-to-all around attention).  Here:

  * rank r owns token rows [r*Ls, (r+1)*Ls) of the (f h w) sequence, Ls = L / P, for everything row-local: LN / modulation,
    q k v projections, RMSNorm + RoPE (at the rows' true grid positions), output projection, cross-attention, MLP, head;
  * around self-attention tokens are traded for heads: an all-to-all turns [Ls tokens, H heads] into [L tokens, H/P heads],
    attention runs on whole sequences of a head group, a second all-to-all turns the result back;
  * after the head an all-gather of [Ls, 64] rows rebuilds the latent.

Layouts are chosen so that NEITHER side of an exchange runs a packing pass (r1 had six layout copies per block):
    q, k   leave the RMSNorm+RoPE launch in send order  [G][P(dest)][Ls][Dg]   (svi_dit_sp_block_qkv, SviScatter), so one
           (operand, head group) is a contiguous all-to-all input, and what arrives, [P(src)][Ls][Dg], IS the token-major
           [L, Dg] operand the attention kernel reads (row stride Dg);
    V^T    [D, ldvt] as the swapped projection GEMM writes it: row block j is rank j's piece; what arrives, [P(src)][Dp][ldvt],
           is put side by side along the token axis by ONE copy kernel (svi_sp_unpack_vt) — the only pass the exchange adds;
    out    the attention kernel writes [G][L][Dg] = [G][P(dest)][Ls][Dg]: contiguous per destination; what arrives is turned
           into [Ls, D] token rows by one copy kernel (svi_sp_unpack_out).
All exchange buffers are allocated once per (rank, world, tokens) and reused by every block of every forward.  With G > 1 head
groups the exchange of group g+1 is issued (async collective) before the attention of group g is enqueued, so transport overlaps
attention; G is chosen so that a group still fills the chip (>= 256 workgroups of 256 query rows).

Per block and rank the exchanges move 3*Ls*D/P*(P-1) + Ls*D/P*(P-1) bf16 values: at C2 on 4 ranks 2 x 75 MB out and back per
block over xGMI.  Constraints: L % P == 0 and heads % P == 0 (1.3B: 12 heads -> P in {2,3,4,6}; 14B: 40 heads -> {2,4,5,8}).
Composes with CfgPair: 8 GPUs = 2 x 4.

When the heads do NOT divide over the ranks (1.3B on 8 ranks) the reference's USP falls back to ring attention (xfuser's hybrid,
distributed/xdit_context_parallel.py:108-129).  The fallback here is `mode="gather"`: a rank keeps its Ls query rows of ALL heads and
all-gathers K and V^T of all tokens (2*L*D*(P-1)/P values per block and rank — twice the Ulysses volume, any P that divides L),
attention runs with Lq = Ls against Lk = L, and no second exchange is needed.  Same kernels; rows are grouped into 64-row wavefronts
from the shard's first row, so the deferred-rescale events of the softmax (taken per wavefront) can fall differently than on one rank:
results agree within the attention kernel's stated tolerance, bit for bit when Ls is a multiple of 256.

Every arithmetic kernel is the one the single-GPU forward uses, on the same operands per row / per head, so the result is
bit-identical to `WanDiT.forward` for any P and G (tests/test_gpu_sp.py runs P = 1, 2, 4 shards on one GPU with the exchange
simulated in-process and with real process groups; tests/test_dist_gloo.py runs the layout algebra over gloo on CPU tensors).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib as L
from .dit import WanDiT


# ---- the layouts, stated in plain tensor algebra (device-agnostic; the CPU tests run the exchange algebra on these, the GPU tests
# ---- hold the kernels to them) ----------------------------------------------------------------------------------------------
def send_layout_qk(x: torch.Tensor, P: int, G: int = 1) -> torch.Tensor:
    """x [Ls, D] (q or k rows of this rank) -> [G, P, Ls*Dg]: what svi_dit_sp_block_qkv stores."""
    Ls, D = x.shape
    Dg = D // P // G
    return x.reshape(Ls, P, G, Dg).permute(2, 1, 0, 3).reshape(G, P, Ls * Dg).contiguous()


def unpack_vt(recv: torch.Tensor, Ls: int) -> torch.Tensor:
    """recv [P(src), Dp, lds] -> V^T [Dp, L8] (zero beyond L): what svi_sp_unpack_vt writes."""
    P, Dp, _ = recv.shape
    Lfull = P * Ls
    vt = torch.zeros((Dp, (Lfull + 7) // 8 * 8), dtype=recv.dtype, device=recv.device)
    vt[:, :Lfull] = recv[:, :, :Ls].permute(1, 0, 2).reshape(Dp, Lfull)
    return vt


def unpack_out(recv: torch.Tensor, Ls: int) -> torch.Tensor:
    """recv [G, P(src), Ls*Dg] -> attn [Ls, D], source j's head block at columns [j*Dp, (j+1)*Dp): what svi_sp_unpack_out writes."""
    G, P, n = recv.shape
    Dg = n // Ls
    return recv.reshape(G, P, Ls, Dg).permute(2, 1, 0, 3).reshape(Ls, P * G * Dg).contiguous()


# The stacked CFG pair (svi_dit_sp_begin_pair): a rank's rows are [branch][Ls]; on the wire the two branches sit side by side per token, so that after the
# first exchange a head group's operand is token-major [L, 2 Dg] — the unconditional branch's heads are simply MORE HEADS of the same attention launch.
def send_layout_qk_pair(x2: torch.Tensor, P: int, G: int = 1) -> torch.Tensor:
    """x2 [2 Ls, D] (q or k rows of this rank, conditional branch on top) -> [G, P, Ls*2*Dg] = [group][destination][row][branch][Dg]:
    what svi_dit_sp_block_qkv_part stores for the pair."""
    Ls, D = x2.shape[0] // 2, x2.shape[1]
    Dg = D // P // G
    return x2.reshape(2, Ls, P, G, Dg).permute(3, 2, 1, 0, 4).reshape(G, P, Ls * 2 * Dg).contiguous()


def unpack_vt_pair(recv: torch.Tensor, Ls: int, G: int = 1) -> torch.Tensor:
    """recv [P(src), Dp, lds2] (a source's V^T piece: branch b's tokens at columns [b Ls, (b + 1) Ls)) -> V^T [G, 2, Dg, L8] (zero beyond L),
    rows ordered [group][branch][Dg] like the q | k columns: what svi_sp_unpack_vt writes with nb = 2."""
    P, Dp, _ = recv.shape
    Dg, Lfull = Dp // G, P * Ls
    vt = torch.zeros((G, 2, Dg, (Lfull + 7) // 8 * 8), dtype=recv.dtype, device=recv.device)
    for b in range(2):
        vt[:, b, :, :Lfull] = recv[:, :, b * Ls:(b + 1) * Ls].reshape(P, G, Dg, Ls).permute(1, 2, 0, 3).reshape(G, Dg, Lfull)
    return vt


def unpack_out_pair(recv: torch.Tensor, Ls: int) -> torch.Tensor:
    """recv [G, P(src), Ls*2*Dg] = [group][source][row][branch][Dg] -> attn [2 Ls, D] (unconditional rows below the conditional ones), source j's head
    block at columns [j*Dp, (j+1)*Dp): what svi_sp_unpack_out writes with nb = 2."""
    G, P, n = recv.shape
    Dg = n // (2 * Ls)
    return recv.reshape(G, P, Ls, 2, Dg).permute(3, 2, 1, 0, 4).reshape(2 * Ls, P * G * Dg).contiguous()


def head_groups(heads_local: int, tokens: int, nb: int = 1) -> int:
    """Number of head groups the exchange is pipelined in: as many as possible while one group's attention still fills the chip
    (256 CUs x one 256-row query block each) and divides the rank's heads.  nb = 2 (stacked CFG pair): a group's launch carries both branches' heads."""
    blocks_per_head = (tokens + 255) // 256
    best = 1
    for g in range(1, heads_local + 1):
        if heads_local % g == 0 and nb * (heads_local // g) * blocks_per_head >= 256:
            best = g
    return best


def _staged(t: torch.Tensor, group) -> bool:
    """gloo has no device all-to-all: a gloo group moves GPU tensors through the host (tests on one GPU; RCCL is the real path)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_to_all(send: torch.Tensor, group=None) -> torch.Tensor:
    """send [P, ...] -> recv [P, ...]: recv[i] = rank i's send[my rank]   (RCCL on GPUs, gloo in the tests)."""
    if _staged(send, group):
        host = send.cpu().contiguous()
        out = torch.empty_like(host)
        dist.all_to_all_single(out, host, group=group)
        return out.to(send.device)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send.contiguous(), group=group)
    return recv


def all_gather_rows(rows: torch.Tensor, group=None) -> torch.Tensor:
    world = dist.get_world_size(group)
    src = rows.cpu() if _staged(rows, group) else rows
    gathered = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(gathered, src.contiguous(), group=group)
    return torch.cat(gathered, dim=0).to(rows.device)


def all_to_all_local(sends: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """The all-to-all among P shards living in one process: sends[i] is [..., P, n] with the peer axis second to last;
    returns recv with recv[j][..., i, :] = sends[i][..., j, :]."""
    P = len(sends)
    return [torch.stack([sends[i][..., j, :] for i in range(P)], dim=-2) for j in range(P)]


# ---- one rank's share --------------------------------------------------------------------------------------------------------
class _Buffers:
    """Exchange buffers of one (rank, world, tokens, head groups) configuration; allocated once, reused by every block."""

    def __init__(self, d: WanDiT, world: int, Ls: int, G: int, dev, nb: int = 1):
        """nb = 2: the stacked CFG pair (SequenceShard.begin_pair).  A token's two branches travel side by side: every q | k / output piece row is
        [branch][Dg], so what arrives per head group is token-major [L, nb * Dg] and the two branches are nb x as many heads of ONE attention launch
        (P = 4 of the 1.3B model: 6 head-equivalents x 128 query blocks = 768 work items = three whole rounds of 256 CUs, where one branch's 384 are one
        and a half); V^T carries both branches' token rows side by side along the token axis; attn has the unconditional rows below the conditional."""
        D, Dp = d.dim, d.dim // world
        Dg, Lfull = Dp // G, Ls * world
        self.nb = nb
        self.ldvt, self.L8 = (nb * Ls + 7) // 8 * 8, (Lfull + 7) // 8 * 8
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.qk_send = torch.empty((2, G, world, Ls * nb * Dg), **bf)  # [q|k][group][dest][rows of this rank x branch x Dg]
        self.qk_recv = torch.empty((2, G, world, Ls * nb * Dg), **bf)  # [q|k][group][src] = [q|k][group] x token-major [L, nb * Dg]
        self.vt_send = torch.zeros((world, Dp, self.ldvt), **bf)       # = V^T [D, ldvt] of this rank's rows (pad columns stay zero)
        self.vt_recv = torch.empty((world, Dp, self.ldvt), **bf)
        self.vt_full = torch.zeros((G, nb * Dg, self.L8), **bf)        # V^T of this rank's head block over all tokens: per group [branch][Dg] channel rows (pad stays zero)
        self.o_send = torch.empty((G, world, Ls * nb * Dg), **bf)      # attention output [group][L, nb * Dg] = [group][dest][Ls * nb * Dg]
        self.o_recv = torch.empty((G, world, Ls * nb * Dg), **bf)
        self.attn = torch.empty((nb * Ls, D), **bf)
        self.head_rows = None


class _GatherBuffers:
    """Buffers of the K / V all-gather mode (heads need not divide): q | k rows of this rank, K and V^T of all tokens."""

    def __init__(self, d: WanDiT, world: int, Ls: int, dev):
        D, Lfull = d.dim, Ls * world
        self.ldvt, self.L8 = (Ls + 7) // 8 * 8, (Lfull + 7) // 8 * 8
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.q = torch.empty((Ls, D), **bf)
        self.k = torch.empty((Ls, D), **bf)
        self.vt = torch.zeros((D, self.ldvt), **bf)
        self.k_all = torch.empty((world, Ls, D), **bf)                 # = K [L, D], token-major
        self.vt_all = torch.empty((world, D, self.ldvt), **bf)
        self.vt_full = torch.zeros((D, self.L8), **bf)
        self.attn = torch.empty((Ls, D), **bf)
        self.head_rows = None


class SequenceShard:
    def __init__(self, dit: WanDiT, rank: int, world: int, groups: Optional[int] = None, mode: Optional[str] = None):
        """mode "ulysses" (heads traded for tokens; needs heads % world == 0), "gather" (K / V^T all-gathered; any world that divides
        the tokens), or None: ulysses when the heads divide, else gather."""
        if mode is None:
            mode = "ulysses" if dit.num_heads % world == 0 else "gather"
        if mode not in ("ulysses", "gather"):
            raise ValueError(f"unknown sequence-parallel mode {mode!r}")
        if mode == "ulysses" and dit.num_heads % world:
            raise ValueError(f"{dit.num_heads} heads do not divide over {world} ranks")
        self.dit, self.rank, self.world, self.mode = dit, rank, world, mode
        self.Dp = dit.dim // world if mode == "ulysses" else dit.dim
        self.heads_local = dit.num_heads // world if mode == "ulysses" else dit.num_heads
        if mode == "gather":
            groups = 1
        if groups is not None and (groups < 1 or self.heads_local % groups):
            raise ValueError(f"{self.heads_local} heads per rank do not split into {groups} head groups")
        self._groups = groups

    def _setup(self, x, contexts, clip_feature, y, add_condition, nb: int):
        d = self.dit
        d.check_inputs(x, contexts, clip_feature, y, add_condition)
        x = x.to(torch.bfloat16).contiguous()
        B, _, T, H, W = x.shape
        if B != 1:
            raise ValueError("sequence-parallel forward takes one sample (batch over samples with clip sharding instead)")
        self.T, self.H, self.W = T, H, W
        self.L = d.tokens(T, H, W)
        if self.L % self.world:
            raise ValueError(f"{self.L} tokens do not divide over {self.world} ranks")
        self.Ls = self.L // self.world
        self.nb = nb
        self.G = self._groups if self._groups is not None else head_groups(self.heads_local, self.L, nb)
        self.Dg = self.Dp // self.G
        cache = d.__dict__.setdefault("_sp_buffers", {})
        key = (self.rank, self.world, self.L, self.G if self.mode == "ulysses" else "gather", x.device, nb)
        if key not in cache:
            if len(cache) >= 4:
                cache.clear()
            cache[key] = _Buffers(d, self.world, self.Ls, self.G, x.device, nb) if self.mode == "ulysses" else _GatherBuffers(d, self.world, self.Ls, x.device)
        self.buf = cache[key]
        return x

    def begin(self, x, timestep, context, clip_feature=None, y=None, add_condition=None):
        d = self.dit
        if not d.has_image_input:
            clip_feature = None
        x = self._setup(x, (context,), clip_feature, y, add_condition, 1)
        T, H, W = self.T, self.H, self.W
        context, clip_feature = d._prompt_args(context, clip_feature)
        self._keep = [x, context, timestep.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous(), clip_feature,
                      None if y is None else y.to(torch.bfloat16).contiguous(),
                      None if add_condition is None else add_condition.to(torch.bfloat16).contiguous()]
        x, context, ts, clip, yy, addc = self._keep
        L.check(L.lib().svi_dit_sp_begin(d._h, L.ptr(x), L.ptr(ts), L.ptr(context), L.ptr(clip), L.ptr(yy), L.ptr(addc), T, H, W,
                                         context.shape[1], self.rank * self.Ls, self.Ls, L.current_stream()), "svi_dit_sp_begin")

    def begin_pair(self, x, timestep, context_cond, context_uncond, clip_feature=None, y=None, add_condition=None):
        """Both forwards of a CFG step on this shard, stacked (svi_dit_sp_begin_pair): the calls that follow act on 2 Ls rows — the conditional
        branch's on top — and every exchange block carries a leading branch axis.  Ulysses mode, context cache on."""
        d = self.dit
        if self.mode != "ulysses":
            raise ValueError("the stacked CFG pair is served in ulysses mode (heads divide over the ranks)")
        if not d._ctx_cache_on:
            raise RuntimeError("the stacked CFG pair needs the context cache (WanDiT.context_cache(True)): each prompt's K / V live in buffers of their own")
        if context_cond.shape != context_uncond.shape:
            raise ValueError("the two prompt embeddings of a CFG pair must have the same shape")
        if not d.has_image_input:
            clip_feature = None
        x = self._setup(x, (context_cond, context_uncond), clip_feature, y, add_condition, 2)
        T, H, W = self.T, self.H, self.W
        context_cond, context_uncond, clip_feature = d._prompt_args(context_cond, context_uncond, clip_feature)
        self._keep = [x, context_cond, context_uncond, timestep.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous(), clip_feature,
                      None if y is None else y.to(torch.bfloat16).contiguous(),
                      None if add_condition is None else add_condition.to(torch.bfloat16).contiguous()]
        x, ca, cb, ts, clip, yy, addc = self._keep
        L.check(L.lib().svi_dit_sp_begin_pair(d._h, L.ptr(x), L.ptr(ts), L.ptr(ca), L.ptr(cb), L.ptr(clip), L.ptr(yy), L.ptr(addc), T, H, W,
                                              ca.shape[1], self.rank * self.Ls, self.Ls, L.current_stream()), "svi_dit_sp_begin_pair")

    # ---- gather mode: q stays, K / V^T of all ranks are gathered ---------------------------------------------------------------
    def block_qkv_rows(self, layer: int) -> None:
        """-> buf.q, buf.k ([Ls, dim] token rows, RMSNorm + RoPE applied, q pre-scaled), buf.vt (V^T [dim, ldvt]) of this rank's rows
        (svi_dit_sp_block_qkv with one destination and one head group: the send order IS the plain row-major order)."""
        b = self.buf
        L.check(L.lib().svi_dit_sp_block_qkv(self.dit._h, layer, L.ptr(b.q), L.ptr(b.k), L.ptr(b.vt), b.ldvt, 1, 1, L.current_stream()), "svi_dit_sp_block_qkv")

    def block_v_rows(self, layer: int) -> None:
        """First half of block_qkv_rows: LN + modulate and V^T of this rank's rows -> buf.vt (then on its way to the other ranks)."""
        b = self.buf
        L.check(L.lib().svi_dit_sp_block_qkv_part(self.dit._h, layer, L.ptr(b.q), L.ptr(b.k), L.ptr(b.vt), b.ldvt, 1, 1, 1, L.current_stream()), "svi_dit_sp_block_qkv_part")

    def block_qk_rows(self, layer: int) -> None:
        """Second half: q | k of this rank's rows (RMSNorm + RoPE applied) -> buf.q, buf.k; the same kernels in the same order per element as
        block_qkv_rows, so the same bits."""
        b = self.buf
        L.check(L.lib().svi_dit_sp_block_qkv_part(self.dit._h, layer, L.ptr(b.q), L.ptr(b.k), L.ptr(b.vt), b.ldvt, 1, 1, 2, L.current_stream()), "svi_dit_sp_block_qkv_part")

    def attention_rows(self) -> None:
        """This rank's query rows against the gathered K (buf.k_all = [L, dim]) and V^T (buf.vt_all -> buf.vt_full) -> buf.attn [Ls, dim]."""
        b = self.buf
        L.check(L.lib().svi_sp_unpack_vt(L.ptr(b.vt_all), L.ptr(b.vt_full), self.world, self.dit.dim, self.Ls, b.ldvt, b.L8, 1, self.dit.dim, L.current_stream()), "svi_sp_unpack_vt")
        L.check(L.lib().svi_attention_vt_fwd(L.ptr(b.q), self.dit.dim, L.ptr(b.k_all), self.dit.dim, L.ptr(b.vt_full), b.L8, L.ptr(b.attn), self.dit.dim,
                                             self.Ls, self.L, self.dit.num_heads, 1, L.current_stream()), "svi_attention_vt_fwd")

    def block_rest_rows(self, layer: int) -> None:
        L.check(L.lib().svi_dit_sp_block_rest(self.dit._h, layer, L.ptr(self.buf.attn), L.current_stream()), "svi_dit_sp_block_rest")

    # ---- ulysses mode ----------------------------------------------------------------------------------------------------------------
    def block_qkv(self, layer: int) -> None:
        """-> buf.qk_send (q | k in send order, per branch), buf.vt_send (V^T, row block j = rank j's piece; a stacked pair's branches side by side
        along the token axis)."""
        b = self.buf
        L.check(L.lib().svi_dit_sp_block_qkv(self.dit._h, layer, L.ptr(b.qk_send[0]), L.ptr(b.qk_send[1]), L.ptr(b.vt_send), b.ldvt,
                                             self.world, self.G, L.current_stream()), "svi_dit_sp_block_qkv")

    def unpack_v(self) -> None:
        b = self.buf
        L.check(L.lib().svi_sp_unpack_vt(L.ptr(b.vt_recv), L.ptr(b.vt_full), self.world, self.Dp, self.Ls, b.ldvt, b.L8, self.nb, self.Dg,
                                         L.current_stream()), "svi_sp_unpack_vt")

    def attention(self, g: int) -> None:
        """Attention of head group g on the received operands -> buf.o_send[g] ([L, nb * Dg], contiguous per destination).  A stacked pair's two branches
        are 2 x as many heads of this one launch."""
        b = self.buf
        ld = self.nb * self.Dg
        L.check(L.lib().svi_attention_vt_fwd(L.ptr(b.qk_recv[0, g]), ld, L.ptr(b.qk_recv[1, g]), ld, L.ptr(b.vt_full[g]), b.L8,
                                             L.ptr(b.o_send[g]), ld, self.L, self.L, self.nb * self.heads_local // self.G, 1, L.current_stream()), "svi_attention_vt_fwd")

    def block_rest(self, layer: int) -> None:
        b = self.buf
        L.check(L.lib().svi_sp_unpack_out(L.ptr(b.o_recv), L.ptr(b.attn), self.world, self.G, self.Ls, self.Dg, self.nb, L.current_stream()), "svi_sp_unpack_out")
        L.check(L.lib().svi_dit_sp_block_rest(self.dit._h, layer, L.ptr(b.attn), L.current_stream()), "svi_dit_sp_block_rest")

    def tea(self, mode: int, residual: Optional[torch.Tensor] = None) -> None:
        """TeaCache on this shard's rows: 0 snapshot before the blocks, 1 residual [Ls, dim] = x_after - x_before, 2 x += residual."""
        if mode and (residual is None or tuple(residual.shape) != (self.Ls, self.dit.dim) or residual.dtype != torch.bfloat16 or not residual.is_contiguous()):
            raise ValueError("TeaCache residual of a shard must be a contiguous bf16 [Ls, dim] tensor")
        L.check(L.lib().svi_dit_sp_tea(self.dit._h, mode, L.ptr(residual), L.current_stream()), "svi_dit_sp_tea")

    def head(self) -> torch.Tensor:
        """-> head rows of this shard [nb * Ls, ld] (a stacked pair: the unconditional branch's rows below the conditional one's)."""
        ld = L.lib().svi_dit_head_ld(self.dit._h)
        b = self.buf
        if b.head_rows is None or tuple(b.head_rows.shape) != (self.nb * self.Ls, ld):
            b.head_rows = torch.empty((self.nb * self.Ls, ld), dtype=torch.bfloat16, device=b.attn.device)
        L.check(L.lib().svi_dit_sp_head(self.dit._h, L.ptr(b.head_rows), L.current_stream()), "svi_dit_sp_head")
        return b.head_rows

    def unpatchify(self, head_rows: torch.Tensor) -> torch.Tensor:
        out = torch.empty((1, self.dit.out_dim, self.T, self.H, self.W), dtype=torch.bfloat16, device=head_rows.device)
        head_rows = head_rows.contiguous()
        L.check(L.lib().svi_dit_unpatchify(self.dit._h, L.ptr(head_rows), L.ptr(out), self.T, self.H, self.W, L.current_stream()), "svi_dit_unpatchify")
        return out


# ---- drivers -----------------------------------------------------------------------------------------------------------------
def _exchange(recv: torch.Tensor, send: torch.Tensor, group, overlap: bool):
    """all-to-all of [P, n] pieces.  RCCL: an async collective on the communicator's stream (the caller waits on the returned
    handle where the data is needed, so the next kernels enqueue meanwhile); gloo (tests): staged through the host, synchronous."""
    if _staged(send, group):
        host = send.cpu().contiguous()
        out = torch.empty_like(host)
        dist.all_to_all_single(out, host, group=group)
        recv.copy_(out)
        return None
    return dist.all_to_all_single(recv, send, group=group, async_op=overlap)


def _all_gather_into(out: torch.Tensor, mine: torch.Tensor, group, overlap: bool = False):
    """out [P, ...] <- every rank's `mine` (RCCL all-gather; gloo in the tests goes through the host).  overlap: an async collective on the
    communicator's stream — the caller enqueues more kernels and waits on the returned handle where the data is needed."""
    if _staged(mine, group):
        host = mine.cpu().contiguous()
        got = [torch.empty_like(host) for _ in range(out.shape[0])]
        dist.all_gather(got, host, group=group)
        out.copy_(torch.stack(got))
        return None
    return dist.all_gather_into_tensor(out, mine.contiguous(), group=group, async_op=overlap)


def forward_distributed(dit: WanDiT, x, timestep, context, group=None, groups: Optional[int] = None, tea_mode: int = 0,
                        residual: Optional[torch.Tensor] = None, mode: Optional[str] = None, **cond) -> torch.Tensor:
    """model_fn_wan_video(..., use_unified_sequence_parallel=True) for this rank of `group`: every rank passes the same inputs
    and receives the full output.  Per block: the q | k | V^T exchange pipelined over G head groups against attention, the output
    exchange pipelined the same way; one all-gather per forward.
    tea_mode / residual (TeaCache, as WanDiT.forward): 1 = also write this rank's residual rows [Ls, dim]; 2 = skip the blocks and add
    `residual` — no exchange at all in that forward besides the final all-gather."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sh = SequenceShard(dit, rank, world, groups, mode)
    sh.begin(x, timestep, context, **cond)
    b, G = sh.buf, sh.G
    if tea_mode == 2:
        sh.tea(2, residual)
        return sh.unpatchify(all_gather_rows(sh.head(), group))
    if tea_mode == 1:
        sh.tea(0)
    if sh.mode == "gather":
        for layer in range(dit.num_layers):
            sh.block_v_rows(layer)                                   # V^T first: its all-gather runs under the q | k projection and RoPE
            hv = _all_gather_into(b.vt_all, b.vt, group, True)
            sh.block_qk_rows(layer)
            hk = _all_gather_into(b.k_all, b.k, group, True)
            for h_ in (hv, hk):
                if h_ is not None:
                    h_.wait()
            sh.attention_rows()
            sh.block_rest_rows(layer)
        if tea_mode == 1:
            sh.tea(1, residual)
        return sh.unpatchify(all_gather_rows(sh.head(), group))
    _blocks_distributed(sh, group)
    if tea_mode == 1:
        sh.tea(1, residual)
    return sh.unpatchify(all_gather_rows(sh.head(), group))


def forward_local(dits: Sequence[WanDiT], x, timestep, context, groups: Optional[int] = None, mode: Optional[str] = None, **cond) -> torch.Tensor:
    """The same schedule with P = len(dits) shards in ONE process (each shard needs its own handle: a handle holds one
    workspace); the exchanges are device copies between the shards' buffers.  For tests and for measuring what the schedule costs
    besides the transport on a single GPU (tools/sp_overhead.py)."""
    P = len(dits)
    shards = [SequenceShard(d, r, P, groups, mode) for r, d in enumerate(dits)]
    for sh in shards:
        sh.begin(x, timestep, context, **cond)
    G = shards[0].G
    if shards[0].mode == "gather":
        for layer in range(dits[0].num_layers):
            for sh in shards:
                sh.block_v_rows(layer)                               # the split form forward_distributed uses (V^T leaves first)
                sh.block_qk_rows(layer)
            for sj in shards:
                for i, si in enumerate(shards):
                    sj.buf.k_all[i].copy_(si.buf.k)
                    sj.buf.vt_all[i].copy_(si.buf.vt)
            for sh in shards:
                sh.attention_rows()
                sh.block_rest_rows(layer)
        rows = torch.cat([sh.head() for sh in shards], dim=0)
        return shards[0].unpatchify(rows)
    for layer in range(dits[0].num_layers):
        for sh in shards:
            sh.block_qkv(layer)
        for j, sj in enumerate(shards):                 # rank j receives piece j of every rank i
            for i, si in enumerate(shards):
                sj.buf.vt_recv[i].copy_(si.buf.vt_send[j])
                sj.buf.qk_recv[:, :, i].copy_(si.buf.qk_send[:, :, j])
        for sh in shards:
            sh.unpack_v()
            for g in range(G):
                sh.attention(g)
        for j, sj in enumerate(shards):
            for i, si in enumerate(shards):
                sj.buf.o_recv[:, i].copy_(si.buf.o_send[:, j])
        for sh in shards:
            sh.block_rest(layer)
    rows = torch.cat([sh.head() for sh in shards], dim=0)
    return shards[0].unpatchify(rows)


def _pair_outputs(sh: SequenceShard, rows_all: torch.Tensor):
    """rows_all [P, 2 Ls, ld] (every rank's stacked head rows, rank-major) -> (cond, uncond) latents."""
    P, Ls = rows_all.shape[0], sh.Ls
    return tuple(sh.unpatchify(rows_all[:, br * Ls:(br + 1) * Ls].reshape(P * Ls, -1)) for br in range(2))


def _blocks_distributed(sh: SequenceShard, group) -> None:
    """The block loop of the Ulysses mode for this rank (one branch or a stacked pair: the same exchanges, pieces nb x as long)."""
    b, G = sh.buf, sh.G
    for layer in range(sh.dit.num_layers):
        sh.block_qkv(layer)
        wv = _exchange(b.vt_recv, b.vt_send, group, True)
        wq = [(_exchange(b.qk_recv[0, g], b.qk_send[0, g], group, True), _exchange(b.qk_recv[1, g], b.qk_send[1, g], group, True)) for g in range(G)]
        if wv is not None:
            wv.wait()
        sh.unpack_v()
        wo = []
        for g in range(G):
            for w_ in wq[g]:
                if w_ is not None:
                    w_.wait()
            sh.attention(g)
            wo.append(_exchange(b.o_recv[g], b.o_send[g], group, True))
        for w_ in wo:
            if w_ is not None:
                w_.wait()
        sh.block_rest(layer)


def forward_distributed_pair(dit: WanDiT, x, timestep, context_cond, context_uncond, group=None, groups: Optional[int] = None, **cond):
    """Both forwards of a CFG step (svi_video.py:401-408) for this rank of `group`, STACKED on the rank's rows: every row-local launch runs once over
    2 L / P rows (the conditional branch's on top), the two branches are twice as many heads of each attention launch and ride in the same exchange
    pieces, and no rank waits for another's noise prediction — each rank ends with both.  Returns (noise_pred_cond, noise_pred_uncond), full latents on
    every rank; bit-identical to two forward_distributed calls with the key axis in one piece and to WanDiT.forward_cfg_pair on one rank (at P = 4 of the
    1.3B model the doubled head count makes the attention's rounds whole, so no key-axis cut is taken at all).  Ulysses mode; context cache on."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sh = SequenceShard(dit, rank, world, groups, "ulysses")
    sh.begin_pair(x, timestep, context_cond, context_uncond, **cond)
    _blocks_distributed(sh, group)
    rows = sh.head()
    gathered = all_gather_rows(rows, group).reshape(world, 2 * sh.Ls, -1)
    return _pair_outputs(sh, gathered)


def forward_local_pair(dits: Sequence[WanDiT], x, timestep, context_cond, context_uncond, groups: Optional[int] = None, **cond):
    """forward_distributed_pair's schedule with P = len(dits) shards in ONE process (device copies for the exchanges): tests, tools/sp_overhead.py."""
    P = len(dits)
    shards = [SequenceShard(d, r, P, groups, "ulysses") for r, d in enumerate(dits)]
    for sh in shards:
        sh.begin_pair(x, timestep, context_cond, context_uncond, **cond)
    G = shards[0].G
    for layer in range(dits[0].num_layers):
        for sh in shards:
            sh.block_qkv(layer)
        for j, sj in enumerate(shards):
            for i, si in enumerate(shards):
                sj.buf.vt_recv[i].copy_(si.buf.vt_send[j])
                sj.buf.qk_recv[:, :, i].copy_(si.buf.qk_send[:, :, j])
        for sh in shards:
            sh.unpack_v()
            for g in range(G):
                sh.attention(g)
        for j, sj in enumerate(shards):
            for i, si in enumerate(shards):
                sj.buf.o_recv[:, i].copy_(si.buf.o_send[:, j])
        for sh in shards:
            sh.block_rest(layer)
    rows = torch.stack([sh.head() for sh in shards], dim=0)
    return _pair_outputs(shards[0], rows)

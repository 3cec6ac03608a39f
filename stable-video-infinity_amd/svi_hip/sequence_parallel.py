"""Sequence-parallel (Ulysses) DiT forward: one clip's forward spread over P ranks — SURVEY §8e axis 3.

The reference's own design for this is USP (pipelines/svi_video.py:119-135: chunk the token axis, all_gather after the head;
distributed/xdit_context_parallel.py: all-to-all around attention).  Here:

  * rank r owns token rows [r*Ls, (r+1)*Ls) of the (f h w) sequence, Ls = L / P, for everything row-local: LN / modulation,
    q k v projections, RMSNorm + RoPE (at the rows' true grid positions), output projection, cross-attention, MLP, head;
  * around self-attention tokens are traded for heads: an all-to-all turns [Ls tokens, H heads] into [L tokens, H/P heads],
    attention runs on whole sequences of a head group, a second all-to-all turns the result back;
  * after the head an all-gather of [Ls, 64] rows rebuilds the latent.

Per block and rank the exchanges move 3*Ls*D/P*(P-1) + Ls*D/P*(P-1) bf16 values: at C2 on 4 ranks 2 x 75 MB out and back per
block over xGMI, ~1 ms at 150 GB/s per link, against ~4 ms of compute per block and rank.  Constraints: L % P == 0 and
heads % P == 0 (1.3B: 12 heads -> P in {2,3,4,6}; 14B: 40 heads -> {2,4,5,8}).  Composes with CfgPair: 8 GPUs = 2 x 4.

Every arithmetic kernel is the one the single-GPU forward uses, on the same operands per row / per head, so the result is
bit-identical to `WanDiT.forward` for any P (tests/test_gpu_sp.py runs P = 1, 2, 4 shards on one GPU with the exchange
simulated in-process; tests/test_dist_gloo.py runs the pack / all-to-all / unpack layout code over gloo).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib as L
from .dit import WanDiT


# ---- layout code of the two exchanges (pure tensor reshuffles, device-agnostic) ------------------------------------------------
def pack_qkv(qk: torch.Tensor, vt: torch.Tensor, P: int) -> torch.Tensor:
    """qk [Ls, 2D] (q | k), vt [D, >=Ls] -> send [P, 3, Ls*D/P]: slot j = this rank's rows of head group j's q, k and V^T."""
    Ls, D2 = qk.shape
    D = D2 // 2
    Dp = D // P
    send = torch.empty((P, 3, Ls * Dp), dtype=qk.dtype, device=qk.device)
    send[:, 0] = qk[:, :D].reshape(Ls, P, Dp).permute(1, 0, 2).reshape(P, Ls * Dp)
    send[:, 1] = qk[:, D:].reshape(Ls, P, Dp).permute(1, 0, 2).reshape(P, Ls * Dp)
    send[:, 2] = vt[:, :Ls].reshape(P, Dp, Ls).reshape(P, Dp * Ls)
    return send


def unpack_qkv(recv: torch.Tensor, Ls: int, Dp: int):
    """recv [P(source rank), 3, Ls*Dp] -> Q, K [L, Dp] (source-major = token order), V^T [Dp, L8] (zero beyond L)."""
    P = recv.shape[0]
    Lfull = P * Ls
    L8 = (Lfull + 7) // 8 * 8
    q = recv[:, 0].reshape(Lfull, Dp)
    k = recv[:, 1].reshape(Lfull, Dp)
    vt = torch.zeros((Dp, L8), dtype=recv.dtype, device=recv.device)
    vt[:, :Lfull] = recv[:, 2].reshape(P, Dp, Ls).permute(1, 0, 2).reshape(Dp, Lfull)
    return q.contiguous(), k.contiguous(), vt


def pack_out(o: torch.Tensor, P: int) -> torch.Tensor:
    """o [L, Dp] (all tokens, this rank's head group) -> send [P, Ls*Dp]: slot r = the rows rank r owns."""
    Lfull, Dp = o.shape
    return o.reshape(P, (Lfull // P) * Dp)


def unpack_out(recv: torch.Tensor, Ls: int, Dp: int) -> torch.Tensor:
    """recv [P(head group), Ls*Dp] -> attn [Ls, D] with head group g at columns [g*Dp, (g+1)*Dp)."""
    P = recv.shape[0]
    return recv.reshape(P, Ls, Dp).permute(1, 0, 2).reshape(Ls, P * Dp).contiguous()


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
    """The same exchange among P shards living in one process (single-GPU test of the whole scheme)."""
    P = len(sends)
    return [torch.stack([sends[i][j] for i in range(P)]) for j in range(P)]


# ---- one rank's share --------------------------------------------------------------------------------------------------------
class SequenceShard:
    def __init__(self, dit: WanDiT, rank: int, world: int):
        if dit.num_heads % world:
            raise ValueError(f"{dit.num_heads} heads do not divide over {world} ranks")
        self.dit, self.rank, self.world = dit, rank, world
        self.Dp = dit.dim // world
        self.heads_local = dit.num_heads // world

    def begin(self, x, timestep, context, clip_feature=None, y=None, add_condition=None):
        d = self.dit
        d.check_inputs(x, (context,), clip_feature, y, add_condition)
        if not d.has_image_input:
            clip_feature = None
        x = x.to(torch.bfloat16).contiguous()
        B, _, T, H, W = x.shape
        if B != 1:
            raise ValueError("sequence-parallel forward takes one sample (batch over samples with clip sharding instead)")
        self.T, self.H, self.W = T, H, W
        self.L = d.tokens(T, H, W)
        if self.L % self.world:
            raise ValueError(f"{self.L} tokens do not divide over {self.world} ranks")
        self.Ls = self.L // self.world
        self.ldvt = (self.Ls + 7) // 8 * 8
        context, clip_feature = d._prompt_args(context, clip_feature)
        self._keep = [x, context, timestep.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous(), clip_feature,
                      None if y is None else y.to(torch.bfloat16).contiguous(),
                      None if add_condition is None else add_condition.to(torch.bfloat16).contiguous()]
        x, context, ts, clip, yy, addc = self._keep
        L.check(L.lib().svi_dit_sp_begin(d._h, L.ptr(x), L.ptr(ts), L.ptr(context), L.ptr(clip), L.ptr(yy), L.ptr(addc), T, H, W,
                                         context.shape[1], self.rank * self.Ls, self.Ls, L.current_stream()), "svi_dit_sp_begin")
        dev = x.device
        self.qk = torch.empty((self.Ls, 2 * d.dim), dtype=torch.bfloat16, device=dev)
        self.vt = torch.zeros((d.dim, self.ldvt), dtype=torch.bfloat16, device=dev)
        self.o = torch.empty((self.L, self.Dp), dtype=torch.bfloat16, device=dev)

    def block_qkv(self, layer: int) -> torch.Tensor:
        L.check(L.lib().svi_dit_sp_block_qkv(self.dit._h, layer, L.ptr(self.qk), L.ptr(self.vt), self.ldvt, L.current_stream()), "svi_dit_sp_block_qkv")
        return pack_qkv(self.qk, self.vt, self.world)

    def attention(self, recv: torch.Tensor) -> torch.Tensor:
        q, k, vt = unpack_qkv(recv, self.Ls, self.Dp)
        L.check(L.lib().svi_attention_vt_fwd(L.ptr(q), self.Dp, L.ptr(k), self.Dp, L.ptr(vt), vt.shape[1], L.ptr(self.o), self.Dp,
                                             self.L, self.L, self.heads_local, 1, L.current_stream()), "svi_attention_vt_fwd")
        self._alive = (q, k, vt)                      # keep the operands until the stream has consumed them
        return pack_out(self.o, self.world)

    def block_rest(self, layer: int, recv: torch.Tensor) -> None:
        attn = unpack_out(recv, self.Ls, self.Dp)
        L.check(L.lib().svi_dit_sp_block_rest(self.dit._h, layer, L.ptr(attn), L.current_stream()), "svi_dit_sp_block_rest")
        self._alive2 = attn

    def head(self) -> torch.Tensor:
        ld = L.lib().svi_dit_head_ld(self.dit._h)
        rows = torch.empty((self.Ls, ld), dtype=torch.bfloat16, device=self.qk.device)
        L.check(L.lib().svi_dit_sp_head(self.dit._h, L.ptr(rows), L.current_stream()), "svi_dit_sp_head")
        return rows

    def unpatchify(self, head_rows: torch.Tensor) -> torch.Tensor:
        out = torch.empty((1, self.dit.out_dim, self.T, self.H, self.W), dtype=torch.bfloat16, device=head_rows.device)
        head_rows = head_rows.contiguous()
        L.check(L.lib().svi_dit_unpatchify(self.dit._h, L.ptr(head_rows), L.ptr(out), self.T, self.H, self.W, L.current_stream()), "svi_dit_unpatchify")
        return out


# ---- drivers -----------------------------------------------------------------------------------------------------------------
def forward_distributed(dit: WanDiT, x, timestep, context, group=None, **cond) -> torch.Tensor:
    """model_fn_wan_video(..., use_unified_sequence_parallel=True) for this rank of `group`: every rank passes the same inputs
    and receives the full output.  Two all-to-alls per block, one all-gather per forward."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sh = SequenceShard(dit, rank, world)
    sh.begin(x, timestep, context, **cond)
    for layer in range(dit.num_layers):
        o_send = sh.attention(all_to_all(sh.block_qkv(layer), group))
        sh.block_rest(layer, all_to_all(o_send, group))
    return sh.unpatchify(all_gather_rows(sh.head(), group))


def forward_local(dits: Sequence[WanDiT], x, timestep, context, **cond) -> torch.Tensor:
    """The same schedule with P = len(dits) shards in ONE process (each shard needs its own handle: a handle holds one
    workspace); the exchanges are in-process gathers.  For tests and for checking a sharding on a single GPU."""
    P = len(dits)
    shards = [SequenceShard(d, r, P) for r, d in enumerate(dits)]
    for sh in shards:
        sh.begin(x, timestep, context, **cond)
    for layer in range(dits[0].num_layers):
        recv = all_to_all_local([sh.block_qkv(layer) for sh in shards])
        back = all_to_all_local([sh.attention(r) for sh, r in zip(shards, recv)])
        for sh, r in zip(shards, back):
            sh.block_rest(layer, r)
    rows = torch.cat([sh.head() for sh in shards], dim=0)
    return shards[0].unpatchify(rows)

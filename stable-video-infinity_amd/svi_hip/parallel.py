"""Multi-GPU execution of the rolling window: one process per GPU, clips sharded over ranks, RCCL over xGMI.

What shards: for the T2V model the clip-to-clip hand-off never reaches the network
(pipelines/svi_video.py:481,493-494 ignore `input_image` without an image encoder), so clip k depends only on
(prompt_k, seed_k = k * seed_times) (test_svi.py:424-476).  Clips are therefore independent units:

    clip k  ->  rank k mod P      (weights replicated: 2.8 GB bf16 on 288 GB)

There is NO collective inside the 50-step denoise loop.  The only exchange is at the end of a round of clips:
an all-gather of each clip's final latents ([16,21,h,w] bf16 = 4.2 MB) — which contains the tail (motion) latent
frames every rank needs to know where the next clip of the window starts — so that any rank can decode and
stitch the window in order with the reference's stitching rule (drop the last `num_motion_frames` frames of every
clip but the last, test_svi.py:472-476).  Over xGMI a 4.2 MB all-gather on 8 GPUs is ~30 us of wire time; it is a
per-clip (every ~40 s) event.

For I2V (image-conditioned) streams clip k+1 needs clip k's decoded frames, which is sequential by definition;
those shard over *samples* with the same code (`units` are then whole samples).

Second axis, for the latency of ONE clip: the cond and uncond forwards of a step are independent
(pipelines/svi_video.py:401-408), so a pair of ranks runs one each and exchanges `noise_pred` ([1,16,21,h,w] bf16 =
4.2 MB per step, one 2-rank all-gather over a single xGMI link, ~30 us against a ~250 ms forward); both ranks then apply
CFG + Euler redundantly, so both hold bit-identical latents and the result equals the serial order (`CfgPair`).  The two
axes compose: world = pairs x 2, clips round-robin over pairs (`CfgPair.split_world`).

Backends: "nccl" (= RCCL on ROCm) on GPUs; "gloo" on CPU for the world_size-2 tests of the sharding/stitching logic.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_units(num_units: int, rank: int, world: int) -> List[int]:
    """Round-robin: unit k -> rank k mod world (keeps every rank's load within one clip of the others)."""
    return [k for k in range(num_units) if k % world == rank]


def clip_seed(chunk_idx: int, seed_times: int = 42) -> Optional[int]:
    """seed = chunk_idx * seed_times; seed_times == -1 means unseeded (test_svi.py:425-428)."""
    return None if seed_times == -1 else int(chunk_idx * seed_times)


def clip_prompt_index(chunk_idx: int, num_prompts: int, prompt_repeat_times: int = 1, use_first_prompt_only: bool = False) -> int:
    """Prompt cycling of the clip loop (test_svi.py:430-438)."""
    if use_first_prompt_only:
        return 0
    return (chunk_idx // prompt_repeat_times) % num_prompts


class ClipParallel:
    """Process-group wrapper: who am I, which clips are mine, and the end-of-round all-gather."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1

    def my_clips(self, num_clips: int) -> List[int]:
        return shard_units(num_clips, self.rank, self.world)

    def all_gather_clips(self, local: Dict[int, torch.Tensor], num_clips: int, like: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """Every rank contributes the latents of its clips; every rank receives all `num_clips`, in clip order.

        One all-gather per round r (clips r*P .. r*P+P-1): a rank with no clip in the last, partial round
        contributes a zero tensor that is dropped on receipt (all ranks must enter the collective)."""
        if not self.enabled or self.world == 1:
            return [local[k] for k in range(num_clips)]
        proto = like if like is not None else next(iter(local.values()))
        out: List[Optional[torch.Tensor]] = [None] * num_clips
        rounds = (num_clips + self.world - 1) // self.world
        for r in range(rounds):
            k = r * self.world + self.rank
            mine = local[k].contiguous() if k < num_clips else torch.zeros_like(proto)
            bucket = [torch.empty_like(proto) for _ in range(self.world)]
            dist.all_gather(bucket, mine, group=self.group)
            for src in range(self.world):
                kk = r * self.world + src
                if kk < num_clips:
                    out[kk] = bucket[src]
        return out  # type: ignore[return-value]

    def all_gather_motion_tails(self, local: Dict[int, torch.Tensor], num_clips: int, num_motion_latents: int = 1) -> List[torch.Tensor]:
        """Only the last `num_motion_latents` latent frames of each clip ([16, n, h, w]); what the next clip's
        conditioning would be built from.  Same collective shape as all_gather_clips, 1/21 of the bytes."""
        tails = {k: v[..., -num_motion_latents:, :, :].contiguous() for k, v in local.items()}
        return self.all_gather_clips(tails, num_clips)


class CfgPair:
    """Two ranks share one clip: role 0 computes the conditional forward, role 1 the unconditional one.

    `step()` has the contract of DenoiseLoop.step (svi_hip/pipeline.py) and takes the two device functions it needs as
    arguments, so the exchange logic is the same object on GPUs (RCCL) and in the CPU gloo tests."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("CfgPair needs an initialised process group")
        self.group = group
        if dist.get_world_size(group) != 2:
            raise RuntimeError(f"a CFG pair is exactly 2 ranks, got {dist.get_world_size(group)}")
        self.role = dist.get_rank(group)

    @staticmethod
    def split_world() -> "tuple[CfgPair, int, int]":
        """world = P pairs x 2 ranks: ranks (2p, 2p+1) form pair p.  Returns (pair, pair_index, num_pairs); clips are
        then sharded over pairs with shard_units(num_clips, pair_index, num_pairs).  Collective: every rank must call."""
        world, rank = dist.get_world_size(), dist.get_rank()
        if world % 2:
            raise RuntimeError(f"CFG pairing needs an even world size, got {world}")
        mine = None
        for p in range(world // 2):                       # new_group is collective over the whole world
            g = dist.new_group(ranks=[2 * p, 2 * p + 1])
            if rank // 2 == p:
                mine = g
        return CfgPair(mine), rank // 2, world // 2

    def step(self, forward: Callable[..., torch.Tensor], cfg_step: Callable[..., None], latents: torch.Tensor,
             timestep: torch.Tensor, dsigma: float, ctx_pos: torch.Tensor, ctx_neg: torch.Tensor, cfg_scale: float,
             uncond_overrides: Optional[dict] = None, **cond) -> torch.Tensor:
        """One scheduler step in place on `latents`.  forward(latents, timestep, context, **cond) -> noise_pred;
        cfg_step(latents, cond_pred, uncond_pred, cfg_scale, dsigma) applies u + s(c - u) and the Euler update.
        `uncond_overrides`: keyword inputs that differ for the unconditional branch (the dance sampler's add_condition=None)."""
        if self.role == 1 and uncond_overrides:
            cond = dict(cond, **uncond_overrides)
        mine = forward(latents, timestep, ctx_pos if self.role == 0 else ctx_neg, **cond).contiguous()
        both = [torch.empty_like(mine), torch.empty_like(mine)]
        dist.all_gather(both, mine, group=self.group)
        cfg_step(latents, both[0], both[1], cfg_scale, dsigma)
        return latents


def split_cfg_sequence(world: Optional[int] = None, rank: Optional[int] = None):
    """One clip on an even number of ranks: world = 2 (CFG branches) x S (sequence shards).  Rank r = branch * S + shard:
    ranks [0, S) carry the conditional forward, [S, 2S) the unconditional one, each spread Ulysses-style over its S ranks;
    ranks (s, S + s) form a CFG pair and exchange noise_pred once per step.  Returns (CfgPair, sp_group, S).  Collective."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    if world % 2:
        raise RuntimeError(f"CFG x sequence needs an even world size, got {world}")
    S = world // 2
    sp_mine = pair_mine = None
    for b in range(2):
        g = dist.new_group(ranks=list(range(b * S, (b + 1) * S)))
        if rank // S == b:
            sp_mine = g
    for s_ in range(S):
        g = dist.new_group(ranks=[s_, S + s_])
        if rank % S == s_:
            pair_mine = g
    return CfgPair(pair_mine), sp_mine, S


def stitch_window(clips: Sequence[Sequence], num_motion_frames: int) -> list:
    """The reference's stitching rule (test_svi.py:472-476): every clip but the last loses its final
    `num_motion_frames` frames (they are re-generated as the head of the next clip)."""
    frames: list = []
    n = len(clips)
    for i, c in enumerate(clips):
        c = list(c)
        frames += c[:-num_motion_frames] if (i < n - 1 and num_motion_frames > 0) else c
    return frames


def run_window(denoise_clip: Callable[[int], torch.Tensor], num_clips: int, par: Optional[ClipParallel] = None,
               ) -> List[torch.Tensor]:
    """Denoise `num_clips` independent clips across the ranks of `par`; returns all final latents, in clip order, on
    every rank.  `denoise_clip(k)` must depend only on k (seed/prompt derive from k) — that is what makes the result
    identical for any number of ranks."""
    par = par or ClipParallel()
    local = {k: denoise_clip(k) for k in par.my_clips(num_clips)}
    proto = None
    if not local:                                  # more ranks than clips: still join the collective
        proto = denoise_clip.__dict__.get("proto")
        if proto is None:
            raise RuntimeError("rank without clips needs denoise_clip.proto (a tensor shaped like a clip's latents)")
    return par.all_gather_clips(local, num_clips, like=proto)

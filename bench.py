"""bench.py — denoised latent frames / second of the SVI rolling-window hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks (it re-executes itself under
torch.distributed.run on 127.0.0.1, one rank per GPU, RCCL); it refuses — one line on stderr, non-zero exit — when fewer than N GPUs
are visible or fewer than N ranks join, and never falls back to one rank.  `n_gpus` on the JSON line is the number of ranks that
completed an RCCL all-reduce, `config.ranks` lists each rank's device and PCI bus id, `config.rccl` the library version.

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): Wan2.1-T2V-1.3B architecture, random-init bf16 weights,
one 81-frame 832x480 clip = latents [1,16,21,60,104] -> 32760 tokens, text context [1,512,4096] (pos/neg), CFG 5.0,
flow-match shift 5, 50 scheduler steps per clip.  A bench "step" is ONE scheduler step of the clip's denoise loop:
cond forward + uncond forward through all 30 DiT blocks + CFG combine + Euler update, latents resident in HBM.
With N ranks every rank denoises its own clip (weak scaling, SURVEY §8e axis 1: T2V clips are independent); the
only data-path exchange is the all-gather of each clip's tail (motion) latents over RCCL at the end of the region.

    value = N * 21 latent frames / (50 * step_time + vae_decode_time)        [latent frames / s]

The VAE decode of the finished clip is timed in the same run (outside the K-step region, on the same stream) and
its time is part of `value`; `config.dit_only_value` reports the a1-only variant.

The single-rank step replays ONE hipGraph per step (the installed sampler's default, DenoiseLoop(graph=True): the ~1500 launches of a
step as one; bit-identical); `--no-graph` launches eagerly.  `--transport gloo` runs the multi-rank path with the exchanges through
the host and the ranks sharing the visible device(s): a probe of that code path on a one-GPU box, never a scaling figure.

Round 6: `--attn-gain G` / `--prompt-tokens N` (the non-benign lines: peaky attention logits through learned-gain factors; a long prompt on the streaming
cross-attention kernel), `--budget-s S` (wall-clock budget counted from process start, default 600: optional parts are dropped, the K timed steps never; the plan is
printed after the warm-up), `--rccl-probe` (N = 1 with a live one-rank RCCL communicator), `--fp8-all` (opt-in: every token-side GEMM and QK^T on MX fp8), and beside an
RCCL communicator the step graph is tried behind a guard and voted on (`config.hip_graph`, `config.hip_graph_note`).

Extra objects on the JSON line:
  roofline      dominant kernel = self-attention flash kernel: algorithmic 4*L^2*D FLOP per launch divided by
                its mean launch time measured with HIP events on the launch stream — inside the timed region with --no-graph; with
                the hipGraph (default) over eager steps of the same loop run directly behind it (a replay cannot carry event pairs).
                `traffic` / `mfma_busy_in_clock` come from the rocprofv3 --pmc summaries under profiles/ and are given only when
                EVERY kernel source hash recorded with them equals the tree being timed.
  vendor_yardstick / roofline_all.<family>.vendor   (N = 1, c2) tools/yardstick.py as a child process after the timed parts: hipBLASLt (torch.matmul) and
                F.scaled_dot_product_attention on the step's six hot shapes beside this repo's launches, same box, same job.  Measurement only.
  cpu_baseline  the reference's own DiTBlock.forward and WanVideoVAE.decode (oracle/_ref, built by oracle/build_ref.py: kind
                "reference") — or, where that copy is absent, the oracle's restatement (kind "port") — timed on this box's
                host cores on a bounded sample (one of the 30 blocks of one of the 100 forwards, full size; 2 of the 21 latent
                frames), extrapolated to a clip.  Rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "stable-video-infinity_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

T_PROCESS_START = time.perf_counter()      # --budget-s and config.wall_s count from here (the interpreter's own start-up is ahead of it)
# the pool's host driver only supports dmabuf IPC: RCCL between processes needs this before the HSA runtime starts (a launcher that did not export it, e.g. a bare torchrun)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

WORKLOADS = {
    # name: (latent T, H, W), context tokens, steps per clip
    "c2": dict(lat=(21, 60, 104), lc=512, steps_per_clip=50, desc="Wan2.1-T2V-1.3B 81f@832x480 50-step CFG5 single clip per GPU"),
    "c4": dict(lat=(21, 60, 104), lc=512, steps_per_clip=50, model="WAN_14B_I2V",
               desc="Wan2.1-I2V-14B 81f@832x480 50-step CFG5 single clip per GPU (SVI's own base model; DiT only, y/clip_feature synthetic)"),
    "c1": dict(lat=(5, 32, 32), lc=512, steps_per_clip=10, desc="Wan2.1-T2V-1.3B 17f@256x256 10-step CFG5 (reference CPU-runnable case)"),
    # BASELINE configs[4] names "Wan2.2-5B fp8 MFMA path, skeleton-conditioned (test_svi_dance.py)".  The reference contains neither a
    # Wan2.2-5B model nor fp8 arithmetic; what test_svi_dance.py runs is SVIDanceVideoPipeline = Wan2.1-I2V-14B + the pose
    # embedder, optionally with FP8 weight STORAGE.  That workload, as the reference has it:
    "c5": dict(lat=(21, 60, 104), lc=512, steps_per_clip=50, model="WAN_14B_I2V", pose=True, fp8_storage=True,
               desc="skeleton-conditioned clip as test_svi_dance.py runs it: Wan2.1-I2V-14B + dwpose embedder (add_condition on the conditional branch), "
                    "FP8 weight storage (exact cast at bind, bf16 arithmetic), 81f@832x480 50-step CFG5; no Wan2.2-5B / fp8-MFMA exists in the reference"),
}
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def L_TOKENS(T: int, H: int, W: int) -> int:
    return T * (H // 2) * (W // 2)


def vendor_yardstick(timeout_s: int = 240):
    """tools/yardstick.py in a child process on this box, this job: torch.matmul (hipBLASLt) and F.scaled_dot_product_attention on the step's six hot
    shapes beside this repo's launches of the same shapes.  Measurement only — the product neither links nor calls a vendor kernel."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"svi_yardstick_{os.getpid()}.json")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "yardstick.py"), "3", "--json", out], capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.load(open(out))
    except Exception as ex:  # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}
    finally:
        try:
            os.remove(out)
        except OSError:
            pass


def device_weights(cfg: dict, seed: int, device) -> dict:
    """Random-init weights of the named architecture, generated on the GPU (no checkpoints, no network)."""
    import synth
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in synth.dit_param_shapes(**cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "modulation":
            t = torch.randn(shape, generator=g, device=device) / math.sqrt(shape[-1])
        elif leaf == "weight" and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif leaf == "bias" and "norm" in name:
            t = 0.1 * torch.randn(shape, generator=g, device=device)
        elif leaf == "weight":
            bound = 1.0 / math.sqrt(float(torch.tensor(shape[1:]).prod()))
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) * bound
        else:
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) * 0.05
        out[name] = t.to(torch.bfloat16).contiguous()
    return out


def cpu_baseline_worker() -> None:
    """Child process: time ONE DiTBlock forward at the full C2 size (L=32760, fp32) and the VAE decode of two latent frames (5 video
    frames) at the C2 spatial size on the host cores; print seconds.  With oracle/_ref present (oracle/build_ref.py: the reference's
    own modules, copied at build time) it is the REFERENCE's DiTBlock.forward (wan_video_dit.py:354-374) and WanVideoVAE.decode
    (wan_video_vae.py:777-789) that are timed — kind "reference"; otherwise the oracle's restatement — kind "port"."""
    import synth
    from oracle import build_ref
    c = dict(synth.WAN_1_3B)
    c["num_layers"] = 1
    grid = (21, 30, 52)
    f, h, w = grid
    L = f * h * w
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(0, **c).items()}
    x = torch.from_numpy(synth.randn(1, 1, L, 1536))
    ctx = torch.from_numpy(synth.randn(2, 1, 512, 1536))
    tm = torch.from_numpy(0.1 * synth.randn(3, 1, 6, 1536))
    vsd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    z = torch.from_numpy(synth.randn(511, 1, 16, 2, 60, 104))
    kind = "port"
    if build_ref.available():
        try:
            from oracle import ref_shim
            dit_mod, vae_mod, _ = ref_shim.load(build_ref.OUT)
            blk = dit_mod.DiTBlock(False, c["dim"], 12, c["ffn_dim"], 1e-6).eval()
            blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")}, strict=True)
            fr = dit_mod.precompute_freqs_cis_3d(128)
            freqs = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                               fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
            v = vae_mod.WanVideoVAE()
            v.load_state_dict(vsd, strict=True)
            kind = "reference"
        except Exception as ex:      # a broken copy must not pass for the reference: fall back to the port and say so
            print(f"oracle/_ref present but not importable ({type(ex).__name__}: {ex}); timing the port", file=sys.stderr, flush=True)
            kind = "port"
    if kind == "reference":
        with torch.no_grad():
            t0 = time.time()
            blk(x, ctx, tm, freqs)
            block_s = time.time() - t0
            t0 = time.time()
            v.decode([z[0]], device="cpu")
            vae_s = time.time() - t0
    else:
        from oracle import wan_dit_oracle as wdo
        from oracle import wan_vae_oracle as wvo
        cfg = wdo.DiTConfig(num_layers=1)
        rope = wdo.rope_table_3d(128, grid)
        with torch.no_grad():
            t0 = time.time()
            wdo.dit_block(sd, "blocks.0.", x, ctx, tm, rope, cfg)
            block_s = time.time() - t0
            t0 = time.time()
            wvo.vae_decode(vsd, z)
            vae_s = time.time() - t0
    print(json.dumps({"block_seconds": block_s, "vae_2_latent_frames_seconds": vae_s, "threads": torch.get_num_threads(), "kind": kind}), flush=True)


def cpu_baseline(max_threads: int = 32, timeout_s: int = 300) -> dict:
    """The reference's CPU path on this box's host cores, in a child process pinned to `threads` OpenMP threads, on a bounded sample of
    the workload: one of the 30 blocks of one of the 100 forwards of a clip at full size, and the VAE decode of 2 of the clip's 21 latent
    frames (5 of its 81 frames); extrapolated to a clip.  kind "reference": the reference's own DiTBlock / WanVideoVAE from oracle/_ref
    (oracle/build_ref.py); kind "port": oracle/_ref is absent and the oracle's restatement was timed instead."""
    import subprocess
    threads = max(1, min(max_threads, os.cpu_count() or 1))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    base = {"unit": "latent frames/s", "cores": threads, "kind": "port"}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], env=env, capture_output=True,
                           text=True, timeout=timeout_s)
        res = json.loads(r.stdout.strip().splitlines()[-1])
        dt, vt = res["block_seconds"], res["vae_2_latent_frames_seconds"]
        base["kind"] = res.get("kind", "port")
    except Exception as ex:  # timeout or failure: say so instead of inventing a number
        return dict(base, value=None, sample=f"DiTBlock at L=32760 + VAE sample did not finish within {timeout_s}s on {threads} threads ({type(ex).__name__})")
    clip_s = dt * 30 * 100 + vt * 81.0 / 5.0
    what = ("the reference's own DiTBlock.forward (diffsynth/models/wan_video_dit.py:354-374, fp32, F.scaled_dot_product_attention) and WanVideoVAE.decode "
            "(wan_video_vae.py:777-789) from oracle/_ref") if base["kind"] == "reference" else \
           "the oracle's restatement (oracle/wan_dit_oracle.py, oracle/wan_vae_oracle.py; oracle/_ref absent)"
    return dict(base, value=21.0 / clip_s,
                sample=f"1 DiTBlock forward at L=32760 took {dt:.1f}s and the VAE decode of 2 latent frames (5 frames 480x832) {vt:.1f}s on {threads} threads — {what}; "
                       f"extrapolated x30 blocks x100 forwards + decode x81/5 per clip")


def pmc_summaries() -> dict:
    """The newest rocprofv3 --pmc summaries under profiles/ (tools/profile_round.sh) that were collected on EXACTLY the kernel sources being
    timed: every hash of `<tag>_source_hashes.json` must equal the sha256[:16] of the csrc file of that name in this tree.  A summary collected
    on any other source state is not quoted: {"why": ...} instead."""
    import glob
    import hashlib
    try:
        csrc = os.path.join(ROOT, "stable-video-infinity_amd", "csrc")
        tags = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_source_hashes.json")), key=os.path.getmtime, reverse=True)
        for f in tags:
            rec = json.load(open(f))
            if not rec or any(not os.path.exists(os.path.join(csrc, n)) or hashlib.sha256(open(os.path.join(csrc, n), "rb").read()).hexdigest()[:16] != h
                              for n, h in rec.items()):
                continue
            tag = os.path.basename(f)[:-len("_source_hashes.json")]
            out = {"tag": tag}
            for key in ("flash", "flash_qk8", "gemm_ffn1", "gemm_ffn2", "flash_cross", "vae_conv"):
                p = os.path.join(ROOT, "profiles", f"{tag}_{key}_pmc.json")
                if os.path.exists(p):
                    out[key] = json.load(open(p))
                    out[key + "_file"] = os.path.relpath(p, ROOT)
            return out
        return {"why": "no profiles/*_source_hashes.json matches the csrc/*.hip being timed (tools/profile_round.sh was not run on this tree)"}
    except Exception as ex:
        return {"why": f"profiles/ not readable ({type(ex).__name__})"}


class Heartbeat:
    """Per-rank progress marks for a multi-GPU run nobody watches (the driver's 8-GPU scaling run is the first time RCCL carries this code): every rank
    prints one stderr line per stage — before the process group exists, after the ranks joined, after warm-up, after the timed region — and records the
    stage in a small file the other ranks can read.  A watchdog thread turns a hang (a collective some rank never enters) into a non-zero exit that NAMES
    the rank(s) furthest behind, instead of a silent timeout of the whole job (`--rank-timeout`)."""
    STAGES = ["start", "process-group", "joined", "model-bound", "warmup-done", "timed-done", "clips-done", "line-printed"]

    def __init__(self, rank: int, world: int, timeout_s: float):
        import tempfile
        import threading
        self.rank, self.world, self.timeout_s = rank, world, float(timeout_s)
        self.t0 = self.last = time.monotonic()
        self.stage = "start"
        tag = os.environ.get("MASTER_PORT", "0") + "_" + os.environ.get("TORCHELASTIC_RUN_ID", "none")
        self.dir = os.path.join(tempfile.gettempdir(), f"svi_bench_hb_{tag}")
        os.makedirs(self.dir, exist_ok=True)
        self._write()
        self._stop = threading.Event()
        if world > 1 and self.timeout_s > 0:
            threading.Thread(target=self._watch, name="svi-bench-watchdog", daemon=True).start()

    def _write(self) -> None:
        try:
            with open(os.path.join(self.dir, f"rank_{self.rank}"), "w") as f:
                f.write(f"{self.STAGES.index(self.stage)} {self.stage} {os.getpid()}\n")
        except OSError:
            pass

    def mark(self, stage: str, note: str = "") -> None:
        self.stage, self.last = stage, time.monotonic()
        self._write()
        if self.world > 1:
            print(f"bench.py[rank {self.rank}/{self.world} pid {os.getpid()}] {stage} +{self.last - self.t0:.1f}s {note}".rstrip(), file=sys.stderr, flush=True)

    def laggards(self) -> str:
        seen = {}
        for r in range(self.world):
            try:
                idx, name, pid = open(os.path.join(self.dir, f"rank_{r}")).read().split()
                seen[r] = (int(idx), name, pid)
            except (OSError, ValueError):
                seen[r] = (-1, "never-started", "?")
        low = min(v[0] for v in seen.values())
        return ", ".join(f"rank {r} (pid {v[2]}) last reached '{v[1]}'" for r, v in sorted(seen.items()) if v[0] == low)

    def _watch(self) -> None:
        while not self._stop.wait(min(5.0, max(0.2, self.timeout_s / 4))):
            if time.monotonic() - self.last > self.timeout_s:
                print(f"bench.py: rank {self.rank} made no progress for {self.timeout_s:.0f}s after '{self.stage}' — a collective some rank never entered? "
                      f"furthest behind: {self.laggards()}", file=sys.stderr, flush=True)
                os._exit(4)

    def done(self) -> None:
        self._stop.set()


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, selftest: bool, transport: str = "nccl") -> int:
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (torch.distributed.run on 127.0.0.1, one
    process per GPU) and hand its exit code on.  Refuses instead of shrinking: N visible GPUs or nothing."""
    import subprocess
    if not selftest:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if transport == "gloo" and have >= 1:
            pass                                    # --transport gloo: ranks may share a device (probe of the multi-rank path)
        elif have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {have}: refusing to run fewer ranks than asked for", file=sys.stderr, flush=True)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def join_ranks(dist, dev, want: int, backend: str) -> list:
    """Every rank adds a one over the process group: the sum is the number of ranks that really joined.  Returns each rank's
    (rank, device, pci bus id) — or exits non-zero with a one-line reason when fewer than `want` arrived."""
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    joined = int(one.item())
    if joined != want or dist.get_world_size() != want:
        print(f"bench.py: {joined} of {want} ranks joined the {backend} group (world size {dist.get_world_size()}): refusing to report a {want}-GPU line",
              file=sys.stderr, flush=True)
        sys.exit(3)
    me = {"rank": dist.get_rank(), "device": str(dev)}
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        me["pci_bus_id"] = getattr(pr, "pci_bus_id", None)
        me["name"] = pr.name
    everyone = [None] * want
    dist.all_gather_object(everyone, me)
    return everyone


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--cfg-pair", action="store_true", help="SURVEY 8e-2: ranks (2p,2p+1) split the cond/uncond forwards of clip p "
                    "(one 4.2 MB all-gather per step); needs an even --gpus. Default is one clip per rank.")
    ap.add_argument("--seq-parallel", action="store_true", help="SURVEY 8e-3: ONE clip on all ranks (strong scaling): every forward is "
                    "spread Ulysses-style over the ranks (K / V all-gather fallback when the heads do not divide); with --cfg-pair: 2 CFG branches x N/2 sequence shards")
    ap.add_argument("--fp8-storage", action="store_true", help="the reference's FP8 mode (test_svi.py:337): parameters stored as float8_e4m3fn; the exact "
                    "cast to bf16 happens once at bind time, arithmetic stays bf16 (a separate line, never the headline)")
    ap.add_argument("--fp8-mfma", action="store_true", help="opt-in MX-fp8 MLP (north_star 'bf16/fp8 MFMA'): FP8 weight storage + both MLP GEMMs of every block on "
                    "v_mfma_scale_f32_32x32x64_f8f6f4 with per-32-element activation scales.  Arithmetic the reference never performs (it computes in bf16): "
                    "a separately toleranced line (tests/test_gpu_mx8.py), never the headline")
    ap.add_argument("--fp8-all", action="store_true", help="opt-in, same caveats: --fp8-mfma --fp8-attn AND the block's other six projections (self-attention q, k, V^T, o; "
                    "cross-attention q, o) on the MX block-scaled fp8 matrix path (WanDiT.proj_fp8_mfma): every token-side GEMM of the step and QK^T on fp8")
    ap.add_argument("--fp8-attn", action="store_true", help="opt-in quantised QK^T (SVI_ATTN_QK8=1): every long-sequence self-attention quantises Q and K to MX e4m3 (one E8M0 "
                    "scale per 32 channels) and takes QK^T on v_mfma_scale_f32_32x32x64_f8f6f4; P·V stays bf16.  Arithmetic the reference never performs (its dispatch only "
                    "ACCEPTS a quantised-QK^T backend, wan_video_dit.py:116-147): a separately toleranced line (tests/test_gpu_attn_qk8.py), never the headline")
    ap.add_argument("--profile-all", action="store_true", help="bracket every tagged kernel family with HIP events inside the timed region (about 1 %% slower steps) "
                    "instead of the dominant kernel there and the full breakdown in 4 extra steps behind it")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None, help="replay each step's two forwards from one hipGraph (DenoiseLoop(graph=True)). "
                    "DEFAULT for the single-rank step (what the installed sampler does); per-kernel HIP events cannot be taken inside a replay, so "
                    "`roofline` / `roofline_all` then come from eager, instrumented steps run directly behind the timed region")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="eager launches in the timed region (rounds 1-3's behaviour); the dominant kernel is "
                    "then bracketed by HIP events inside the timed region itself")
    ap.add_argument("--transport", default="nccl", choices=["nccl", "gloo"], help="process-group backend of a multi-rank run.  nccl (= RCCL over xGMI): one GPU per "
                    "rank, the measured configuration.  gloo: the exchanges go through the host and the ranks may SHARE a device (rank r -> device r mod visible) — a "
                    "probe that drives the whole multi-rank code path (shard, exchange, max-over-ranks, aggregation) on a one-GPU box; its line says so and is not a scaling figure")
    ap.add_argument("--no-full-clip", dest="full_clip", action="store_false", help="skip the two COMPLETE clips timed behind the K-step region (the metric as defined, "
                    "not its extrapolation): clip 1 = first eager step + hipGraph capture + context-cache fill + the remaining replays + VAE decode + 8-bit frames; clip 2 = the "
                    "same stream's next clip on the resident loop (replays only) -> config.full_clip_s / value_full_clip / full_clip_steady_s")
    ap.add_argument("--window", type=int, default=0, metavar="K", help="BASELINE configs[2] as a workload: a rolling window of K clips (seed = k x 42, prompts cycled, "
                    "test_svi.py:424-476), sharded clip k -> rank k mod N, every clip denoised, decoded and turned into 8-bit frames, frames all-gathered, window stitched; "
                    "`value` is then the WINDOW's latent frames / s (wall clock over everything) and config.window holds the per-clip fixed cost")
    ap.add_argument("--window-ab", action="store_true", help="with --window: run the window a second time on a loop that re-captures its step graph for every clip "
                    "(the round 1-4 behaviour) and report the difference")
    ap.add_argument("--attn-gain", type=float, default=1.0, metavar="G", help="multiply every block's self_attn.norm_q / norm_k weight by G: the logits of the self-attention scale "
                    "with G^2 (random-init weights give diffuse attention, G = 1; a trained checkpoint's learned gains make it peaky).  The line reports the workgroups the "
                    "optimistic attention pass flagged for its second pass (config.attention_second_pass); tools/attn_stats.py sweeps the kernel alone")
    ap.add_argument("--prompt-tokens", type=int, default=64, metavar="N", help="valid (non-padded) tokens of the positive prompt (the negative prompt gets N // 2): the context's "
                    "remaining rows are the prompter's zero padding = one more distinct key.  Default 64 / 32; beyond 127 valid tokens the cross-attention leaves "
                    "flash_cross_resident_kernel (keys resident in LDS) for the streaming kernel")
    ap.add_argument("--budget-s", type=float, default=600.0, metavar="S", help="wall-clock budget of this invocation, counted from process start.  The K timed steps are never "
                    "trimmed; what is optional around them is: the two timed complete clips go when they would not fit, then the vendor yardstick.  The expected total is "
                    "printed on stderr after the warm-up and reported as config.wall_s_expected beside config.wall_s")
    ap.add_argument("--no-vendor", action="store_true", help="skip the vendor yardstick child (tools/yardstick.py: hipBLASLt / SDPA on the step's six hot shapes, same box, same "
                    "job; N = 1, c2 only) whose figures stand beside every family of roofline_all as `vendor`")
    ap.add_argument("--rccl-probe", action="store_true", help="N = 1 only: create the RCCL process group anyway (a one-rank communicator) so that the barrier, the tail all-gather and "
                    "the MAX-reduce of the timed region go through RCCL and the step's hipGraph is captured beside a live communicator — the part of the multi-GPU path a "
                    "one-GPU box can run on hardware")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="multi-rank runs: a rank that reaches no new stage (process group, join, warm-up, timed region, "
                    "clips) for this many seconds exits non-zero and names the rank(s) furthest behind; 0 = off")
    ap.add_argument("--selftest-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)      # --launcher-selftest: this rank never joins (tests the watchdog)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launcher-selftest", action="store_true", help="only start the ranks, join them over gloo on the CPU and print who joined "
                    "(tests/test_bench_launcher.py: the launch path without GPUs)")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker()
        return
    if args.fp8_all:
        args.fp8_mfma = args.fp8_attn = True
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, args.launcher_selftest, args.transport))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    hb = Heartbeat(rank, world, args.rank_timeout)
    if args.launcher_selftest:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        hb.mark("process-group", "gloo")
        if rank == args.selftest_hang_rank:
            time.sleep(3600)                                  # the rank that never enters the collective
        who = join_ranks(dist, torch.device("cpu"), args.gpus, "gloo")
        hb.mark("joined")
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": len(who), "ranks": who, "backend": "gloo"}), flush=True)
        dist.destroy_process_group()
        hb.done()
        return
    local = int(os.environ.get("SVI_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))      # SVI_BENCH_DEVICE: pin every rank to one device (a probe only)
    if args.transport == "gloo" and torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()          # ranks share devices when there are fewer devices than ranks
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU {local} ({torch.cuda.device_count() if torch.cuda.is_available() else 0} visible)", file=sys.stderr, flush=True)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist, who = None, [{"rank": 0, "device": str(dev), "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None),
                        "name": torch.cuda.get_device_properties(dev).name}]
    rccl = None
    if args.rccl_probe and world == 1:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.rccl_probe:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        hb.mark("start", f"device cuda:{local}, initialising {args.transport}")
        if args.transport == "gloo":
            dist.init_process_group("gloo")
            hb.mark("process-group", "gloo")
            who = join_ranks(dist, dev, args.gpus, "gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
            hb.mark("process-group", "nccl (RCCL)")
            who = join_ranks(dist, dev, args.gpus, "nccl (RCCL)")
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        hb.mark("joined")

    import synth
    import svi_hip
    from svi_hip import _lib

    wl = WORKLOADS[args.workload]
    T, H, W = wl["lat"]
    cfg = dict(getattr(synth, wl.get("model", "WAN_1_3B")))
    D, F, NL, heads = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"], cfg["dim"] // 128
    dit = svi_hip.WanDiT(eps=1e-6, num_heads=heads, **cfg)
    weights = device_weights(cfg, 0, dev)
    if args.fp8_storage or wl.get("fp8_storage") or args.fp8_mfma:
        args.fp8_storage = True
        weights = {k: v.to(torch.float8_e4m3fn) for k, v in weights.items()}
    if args.attn_gain != 1.0:
        for name in weights:
            if name.endswith("self_attn.norm_q.weight") or name.endswith("self_attn.norm_k.weight"):
                weights[name] = (weights[name].float() * args.attn_gain).to(weights[name].dtype)
    dit.bind(weights)
    hb.mark("model-bound")
    if args.fp8_mfma:
        dit.ffn_fp8_mfma(True)
    if args.fp8_all:
        dit.proj_fp8_mfma(True)
    if args.fp8_attn:
        from svi_hip import _lib as _L
        _L.set_switch("SVI_ATTN_QK8", 1)
    pair, units, sp_group, sp = None, world, None, False
    if args.seq_parallel and dist is not None:
        sp, units = True, 1
        if args.cfg_pair:
            from svi_hip.parallel import split_cfg_sequence
            pair, sp_group, _ = split_cfg_sequence()
            sp = dist.get_world_size(sp_group) > 1
    elif args.cfg_pair:
        assert dist is not None and world % 2 == 0, "--cfg-pair needs an even number of ranks"
        from svi_hip.parallel import CfgPair
        pair, pair_idx, units = CfgPair.split_world()
    graph_note = None
    graph_try = False
    if args.graph is None:
        # the single-rank step replays a hipGraph by default (DenoiseLoop's own default).  Beside an RCCL communicator (--gpus N > 1) the graph is TRIED: one
        # untimed probe step is captured and replayed inside a guard; the ranks then vote (MIN all-reduce) and either all keep the graph or all fall back to
        # eager launches, and the line says which ran and why (config.hip_graph, config.hip_graph_note).  A scaling run does not hinge on the capture.
        args.graph = pair is None and not sp
        graph_try = bool(args.graph) and dist is not None and args.transport == "nccl"
    loop = svi_hip.DenoiseLoop(dit, cfg_pair=pair, sp_group=sp_group, sequence_parallel=sp, graph=args.graph)
    eager_loop = svi_hip.DenoiseLoop(dit, cfg_pair=pair, sp_group=sp_group, sequence_parallel=sp, graph=False) if args.graph else loop
    eager_loop.scheduler = loop.scheduler
    spc = wl["steps_per_clip"]
    loop.scheduler.set_timesteps(spc, shift=5.0)

    # clip `rank` of the rolling window: seed = chunk_idx * 42 (test_svi.py:425), noise from the CPU generator
    lat = svi_hip.generate_noise((1, 16, T, H, W), seed=(0 if args.seq_parallel else rank // 2 if args.cfg_pair else rank) * 42, device="cpu", dtype=torch.float32).to(dev, torch.bfloat16)
    gen = torch.Generator(device=dev).manual_seed(1234)
    ctx_pos = torch.randn((1, wl["lc"], 4096), generator=gen, device=dev).to(torch.bfloat16)
    ctx_neg = torch.randn((1, wl["lc"], 4096), generator=gen, device=dev).to(torch.bfloat16)
    if not 1 <= args.prompt_tokens <= wl["lc"]:
        ap.error(f"--prompt-tokens must be within 1..{wl['lc']}")
    ctx_pos[:, args.prompt_tokens:] = 0
    ctx_neg[:, max(1, args.prompt_tokens // 2):] = 0
    ts_dev = loop.scheduler.timesteps.to(dev, torch.float32)
    cond = {}
    if cfg["has_image_input"]:      # I2V: y = mask(4) | VAE latent(16) as encode_images_adaptive builds it, CLIP tokens of the first frame
        yy = torch.randn((1, 20, T, H, W), generator=gen, device=dev)
        yy[:, :4] = 0
        yy[:, :4, 0] = 1
        cond = dict(y=yy.to(torch.bfloat16), clip_feature=torch.randn((1, 257, 1280), generator=gen, device=dev).to(torch.bfloat16))

    pose_ms = None
    if wl.get("pose"):      # dance variant (svi_video_dance.py:527-530): pose video -> dwpose_embedding -> add_condition, once per clip
        from svi_hip.pose import PoseEmbedder
        pe = PoseEmbedder.from_state_dict({k: torch.from_numpy(v) for k, v in synth.pose_state_dict(7, 16, D).items()})
        pose_video = (torch.rand((3, 4 * (T - 1) + 1, 8 * H, 8 * W), generator=gen, device=dev) * 255.0) * (torch.rand((3, 4 * (T - 1) + 1, 8 * H, 8 * W), generator=gen, device=dev) < 0.3)
        pe(pose_video)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cond["add_condition"] = pe(pose_video)
        e1.record()
        torch.cuda.synchronize()
        pose_ms = e0.elapsed_time(e1)
        del pose_video

    def one_step(i: int, lp=None) -> None:
        j = i % spc
        (lp or loop).step(lat, ts_dev[j:j + 1], loop.scheduler.step_delta(loop.scheduler.timesteps[j]), ctx_pos, ctx_neg, 5.0, **cond)

    def sync() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # as DenoiseLoop.sample() does for a clip: the prompt embeddings are loop constants, their projection and the per-block
    # cross-attention K / V^T are computed in the first forward that sees them (inside the warm-up here, inside step 0 of a clip)
    dit.context_cache(True)
    if graph_try:
        ok, why = 1.0, ""
        try:
            one_step(0)                    # eager on the capture stream + capture
            one_step(1)                    # a replay
            torch.cuda.synchronize()
        except Exception as ex:  # noqa: BLE001 — whatever the capture raised beside the communicator: say it, launch eagerly
            ok, why = 0.0, f"{type(ex).__name__}: {str(ex).splitlines()[0][:160]}"
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
        vote = torch.tensor([ok], device=dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if vote.item() < 1.0:
            graph_note = "hipGraph capture beside the RCCL communicator failed on " + ("this rank: " + why if not ok else "another rank") + " — every rank launches eagerly"
            print(f"bench.py[rank {rank}]: {graph_note}", file=sys.stderr, flush=True)
            args.graph = False
            loop = eager_loop = svi_hip.DenoiseLoop(dit, cfg_pair=pair, sp_group=sp_group, sequence_parallel=sp, graph=False)
            loop.scheduler.set_timesteps(spc, shift=5.0)
        else:
            graph_note = "captured and replayed beside the live RCCL communicator (probe step before the warm-up; all ranks agreed)"
    for i in range(args.warmup):
        one_step(i)
    sync()
    hb.mark("warmup-done", f"hip_graph={bool(args.graph)}")
    # ---- wall-clock plan (--budget-s): what is still ahead, priced with a step time taken now --------------------------------------------------------
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_probe = time.perf_counter()
    one_step(args.warmup)
    torch.cuda.synchronize()
    step_est = time.perf_counter() - t_probe
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    want_vendor = rank == 0 and world == 1 and not args.no_vendor and args.workload == "c2" and not args.seq_parallel
    plan = {"timed_steps": args.steps * step_est, "eager_profile_steps": 8 * step_est * 1.05, "vae": 0.0 if args.no_vae else 6.0,
            "full_clips": 2.0 * (spc * step_est + 1.5) + 3.0 if (args.full_clip and not args.no_vae and not wl.get("pose")) else 0.0,
            "window": (args.window * (2 if args.window_ab else 1) / max(world, 1)) * (spc * step_est + 1.5) if args.window > 0 else 0.0,
            "cpu_baseline": 110.0 if want_cpu else 0.0, "vendor_yardstick": 45.0 if want_vendor else 0.0}
    used = time.perf_counter() - T_PROCESS_START
    trimmed = []
    if args.budget_s > 0:
        def over():
            return used + sum(plan.values()) > args.budget_s
        if over() and plan["full_clips"]:
            plan["full_clips"] = 0.0
            args.full_clip = False
            trimmed.append("full clips")
        if over() and plan["vendor_yardstick"]:
            plan["vendor_yardstick"] = 0.0
            want_vendor = False
            trimmed.append("vendor yardstick")
    if dist is not None:            # every rank drops the same parts (the full clips hold collectives): the slowest rank's plan decides
        tv = torch.tensor([0.0 if args.full_clip else 1.0], device=dev)
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        if tv.item() > 0 and args.full_clip:
            args.full_clip = False
            plan["full_clips"] = 0.0
            trimmed.append("full clips (another rank's plan)")
    wall_expected = used + sum(plan.values())
    print(f"bench.py[rank {rank}/{world}]: {used:.0f} s used, step ~{step_est * 1e3:.0f} ms, expected total ~{wall_expected:.0f} s of --budget-s {args.budget_s:.0f}"
          + (f" (trimmed: {', '.join(trimmed)})" if trimmed else ""), file=sys.stderr, flush=True)
    # Over the timed region only the dominant kernel is bracketed by HIP events (`roofline`): every event record is a packet between two
    # kernels of the stream it measures, and the ~1440 records of a fully instrumented C2 step cost that step about 1 % (profiles/r3m_prof_events_ab.txt).
    # The breakdown over all kernel families (`roofline_all`, `kernel_ms_per_step`) comes from `prof_steps` fully instrumented steps run
    # BEHIND the timed region.  --profile-all keeps every family's events inside the timed region (rounds 1-3's behaviour).
    _lib.prof_select(None if args.profile_all else ["flash_self"])
    _lib.prof_enable(not args.graph)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    if dist is not None:
        # hand the clip's tail (motion) latents to every rank, as the rolling window stitches clips (test_svi.py:472-476)
        tail = lat[:, :, -1:].contiguous()
        gathered = [torch.empty_like(tail) for _ in range(world)]
        dist.all_gather(gathered, tail)
    sync()
    elapsed = time.perf_counter() - t0
    hb.mark("timed-done", f"{elapsed * 1000.0 / max(args.steps, 1):.1f} ms/step")
    prof_timed = _lib.prof_summary()
    _lib.prof_enable(False)
    _lib.prof_select(None)
    prof, prof_steps, roof_steps = prof_timed, args.steps, args.steps
    roof_source = "HIP events on the launch stream over the timed region"
    if args.graph:
        # a replayed hipGraph cannot carry per-kernel event pairs: the dominant kernel is timed over `roof_steps` EAGER steps of the same loop
        # state directly behind the timed region (only that kernel bracketed: the step is otherwise undisturbed), then every family
        loop.drop_graph()
        roof_steps = max(1, min(4, args.steps))
        _lib.prof_select(["flash_self"])
        _lib.prof_enable(True)
        for i in range(roof_steps):
            one_step(args.warmup + args.steps + i, eager_loop)
        torch.cuda.synchronize()
        prof_timed = _lib.prof_summary()
        _lib.prof_enable(False)
        _lib.prof_select(None)
        roof_source = (f"HIP events on the launch stream over {roof_steps} eager steps run directly behind the timed region (the timed region replays one hipGraph per "
                       "step, inside which no event pair can be taken; same kernels, same operands)")
    if not args.profile_all or args.graph:
        prof_steps = max(1, min(4, args.steps))
        _lib.prof_enable(True)
        for i in range(prof_steps):
            one_step(args.warmup + args.steps + roof_steps + i, eager_loop)
        torch.cuda.synchronize()
        prof = _lib.prof_summary()
        _lib.prof_enable(False)
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    finite = bool(torch.isfinite(lat.float()).all().item())
    # the last self-attention launch of the eager steps above: how many of its workgroups the optimistic pass handed to the second pass (0 on every
    # random-init forward measured so far; --attn-gain makes the logits peaky)
    second_pass = None
    if L_TOKENS(T, H, W) >= 2048 and not sp:
        import ctypes as _C
        fa, fb = _C.c_int32(), _C.c_int32()
        if _lib.lib().svi_attention_last_flagged(_lib.current_stream(), _C.byref(fa), _C.byref(fb)) == 0 and fb.value > 0:
            second_pass = {"flagged_workgroups": fa.value, "workgroups": fb.value, "fraction": round(fa.value / fb.value, 5), "attn_gain": args.attn_gain,
                           "what": "last self-attention launch of the instrumented eager steps: workgroups (256 query rows of one head) whose row sums left the range the "
                                   "first key tile's reference maximum covers and were recomputed by the complete kernel"}

    # VAE decode of the finished clip (fp32, as pipelines/svi_video.py:385-389), timed once on the same stream; for the I2V
    # model also the per-clip conditioning encode y = mask | VAE.encode([motion frame | zeros]) (svi_video.py:291-350)
    vae_ms, enc_ms = None, None
    if not args.no_vae:
        from svi_hip.vae import WanVideoVAE, device_vae_weights
        vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
        z = lat.float()
        vae.decode(z, device=dev)                      # warm-up (workspace allocation)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vid = vae.decode(z, device=dev)
        e1.record()
        torch.cuda.synchronize()
        vae_ms = e0.elapsed_time(e1)
        finite = finite and bool(torch.isfinite(vid).all().item())
        if cfg["has_image_input"]:
            first = vid[:, :, :1].permute(0, 2, 1, 3, 4)[0].contiguous()          # the decoded first frame as the motion frame [1,3,H,W]
            nf = 4 * (T - 1) + 1
            svi_hip.image_condition(vae, first, None, nf)
            torch.cuda.synchronize()
            e0.record()
            ycond = svi_hip.image_condition(vae, first, None, nf)
            e1.record()
            torch.cuda.synchronize()
            enc_ms = e0.elapsed_time(e1)
            finite = finite and bool(torch.isfinite(ycond.float()).all().item())

    if dist is not None:      # the slowest rank's decode and every rank's finiteness decide, as for the step time
        red = torch.tensor([vae_ms or 0.0, 0.0 if finite else 1.0, enc_ms or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        vae_ms = float(red[0].item()) if vae_ms is not None else None
        enc_ms = float(red[2].item()) if enc_ms is not None else None
        finite = red[1].item() == 0.0
    # ---- the metric as DEFINED — (clips x 21 latent frames) / wall time — timed, not extrapolated ----------------------------------
    def timed_clip(lp, k: int) -> dict:
        """One complete clip on loop `lp` the way the rolling window runs it: seeded noise -> [I2V: conditioning encode] -> DenoiseLoop.sample (all
        `spc` steps) -> VAE decode -> 8-bit frames; wall clock with device syncs around each stage."""
        from svi_hip.stream import video_to_u8
        noise = svi_hip.generate_noise((1, 16, T, H, W), seed=k * 42, device="cpu", dtype=torch.float32).to(dev, torch.bfloat16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = dict(cond)
        if cfg["has_image_input"] and not args.no_vae:
            c["y"] = svi_hip.image_condition(vae, first, None, 4 * (T - 1) + 1)
        out = lp.sample(noise, ctx_pos, ctx_neg, num_inference_steps=spc, cfg_scale=5.0, sigma_shift=5.0, **c)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        frames_u8 = video_to_u8(vae.decode(out.float(), device=dev)[0])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ok = bool(torch.isfinite(out.float()).all().item()) and tuple(frames_u8.shape) == (4 * (T - 1) + 1, 8 * H, 8 * W, 3)
        return {"s": t2 - t0, "denoise_s": t1 - t0, "decode_u8_s": t2 - t1, "ok": ok}

    full = None
    if args.full_clip and not args.no_vae and not wl.get("pose"):
        dit.context_cache(False)              # a stream starts cold: nothing of the timed region's cache entries or graph is reused
        clip_loop = svi_hip.DenoiseLoop(dit, cfg_pair=pair, sp_group=sp_group, sequence_parallel=sp, graph=args.graph, resident=True)
        first_clip = timed_clip(clip_loop, 0)          # eager first step + capture + cache fill + replays + decode + u8
        hb.mark("timed-done", "full clip 1 done")
        next_clip = timed_clip(clip_loop, 1)           # the stream's next clip: adopt() + replays only (no capture) + decode + u8
        hb.mark("timed-done", "full clip 2 done")
        full = {"first": first_clip, "next": next_clip, "captures": clip_loop.captures}
        if dist is not None:
            red = torch.tensor([first_clip["s"], next_clip["s"], first_clip["denoise_s"], next_clip["denoise_s"]], device=dev, dtype=torch.float64)
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            first_clip["s"], next_clip["s"], first_clip["denoise_s"], next_clip["denoise_s"] = (float(v) for v in red.tolist())
        finite = finite and first_clip["ok"] and next_clip["ok"]
        clip_loop.close()

    window = None
    if args.window > 0 and not args.no_vae and not args.seq_parallel and not wl.get("pose") and not cfg["has_image_input"]:
        from svi_hip.parallel import ClipParallel, clip_prompt_index, clip_seed, stitch_window
        from svi_hip.stream import video_to_u8
        K = args.window
        par = ClipParallel() if (dist is not None and pair is None) else None
        my = [k for k in range(K) if (k % world == rank if par is not None else True)] if pair is None else [k for k in range(K) if k % units == rank // 2]
        g2 = torch.Generator(device=dev).manual_seed(4321)
        alt_pos = torch.randn((1, wl["lc"], 4096), generator=g2, device=dev).to(torch.bfloat16)
        alt_pos[:, 48:] = 0
        prompts = [(ctx_pos, ctx_neg), (alt_pos, ctx_neg)]            # two prompts, cycled (test_svi.py:430-438); one negative prompt (test_svi.py:283)
        nf = 4 * (T - 1) + 1

        def run(resident: bool) -> dict:
            dit.context_cache(False)
            lp = svi_hip.DenoiseLoop(dit, cfg_pair=pair, graph=args.graph, resident=resident)
            sync()
            t0 = time.perf_counter()
            mine = {}
            for k in my:
                cp, cn = prompts[clip_prompt_index(k, len(prompts))]
                noise = svi_hip.generate_noise((1, 16, T, H, W), seed=clip_seed(k), device="cpu", dtype=torch.float32).to(dev, torch.bfloat16)
                lat_k = lp.sample(noise, cp, cn, num_inference_steps=spc, cfg_scale=5.0, sigma_shift=5.0)
                mine[k] = video_to_u8(vae.decode(lat_k.float(), device=dev)[0])
                hb.mark("timed-done", f"window clip {k} done")
            if par is not None:
                clips = par.all_gather_clips(mine, K, like=torch.empty((nf, 8 * H, 8 * W, 3), dtype=torch.uint8, device=dev))
            else:
                clips = [mine[k] for k in sorted(mine)]
            n_motion = 1
            video = torch.cat([c[:-n_motion] if i < len(clips) - 1 else c for i, c in enumerate(clips)], dim=0)      # parallel.stitch_window's rule on tensors
            sync()
            wall = time.perf_counter() - t0
            if dist is not None:
                tm = torch.tensor([wall], device=dev, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                wall = float(tm.item())
            want_frames = (nf - n_motion) * (len(clips) - 1) + nf
            assert video.shape[0] == want_frames == len(stitch_window([range(nf)] * len(clips), n_motion)), (video.shape, want_frames)
            res = {"wall_s": round(wall, 3), "clips": K, "clips_this_rank": len(my), "stitched_frames": int(video.shape[0]), "captures": lp.captures,
                   "checksum": int(video[::7, ::31, ::37].to(torch.int64).sum().item())}
            lp.close()
            return res
        window = {"resident": run(True)}
        if args.window_ab:
            window["recapture_per_clip"] = run(False)

    hb.mark("clips-done")
    ms_per_step = elapsed * 1000.0 / args.steps
    clip_s_dit = spc * ms_per_step / 1000.0
    clip_s = clip_s_dit + ((vae_ms or 0.0) + (enc_ms or 0.0) + (pose_ms or 0.0)) / 1000.0
    frames = float(T)
    value = units * frames / clip_s
    value_src = f"{spc} x ms_per_step (the K timed steady-state steps) + one VAE decode, per clip"
    full_cfg = {}
    if full is not None:
        # `value` stays the steady-state figure of the K timed steps (the contract's definition of a step); the timed clips stand beside it
        fs, ns = full["first"]["s"], full["next"]["s"]
        full_cfg = {"full_clip_s": round(fs, 3), "value_full_clip": round(units * frames / fs, 5),
                    "full_clip_steady_s": round(ns, 3), "value_full_clip_steady": round(units * frames / ns, 5),
                    "full_clip_vs_extrapolated": round(fs / clip_s, 4), "full_clip_steady_vs_extrapolated": round(ns / clip_s, 4),
                    "full_clip_breakdown": {"first": {k: round(v, 3) for k, v in full["first"].items() if k != "ok"},
                                            "next": {k: round(v, 3) for k, v in full["next"].items() if k != "ok"},
                                            "step_graph_captures_over_both_clips": full["captures"],
                                            "what": "first = a stream's first clip: seeded noise, eager step 1 on the capture stream (fills the context cache), hipGraph capture + "
                                                    f"instantiation, {spc - 1} replays, VAE decode, 8-bit frames; next = the same stream's next clip on the resident loop: inputs copied "
                                                    f"into the loop's tensors, {spc} replays (no capture), decode, 8-bit frames.  Wall clock, device-synchronised, max over ranks"}}
    if window is not None:
        w0 = window["resident"]
        value = w0["clips"] * frames / w0["wall_s"]
        value_src = f"wall clock of the whole {w0['clips']}-clip rolling window (denoise + decode + 8-bit frames + frame all-gather + stitch)"
        per_clip = w0["wall_s"] / max(w0["clips_this_rank"], 1)
        window["per_clip_s"] = round(per_clip, 3)
        window["per_clip_fixed_cost_s"] = round(per_clip - clip_s, 3)      # beyond spc steady-state steps + one decode
        if "recapture_per_clip" in window:
            w1 = window["recapture_per_clip"]
            window["graph_kept_across_clips_saves_s_per_clip"] = round((w1["wall_s"] - w0["wall_s"]) / max(w0["clips_this_rank"], 1), 3)
            window["same_video"] = w0["checksum"] == w1["checksum"]
    L = (T // 1) * (H // 2) * (W // 2)
    lc = wl["lc"]
    img_tok = 257 if cfg["has_image_input"] else 0
    # the algorithmic work of one reference forward (what the reference executes; the number BASELINE.md quotes) ...
    flops_forward = NL * (12 * L * D ** 2 + 4 * L * L * D + 4 * lc * D ** 2 + 4 * L * lc * D + 4 * L * D * F
                          + (4 * img_tok * D ** 2 + 4 * L * img_tok * D if img_tok else 0))
    # ... and what a bench step actually executes: the prompt-side projections (text embedding, cross-attention K / V of every block)
    # are loop constants served by the context cache, and block 0's self-attention third of the second CFG forward is shared
    # (forward_cfg_pair; not when the pose condition makes the branches differ).  dit_tflops is computed from THIS.
    per_fwd = NL * (8 * L * D ** 2 + 4 * L * L * D + 4 * L * D ** 2 + 4 * L * lc * D + 4 * L * img_tok * D + 4 * L * D * F) \
        + 2 * L * (cfg["in_dim"] * 4) * D + 2 * L * D * 64
    shared = 0 if (wl.get("pose") or pair or sp) else (8 * L * D ** 2 + 4 * L * L * D)
    flops_step = 2 * per_fwd - shared
    if pair:
        flops_step = per_fwd                 # a rank of a CFG pair runs one branch
    fl = prof_timed.get("flash_self", {"count": 0, "ms": 0.0})          # the dominant kernel: events over the timed region itself
    fp8_attn_on = args.fp8_attn or os.environ.get("SVI_ATTN_QK8", "0") not in ("", "0")
    roof = None
    if fl["count"]:
        per_launch_ms = fl["ms"] / fl["count"]
        alg = 4.0 * L * L * D / (dist.get_world_size(sp_group) if sp else 1)      # a sequence-parallel rank attends with heads / S
        ach = alg / (per_launch_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel from PMC counters: collected in separate rocprofv3 --pmc passes
        # (tools/profile_round.sh -> profiles/*_flash_pmc.json, corrected as MI355X_MICROARCH.md prescribes); not measurable live
        pmc = pmc_summaries() if args.workload == "c2" else {"why": "the PMC summaries are collected on the c2 workload"}
        fp = pmc.get("flash") or {}
        traffic, traffic_src = fp.get("hbm_bytes"), pmc.get("flash_file") or pmc.get("why")
        roof = {"kernel": "flash_fwd3_kernel + flash_fwd2_kernel<.., 2> (self-attention; one launch = the optimistic pass, on v_mfma_f32_16x16x32_bf16 since round 6 "
                          "(SVI_FLASH_M16=0: flash_fwd2_kernel<.., 1>), + the flagged second pass, which exits at once unless a row's exponentials left the optimistic range)", "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "mfma_busy_in_clock": fp.get("mfma_busy_in_clock"), "l2_hit_rate": fp.get("l2_hit_rate"),
                "algorithmic_flop_per_launch": alg, "launches": fl["count"], "ms_per_launch": round(per_launch_ms, 4), "source": roof_source}
        if fp8_attn_on:
            # half of the launch's FLOP (QK^T) runs on the MX fp8 pipe (dense peak 2x bf16), half (P·V) on bf16: the launch's bound is the harmonic mix;
            # the launch also holds the two quantiser kernels.  PMC: the summary of the fp8 kernel (profiles/<tag>_flash_qk8_pmc.json), never the bf16 kernel's.
            mixed = PEAK_BF16_TFLOPS / 0.75
            fq = pmc.get("flash_qk8") or {}
            roof.update({"kernel": "flash_fwd2_kernel<QK8> + 2 x mx8_quantize_kernel (opt-in SVI_ATTN_QK8: QK^T on v_mfma_scale_f32_32x32x64_f8f6f4, P·V bf16)", "peak": round(mixed, 1),
                         "frac": round(ach / mixed, 4), "peak_note": "QK^T at the MX fp8 dense peak (2 x bf16), P·V at the bf16 peak: 1 / (0.5 / 5000 + 0.5 / 2500) TFLOP/s",
                         "frac_of_bf16_peak": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": fq.get("hbm_bytes"),
                         "traffic_source": pmc.get("flash_qk8_file") or pmc.get("why") or "no PMC summary of the fp8 kernel on this tree",
                         "mfma_busy_in_clock": fq.get("mfma_busy_in_clock"), "l2_hit_rate": fq.get("l2_hit_rate")})
    # ---- every kernel family of the step against the roofline that bounds it (per rank; ms from HIP events on the launch stream) ----
    PEAK_HBM_TBS = 8.0
    shard = dist.get_world_size(sp_group) if sp else 1

    # The CFG pair's STACKED form (svi_dit.hip forward_pair: after the shared block-0 self-attention third, every row-local kernel runs ONCE over both
    # branches' rows): a launch then covers 2 L rows.  Work is counted in L-row launch EQUIVALENTS: S self-attention thirds and R rest-thirds per step.
    ffn1_n = (prof.get("gemm_ffn1") or {"count": 0})["count"] / max(prof_steps, 1)
    stacked = world == 1 and not pair and not sp and not wl.get("pose") and 0 < ffn1_n < 1.5 * NL
    S_eq, R_eq = 2 * NL - (1 if shared else 0), 2 * NL

    def fam(tag, work_per_launch, bound, what, total_fn=None, third=None):
        rec = prof.get(tag)
        if not rec or not rec["count"]:
            return None
        ms = rec["ms"] / prof_steps
        n = rec["count"] / prof_steps
        # third: which part of a block the family's launches belong to ("self" / "rest" / "ln": one self + two rest per block) — only used when stacked
        n_eq = n if not (stacked and third) else n * {"self": S_eq / NL, "rest": R_eq / NL, "ln": (S_eq + 2 * R_eq) / (3 * NL)}[third]
        total = total_fn(n_eq) if total_fn else work_per_launch * n_eq
        if bound == "mfma":
            ach, peak, unit = total / (ms * 1e-3) / 1e12, PEAK_BF16_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = total / (ms * 1e-3) / 1e12, PEAK_HBM_TBS, "TB/s"
        out = {"bound": bound, "what": what, "launches_per_step": round(n, 2), "algorithmic_per_step": total, "ms_per_step": round(ms, 3),
               "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4)}
        if stacked and third:
            out["launch_equivalents_per_step"] = round(n_eq, 2)       # the CFG pair is stacked: most launches of this family cover both branches' rows (2 L)
        return out
    Ls = L // shard if sp else L

    def distinct_keys(c):      # what ctx_tail_scan_kernel leaves on the device: rows up to and including the first of the identical suffix
        same = (c[0] == c[0, -1]).all(dim=-1)
        n = c.shape[1] - 1
        while n > 0 and bool(same[n - 1]):
            n -= 1
        return n + 1
    keys_walked = [distinct_keys(ctx_pos), distinct_keys(ctx_neg)] if os.environ.get("SVI_CROSS_DEDUP", "1") != "0" else [lc, lc]
    # default: the cross-attention query is normalised by the attention kernel as it reads q (the q projection's epilogue leaves the row statistics; the tag
    # then also holds the tiny kernel that folds them: 2 launches per block and branch, one [L, D] read + one write between them).  SVI_CROSS_FUSED=0:
    # the query is normalised in place by its own launch (under rmsnorm_rope: one more [L, D] read + write per block and branch)
    cross_fused = os.environ.get("SVI_CROSS_FUSED", "1") != "0"
    n_cross_units = NL * (1 if pair else 2)          # (block, branch) pairs per step and rank
    roof_all = {
        "flash_self": fam("flash_self", 4.0 * L * L * D / shard, "mfma", "4 L^2 D FLOP per launch (QK^T + PV, all heads)"),
        # the cross-attention walks the DISTINCT keys of the zero-padded prompt (svi_dit.hip ctx_tail: n + 1 of the 512): with a few dozen keys
        # the launch is a read of q and a write of o — HBM-bound — and the matrix work it executes is 4 L (n+1) D, not 4 L Lc D
        "flash_cross": fam("flash_cross", 4.0 * Ls * D, "hbm", f"q [L, D] bf16 read + o [L, D] bf16 written per launch; keys walked: {keys_walked} of {lc} context rows "
                           "(identical trailing rows of the zero-padded prompt count as one key)" +
                           ("; q is the raw projection, RMS-normalised as it is read; the tag also holds row_rs_kernel (the statistic): 2 launches per block and branch" if cross_fused else ""),
                           # fused: the tag holds the attention launches (one per block and branch) AND the statistic's launches (one per block where the pair is
                           # stacked, else one per block and branch): the bytes are counted per (block, branch) unit, not per launch
                           total_fn=(lambda n: (4.0 * Ls * D + (D // 64 + 1) * 4.0 * Ls) * n_cross_units) if cross_fused else None),
        # q | k are ONE N = 2D launch (4 L D^2 FLOP), V^T its own (2 L D^2): 3 L D^2 per launch on average; three 2 L D^2 launches with SVI_QK_FUSED=0
        "gemm_qkv": fam("gemm_qkv", (3.0 if os.environ.get("SVI_QK_FUSED", "1") != "0" else 2.0) * Ls * D * D, "mfma",
                        "q | k as one launch over the two weight matrices (4 L D^2 FLOP) + the V^T projection (2 L D^2)", third="self"),
        "gemm_attn_out": fam("gemm_attn_out", 2.0 * Ls * D * D, "mfma", "2 L D^2 FLOP per launch, gate + residual epilogue", third="self"),
        "gemm_cross": fam("gemm_cross", 2.0 * Ls * D * D, "mfma", "2 L D^2 FLOP per launch (cross-attention q and o; cached prompt K / V excluded)", third="rest"),
        "gemm_ffn1": fam("gemm_ffn1", 2.0 * Ls * D * F, "mfma", "2 L D F FLOP per launch, GELU-tanh epilogue", third="rest"),
        "gemm_ffn2": fam("gemm_ffn2", 2.0 * Ls * D * F, "mfma", "2 L D F FLOP per launch, gate + residual epilogue", third="rest"),
        "ln_modulate": fam("ln_modulate", 4.0 * Ls * D, "hbm", "2 L D bf16 read + written per launch", third="ln"),
        # q|k launch: [L, 2D] read + written; cross-attention q launch: [L, D] read + written -> mean of the two launch kinds per block
        # two launch kinds under one tag: the cross-attention q launch ([L, D] read + written, once per block and forward) and the q|k launch
        # ([L, 2D] read + written: every other launch of the tag)
        "rmsnorm_rope": fam("rmsnorm_rope", None, "hbm", "q|k launches: 8 L D bytes read + written, cross-attention q launches: 4 L D",
                            total_fn=(lambda n: (0.0 if cross_fused else 4.0 * Ls * D * R_eq) + 8.0 * Ls * D * S_eq) if stacked else
                            (lambda n: 8.0 * Ls * D * n) if cross_fused else
                            (lambda n: 4.0 * Ls * D * min(n, n_cross_units) + 8.0 * Ls * D * max(0.0, n - n_cross_units))),
    }
    if vae_ms is not None and (T, H, W) == (21, 60, 104):
        vflop = 2.754e14                      # SURVEY 8d, measured by flop counter: fp32 FLOP of one 81f@480x832 decode
        roof_all["vae_decode"] = {"bound": "mfma", "what": "2.754e14 fp32 FLOP per decode, executed as 3 f16 MFMA partial products per product where the input is "
                                  "bounded (two-term split) — algorithmic figure below is the f16-MFMA-equivalent 3 x 2.754e14", "launches_per_step": None,
                                  "algorithmic_per_step": 3 * vflop, "ms_per_step": round(vae_ms, 2), "achieved": round(3 * vflop / (vae_ms * 1e-3) / 1e12, 1),
                                  "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s (f16 MFMA equivalent)", "frac": round(3 * vflop / (vae_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "fp32_equivalent_tflops": round(vflop / (vae_ms * 1e-3) / 1e12, 1)}
    roof_all = {k: v for k, v in roof_all.items() if v is not None}
    if args.workload == "c2" and not sp:
        pm = pmc_summaries()
        for fam_name, key in (("flash_self", "flash"), ("gemm_ffn1", "gemm_ffn1"), ("gemm_ffn2", "gemm_ffn2"), ("flash_cross", "flash_cross"), ("vae_decode", "vae_conv")):
            if fam_name in roof_all and pm.get(key):
                roof_all[fam_name].update(traffic=pm[key].get("hbm_bytes"), mfma_busy_in_clock=pm[key].get("mfma_busy_in_clock"),
                                          l2_hit_rate=pm[key].get("l2_hit_rate"), pmc_source=pm.get(key + "_file"))
    if "flash_cross" in roof_all:
        fc = roof_all["flash_cross"]
        mean_keys = sum(keys_walked) / len(keys_walked)
        attn_launches = float(n_cross_units) if cross_fused else fc["launches_per_step"]
        fc["attention_launches_per_step"] = attn_launches
        fc["executed_tflops"] = round(4.0 * Ls * mean_keys * D * attn_launches / (fc["ms_per_step"] * 1e-3) / 1e12, 1)
        fc["reference_algorithmic_tflops"] = round(4.0 * Ls * lc * D * attn_launches / (fc["ms_per_step"] * 1e-3) / 1e12, 1)
        fc["query_rmsnorm_fused"] = cross_fused
    line = {
        "metric": {"c2": "denoised latent frames/sec, Wan2.1-1.3B 81f@832x480 50-step", "c1": "denoised latent frames/sec, Wan2.1-1.3B 17f@256x256 10-step",
                   "c4": "denoised latent frames/sec, Wan2.1-I2V-14B 81f@832x480 50-step",
                   "c5": "denoised latent frames/sec, Wan2.1-I2V-14B + pose embedder (dance) 81f@832x480 50-step, FP8 weight storage"}[args.workload],
        "value": round(value, 5), "unit": "latent frames/s", "n_gpus": len(who), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.seq_parallel else "weak", "vs_baseline": None,
        "dtype": ("bf16 (" + ", ".join((["MLP GEMMs"] if args.fp8_mfma else []) + (["q / k / v / o and cross q / o projections"] if args.fp8_all else []) + (["self-attention QK^T"] if fp8_attn_on else [])) + ": MX fp8 e4m3, opt-in)")
                 if (args.fp8_mfma or fp8_attn_on) else "bf16", "data": "synthetic (random-init weights of the named architecture, seeded noise/context)",
        "config": {"workload": wl["desc"] if window is None else
                   f"BASELINE configs[2] on {world} GPU(s), INDEPENDENT-clip form (T2V: a clip depends on its prompt and seed only): rolling window of "
                   f"{window['resident']['clips']} clips (seed = k x 42, 2 prompts cycled, clip k -> rank k mod N), each {wl['desc']}.  The reference's I2V rolling "
                   f"window hands clip k's last frames to clip k+1 (test_svi.py:472-476): one stream's clips are sequential there (svi_hip.StreamLoop) and ranks take "
                   f"whole streams, so this figure is an upper bound for a single I2V stream's multi-GPU throughput", "value_is": value_src, **full_cfg, "window": window, "step": (f"1 scheduler step = cond+uncond DiT forward ({NL} blocks each; the pose condition enters the conditional branch only, so the two "
                                                    "forwards share nothing) + CFG + Euler") if wl.get("pose") else
                   f"1 scheduler step = cond+uncond DiT forward ({NL} blocks each; block 0's self-attention, whose operands are identical in both, is computed once — outputs bit-identical to two separate forwards) + CFG + Euler",
                   "cfg_pair_stacked": bool(stacked),       # behind the shared block-0 self-attention, every row-local kernel runs once over both branches' rows (2 L); bit-identical
                   "steps_per_clip": spc, "tokens": L, "clips_per_gpu": round(units / world, 4),
                   "sp_cfg_pair": getattr(loop, "last_sp_form", None),      # sequence-parallel steps: "stacked pair" = both CFG branches stacked on every rank's rows
                   "parallelism": (f"one clip: {'cfg-pair x ' if pair else ''}sequence-parallel over {world} ranks" +
                                   (" (CFG pair stacked on every rank's rows: no CFG exchange)" if getattr(loop, "last_sp_form", None) == "stacked pair" else "") if args.seq_parallel else
                                   f"cfg-pair x{units} clips" if pair else f"clip-per-rank x{world}"),
                   "vae_decode_ms": None if vae_ms is None else round(vae_ms, 2),
                   "vae_condition_encode_ms": None if enc_ms is None else round(enc_ms, 2),
                   "pose_embedder_ms": None if pose_ms is None else round(pose_ms, 2),
                   "value_includes_vae_decode": vae_ms is not None,
                   "dit_only_value": round(units * frames / clip_s_dit, 5),
                   "dit_tflops": round(flops_step / (dist.get_world_size(sp_group) if sp else 1) / (ms_per_step * 1e-3) / 1e12, 1),
                   "flop_per_step_executed": flops_step, "flop_per_forward_reference": flops_forward,
                   "ranks": who, "rccl": rccl,
                   "transport": ("nccl (RCCL), ONE rank (--rccl-probe): a live communicator beside the step graph; barrier, tail all-gather and MAX-reduce go through it, "
                                 "degenerate — a code-path probe, not a scaling figure") if (world == 1 and args.rccl_probe) else None if world == 1 else ("nccl (RCCL), one GPU per rank" if args.transport == "nccl" else
                                                         f"gloo through the host, {len({w.get('pci_bus_id') for w in who})} distinct device(s) under {world} ranks: a probe of "
                                                         "the multi-rank code path, NOT a scaling measurement"),
                   "hip_graph": bool(args.graph),
                   "attention": ("self-attention QK^T on the MX block-scaled fp8 matrix path (Q, K quantised per call: e4m3, one E8M0 scale per 32 channels), P·V and the softmax bf16 / fp32 — "
                                 "NOT the reference's arithmetic, opt-in, separately toleranced (tests/test_gpu_attn_qk8.py)") if fp8_attn_on else "bf16",
                   "weights": ("float8_e4m3fn storage; EVERY token-side GEMM of the block (MLP, self-attention q / k / V^T / o, cross-attention q / o) on the MX block-scaled fp8 matrix path "
                               "(activations e4m3 with one E8M0 scale per 32 elements; prompt-side K / V bf16), everything else bf16 — NOT the reference's arithmetic, opt-in, separately toleranced") if args.fp8_all else
                   ("float8_e4m3fn storage; MLP GEMMs on the MX block-scaled fp8 matrix path (activations e4m3 with one E8M0 scale per 32 elements), everything else "
                               "bf16 — NOT the reference's arithmetic, opt-in, separately toleranced") if args.fp8_mfma else
                   "float8_e4m3fn storage, cast to bf16 at bind (reference FP8 mode)" if args.fp8_storage else "bf16",
                   "outputs_finite": finite},
        "roofline": roof,
        "roofline_all": roof_all,
        "kernel_ms_per_step": {k: round(v["ms"] / prof_steps, 3) for k, v in prof.items()},
        "kernel_ms_source": ("HIP events of every tagged kernel family inside the timed region" if (args.profile_all and not args.graph) else
                             f"{prof_steps} fully instrumented eager steps behind the timed region" + (" (which replays one hipGraph per step)" if args.graph else
                             " (inside it only the dominant kernel is bracketed by events; `roofline` is from those)")),
    }
    line["config"].update({"attention_second_pass": second_pass, "prompt_tokens": [args.prompt_tokens, max(1, args.prompt_tokens // 2)],
                           "hip_graph_note": graph_note, "rccl_probe": bool(args.rccl_probe) or None,
                           "budget_s": args.budget_s, "budget_trimmed": trimmed or None, "wall_s_expected": round(wall_expected, 1)})
    if rank == 0 and want_vendor:
        # the vendor libraries on the same shapes, same box, same job (after every timed part of this process; the child has the GPU to itself)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        y = vendor_yardstick()
        line["vendor_yardstick"] = y
        by = {r["shape"]: r for r in y.get("rows", [])} if isinstance(y, dict) else {}

        def vend(fam_name, shapes, eq_per_step, note):
            f = roof_all.get(fam_name)
            if not f or not all(sh in by for sh in shapes):
                return
            key = "vendor_bare_ms" if "vendor_bare_ms" in by[shapes[0]] else next((k for k in ("vendor_sdpa_flash_ms", "vendor_sdpa_efficient_ms", "vendor_sdpa_default_ms") if k in by[shapes[0]]), None)
            if key is None:
                return
            v_ms = sum(by[sh][key] for sh in shapes)
            o_ms = sum(by[sh]["ours_ms"] for sh in shapes)
            f["vendor"] = {"ms_per_step": round(v_ms * eq_per_step, 3), "isolated_ms_per_L_row_launch": round(v_ms, 4), "ours_isolated_ms_per_L_row_launch": round(o_ms, 4),
                           "ours_over_vendor_time": round(o_ms / v_ms, 3), "what": note}
        bare = "torch.matmul = hipBLASLt, NOTHING fused (no bias, activation, gate or residual: ours does all of them in the same launch)"
        vend("gemm_qkv", ["qk", "v"], S_eq, bare + "; q|k [L,1536]x[3072,1536]^T + v [L,1536]x[1536,1536]^T per self-attention third")
        vend("gemm_attn_out", ["attn_out"], S_eq, bare)
        vend("gemm_cross", ["v"], 2 * R_eq, bare + "; the 1536^2 projection, twice per block and branch (cross-attention q and o)")
        vend("gemm_ffn1", ["ffn1"], R_eq, bare)
        vend("gemm_ffn2", ["ffn2"], R_eq, bare)
        vend("flash_self", ["self_attention"], S_eq, "F.scaled_dot_product_attention [1, 12, L, 128] bf16 (PyTorch-ROCm's flash backend); ours through the public seam "
             "(v transposed by a launch of its own and the softmax scale applied inside the kernel: the DiT's instance needs neither)")
    line["config"]["wall_s"] = round(time.perf_counter() - T_PROCESS_START, 1)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["config"]["wall_s"] = round(time.perf_counter() - T_PROCESS_START, 1)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    hb.mark("line-printed")
    hb.done()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

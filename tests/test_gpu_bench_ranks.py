"""-m gpu: bench.py's MULTI-RANK path on a device (VERDICT r3 missing #1 / next #5).  No 8-GPU node has ever been available to the
round driver, so `bench.py --gpus N` had only ever run its launcher on the CPU.  `--transport gloo` lets N ranks share the one GPU of the
test box: every kernel runs on the device, the exchanges (tail all-gather, the CFG pair's noise_pred all-gather, the sequence-parallel
all-to-all / all-gather, the MAX-reduce of elapsed time and the finiteness vote) go through gloo — the same `torch.distributed` calls
the RCCL run makes.  Checked here on the small workload: the line's aggregation fields; the lines at the C2 size are recorded under
profiles/ by tools (they are a code-path probe, never a scaling figure: the ranks time-share one GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")


def run(args, timeout=900, vae=False):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, *args, "--transport", "gloo", "--workload", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + ([] if vae else ["--no-vae"]),
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def common(line, world):
    assert line["n_gpus"] == world and sorted(w["rank"] for w in line["config"]["ranks"]) == list(range(world))
    assert "gloo" in line["config"]["transport"] and "NOT a scaling measurement" in line["config"]["transport"]
    assert line["config"]["outputs_finite"] and line["ms_per_step"] > 0 and line["value"] > 0
    assert line["cpu_baseline"] is None and line["config"]["rccl"] is None


def test_clip_per_rank_two_ranks():
    line = run(["--gpus", "2"])
    common(line, 2)
    assert line["scaling"] == "weak" and line["config"]["parallelism"] == "clip-per-rank x2" and line["config"]["clips_per_gpu"] == 1.0
    assert line["config"]["hip_graph"] is True            # each rank's own clip: the single-rank step, replayed from its hipGraph
    # value = 2 clips x 5 latent frames / (10 steps x ms_per_step)
    assert abs(line["value"] - 2 * 5 / (10 * line["ms_per_step"] / 1000.0)) < 1e-3 * line["value"]


def test_cfg_pair_two_ranks():
    line = run(["--gpus", "2", "--cfg-pair"])
    common(line, 2)
    assert line["config"]["parallelism"] == "cfg-pair x1 clips" and line["config"]["clips_per_gpu"] == 0.5 and line["config"]["hip_graph"] is False
    assert abs(line["value"] - 1 * 5 / (10 * line["ms_per_step"] / 1000.0)) < 1e-3 * line["value"]


def test_cfg_pair_times_sequence_parallel_four_ranks():
    line = run(["--gpus", "4", "--cfg-pair", "--seq-parallel"])
    common(line, 4)
    assert line["scaling"] == "strong" and "sequence-parallel over 4 ranks" in line["config"]["parallelism"]
    assert abs(line["value"] - 1 * 5 / (10 * line["ms_per_step"] / 1000.0)) < 1e-3 * line["value"]


def test_rolling_window_and_full_clips_two_ranks():
    """BASELINE configs[2] as bench.py runs it (--window K): 3 clips over 2 ranks (clip k -> rank k mod 2), each denoised, decoded and turned into
    8-bit frames, the frames all-gathered and the window stitched (every clip but the last loses its motion frame); `value` is the window's frames
    per second of wall clock.  The resident loop captures its step graph once per rank; the A/B against re-capturing per clip gives the same video.
    Beside it the two timed complete clips (config.full_clip_s / full_clip_steady_s)."""
    line = run(["--gpus", "2", "--window", "3", "--window-ab"], vae=True)
    common(line, 2)
    w = line["config"]["window"]
    assert w["resident"]["clips"] == 3 and w["resident"]["clips_this_rank"] == 2 and w["resident"]["stitched_frames"] == 16 * 2 + 17
    assert w["resident"]["captures"] == 1 and w["recapture_per_clip"]["captures"] == 2 and w["same_video"] is True
    assert abs(line["value"] - 3 * 5 / w["resident"]["wall_s"]) < 1e-3 * line["value"]
    assert "rolling window of 3 clips" in line["config"]["workload"]
    c = line["config"]
    assert c["full_clip_s"] > 0 and c["full_clip_steady_s"] > 0 and c["full_clip_breakdown"]["step_graph_captures_over_both_clips"] == 1
    assert abs(c["value_full_clip"] - 2 * 5 / c["full_clip_s"]) < 1e-2 * c["value_full_clip"]          # (full_clip_s is printed to the millisecond)


def test_full_clip_single_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--workload", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = line["config"]
    assert line["n_gpus"] == 1 and c["hip_graph"] is True and c["window"] is None
    assert c["full_clip_s"] > c["full_clip_steady_s"] * 0.5 and c["full_clip_breakdown"]["step_graph_captures_over_both_clips"] == 1
    # the steady clip is the extrapolation's twin: 10 replayed steps + decode (+ 8-bit frames, input copies); generous bound on the tiny workload
    assert 0.8 < c["full_clip_steady_vs_extrapolated"] < 2.0, c


def test_one_rank_rccl_communicator_beside_the_step_graph():
    """--rccl-probe: what a one-GPU box can run of the RCCL path on hardware — the process group is RCCL (one rank), the barrier / tail all-gather / MAX-reduce of
    the timed region go through it, and the step's hipGraph is captured and replayed beside the live communicator behind the guarded probe step that the
    multi-GPU default uses (VERDICT r5 next #7b).  Either outcome of the guard is legal; the line must say which ran."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--rccl-probe", "--workload", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-vae", "--no-vendor"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = line["config"]
    assert line["n_gpus"] == 1 and c["rccl"] is not None and c["rccl_probe"] is True and "ONE rank" in c["transport"]
    assert c["hip_graph_note"] and (c["hip_graph"] is True) == ("captured and replayed" in c["hip_graph_note"])
    assert c["outputs_finite"] and line["ms_per_step"] > 0
    assert c["wall_s_expected"] > 0 and c["wall_s"] > 0 and c["budget_s"] == 600.0


def test_peaky_attention_and_long_prompt_lines():
    """--attn-gain / --prompt-tokens: the line carries the second-pass census of the last self-attention and the prompt's valid tokens."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--workload", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-vae", "--no-full-clip", "--attn-gain", "3",
                        "--prompt-tokens", "200"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = line["config"]
    assert c["prompt_tokens"] == [200, 100] and c["outputs_finite"]
    assert "keys walked: [201, 101]" in line["roofline_all"]["flash_cross"]["what"]
    # c1 has 1280 tokens: below the long-sequence kernel's threshold, so there is no second pass to count
    assert c["attention_second_pass"] is None

"""Host-side logic of bench.py and of the reference-copy recipe that no GPU is needed for:
  * the PMC gate: a rocprofv3 --pmc summary under profiles/ is quoted only when EVERY kernel source hash recorded with it equals
    the file being timed;
  * oracle/build_ref.py: `oracle/_ref` is a faithful, git-ignored copy of the six reference modules of the path, importable through oracle/ref_shim.py
    (what bench.py's cpu_baseline child times as kind "reference")."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = os.environ.get("SVI_REFERENCE", "/root/reference")


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def test_pmc_summaries_are_quoted_only_on_matching_sources(tmp_path, monkeypatch):
    import bench
    csrc = tmp_path / "stable-video-infinity_amd" / "csrc"
    prof = tmp_path / "profiles"
    csrc.mkdir(parents=True)
    prof.mkdir()
    (csrc / "svi_attention.hip").write_text("// attention v1\n")
    (csrc / "svi_gemm.hip").write_text("// gemm v1\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert "why" in bench.pmc_summaries()                                    # nothing collected yet
    hashes = {n: _sha(csrc / n) for n in ("svi_attention.hip", "svi_gemm.hip")}
    (prof / "t1_source_hashes.json").write_text(json.dumps(hashes))
    (prof / "t1_flash_pmc.json").write_text(json.dumps({"hbm_bytes": 123.0, "mfma_busy_in_clock": 0.5}))
    (prof / "t1_gemm_ffn1_pmc.json").write_text(json.dumps({"hbm_bytes": 456.0}))
    got = bench.pmc_summaries()
    assert got["tag"] == "t1" and got["flash"]["hbm_bytes"] == 123.0 and got["gemm_ffn1"]["hbm_bytes"] == 456.0 and "gemm_ffn2" not in got
    (csrc / "svi_gemm.hip").write_text("// gemm v2: one kernel file changed after the profile\n")
    assert "why" in bench.pmc_summaries() and "flash" not in bench.pmc_summaries()      # the attention hash still matches — not enough
    # a newer, matching summary wins over the stale one
    hashes2 = {n: _sha(csrc / n) for n in ("svi_attention.hip", "svi_gemm.hip")}
    (prof / "t2_source_hashes.json").write_text(json.dumps(hashes2))
    (prof / "t2_flash_pmc.json").write_text(json.dumps({"hbm_bytes": 789.0}))
    assert bench.pmc_summaries()["flash"]["hbm_bytes"] == 789.0


def test_committed_pmc_summaries_match_the_committed_kernel_sources():
    """The round's last profile (tools/profile_round.sh) must have been collected on the kernel sources as they are committed: bench.py on a fresh box
    then carries roofline.traffic / mfma_busy_in_clock."""
    import bench
    got = bench.pmc_summaries()
    assert "tag" in got, got
    assert got["flash"]["hbm_bytes"] > 0 and 0 < got["flash"]["mfma_busy_in_clock"] < 1
    assert got["gemm_ffn1"]["hbm_bytes"] > 0 and got["gemm_ffn2"]["mfma_busy_in_clock"] > 0


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "diffsynth")), reason="the reference checkout is not on this box")
def test_oracle_ref_is_a_faithful_ignored_copy():
    from oracle import build_ref
    assert build_ref.build() and build_ref.available()
    man = json.load(open(os.path.join(build_ref.OUT, "MANIFEST.json")))
    assert sorted(man["files"]) == sorted("diffsynth/" + f for f in build_ref.FILES)
    for rel, sha in man["files"].items():
        assert hashlib.sha256(open(os.path.join(REF, rel), "rb").read()).hexdigest() == sha          # byte-identical to the reference's file
    # never in history: the directory is git-ignored (it still travels to the GPU box: no .gpurunignore lists it)
    r = subprocess.run(["git", "check-ignore", "-q", os.path.join("oracle", "_ref", "MANIFEST.json")], cwd=ROOT)
    assert r.returncode == 0
    assert subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.strip() == ""
    assert not os.path.exists(os.path.join(ROOT, ".gpurunignore")) or "oracle/_ref" not in open(os.path.join(ROOT, ".gpurunignore")).read()


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "diffsynth")), reason="the reference checkout is not on this box")
def test_oracle_ref_imports_and_runs_the_reference_block():
    """In a child process (the shim registers `diffsynth` namespace packages in sys.modules): the copy under oracle/_ref gives the reference's DiTBlock,
    and it computes what the oracle's restatement computes."""
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import build_ref, ref_shim
from oracle import wan_dit_oracle as wdo
import synth
dit, vae, fm = ref_shim.load(build_ref.OUT)
assert dit.__file__.startswith(build_ref.OUT), dit.__file__
c = dict(synth.TINY_DIT, num_layers=1)
sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(5, **c).items()}
blk = dit.DiTBlock(False, c["dim"], 1, c["ffn_dim"], 1e-6).eval()
blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")}, strict=True)
f, h, w = 2, 3, 4
L = f * h * w
x = torch.from_numpy(synth.randn(6, 1, L, c["dim"])); ctx = torch.from_numpy(synth.randn(7, 1, 9, c["dim"])); tm = torch.from_numpy(0.3 * synth.randn(8, 1, 6, c["dim"]))
fr = dit.precompute_freqs_cis_3d(128)
freqs = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1), fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
with torch.no_grad():
    ref = blk(x, ctx, tm, freqs)
    got = wdo.dit_block(sd, "blocks.0.", x, ctx, tm, wdo.rope_table_3d(128, (f, h, w)), wdo.DiTConfig(dim=c["dim"], ffn_dim=c["ffn_dim"], num_heads=1, num_layers=1))
print(float((got - ref).norm() / ref.norm()))
''' % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert float(r.stdout.strip().splitlines()[-1]) < 2e-5


def test_committed_default_line_counts_stacked_launches_in_row_equivalents():
    """The CFG pair is stacked at the C2 size: a row-local launch covers both branches' rows.  bench.py must count such families in L-row launch
    equivalents (59 self-attention thirds + 60 rest-thirds per step), or their roofline fractions halve; the fused cross-attention's bytes are counted per
    (block, branch) unit (60 per step), not per launch (its tag also holds the statistic's launches).  Checked on the committed driver-shaped line, which
    also carries the two TIMED complete clips beside the extrapolated value."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, "profiles", "r5zz_bench_default.json")
    if not os.path.exists(p):
        import pytest
        pytest.skip("no committed r5zz line")
    j = json.load(open(p))
    c = j["config"]
    assert c["cfg_pair_stacked"] is True and c["hip_graph"] is True
    ra = j["roofline_all"]
    want = {"gemm_qkv": 118, "gemm_attn_out": 59, "gemm_cross": 120, "gemm_ffn1": 60, "gemm_ffn2": 60, "ln_modulate": 179}
    for k, n in want.items():
        assert abs(ra[k]["launch_equivalents_per_step"] - n) < 0.01, (k, ra[k])
        assert ra[k]["launches_per_step"] < n
    assert 0.40 < ra["gemm_ffn1"]["frac"] < 0.60 and 0.45 < ra["gemm_ffn2"]["frac"] < 0.65 and 0.6 < ra["ln_modulate"]["frac"] < 0.9
    fc = ra["flash_cross"]
    assert fc["query_rmsnorm_fused"] is True and fc["attention_launches_per_step"] == 60 and fc["launches_per_step"] == 90
    assert abs(fc["algorithmic_per_step"] - 60 * (4.0 * 32760 * 1536 + 25 * 4.0 * 32760)) < 1e6 and 0.35 < fc["frac"] < 0.7
    assert fc["traffic"] and abs(fc["traffic"] - 4.0 * 32760 * 1536) < 0.1 * 4.0 * 32760 * 1536      # PMC: what crosses the fabric per launch IS the algorithmic q + o
    r = j["roofline"]
    assert r["traffic"] and r["mfma_busy_in_clock"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the family sum is the step
    assert abs(sum(j["kernel_ms_per_step"].values()) - j["ms_per_step"]) < 0.03 * j["ms_per_step"]
    # the metric as defined, timed: one complete clip as a stream starts, one as it continues — within 1 % of 50 x ms_per_step + the decode
    assert abs(c["value_full_clip"] - 21.0 / c["full_clip_s"]) < 1e-3 and 0.99 < c["full_clip_vs_extrapolated"] < 1.01 and 0.99 < c["full_clip_steady_vs_extrapolated"] < 1.01
    assert c["full_clip_breakdown"]["step_graph_captures_over_both_clips"] == 1

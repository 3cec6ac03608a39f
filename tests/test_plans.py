"""Launch planners (host logic, no GPU): which GEMM kernel and which attention split the launchers choose for the shapes of the C2 clip, of a
sequence-parallel rank's shard, and under the environment switches.  Pure arithmetic behind the C ABI (svi_gemm_plan, svi_attention_plan)."""
import pytest

from svi_hip import _lib as L

L_TOK, D, F = 32760, 1536, 8960
DEFAULT_256 = L.gemm_plan(L_TOK, D, D)          # 259 / 260 (four / two phases per K tile): whichever the library was built to prefer
assert DEFAULT_256 in (259, 260)


@pytest.fixture(autouse=True)
def clean_switches():
    yield
    for k in ("SVI_GEMM_KERNEL", "SVI_FLASH_SPLIT", "SVI_FLASH_KERNEL"):
        L.set_switch(k, None)


@pytest.mark.parametrize("M,N,K,want", [
    (L_TOK, D, D, 256), (L_TOK, F, D, 256), (L_TOK, D, F, 256), (D, L_TOK, D, 256),          # the single-rank C2 shapes: whole rounds of 256^2 tiles
    (L_TOK // 4, D, D, 192), (L_TOK // 4, D, F, 192), (L_TOK // 4, F, D, 256),                # P = 4 shard: 192 -> 256 tiles for N = 1536; ffn1 keeps 256-wide tiles
    (L_TOK // 2, D, D, 192), (L_TOK // 2, F, D, 256), (L_TOK // 3, D, D, 192), (L_TOK // 6, D, F, 192),
    (D, L_TOK // 4, D, 256),                                                                  # the shard's V^T projection: 6 x 43 192-wide tiles would need two rounds
    (200, 264, 136, 128), (1000, 520, 192, 128), (4096, 4096, 72, 128),                       # small problems / K not a multiple of 64: the 128^2 kernel
])
def test_gemm_kernel_choice(M, N, K, want):
    assert L.gemm_plan(M, N, K) == (DEFAULT_256 if want == 256 else want)          # 256: "a 256^2 loop" (the build's default one)


def test_gemm_kernel_choice_follows_the_switch_and_the_skinny_hint():
    assert L.gemm_plan(64, 4096, 4096, skinny=True) == 0 and L.gemm_plan(64, 4096, 4096) == 128
    L.set_switch("SVI_GEMM_KERNEL", 260)
    assert L.gemm_plan(L_TOK // 4, D, D) == 260                   # "never the 192-wide tile"
    L.set_switch("SVI_GEMM_KERNEL", 192)
    assert L.gemm_plan(L_TOK, D, D) == 192 and L.gemm_plan(300, 128, 64) == DEFAULT_256     # N < 192: not this kernel (a forced 256-row tile stays 256 wide)
    L.set_switch("SVI_GEMM_KERNEL", 128)
    assert L.gemm_plan(L_TOK, F, D) == 128
    L.set_switch("SVI_GEMM_KERNEL", 259)
    assert L.gemm_plan(L_TOK, D, F) == 259 and L.gemm_plan(L_TOK // 4, D, D) == 259          # four phases per K tile, wherever a 256-row tile runs
    L.set_switch("SVI_GEMM_KERNEL", 257)                                                         # not a kernel (any more): ignored
    assert L.gemm_plan(L_TOK, D, F) == DEFAULT_256


def test_gemm_kernel_choice_uses_the_device_cu_count():
    """The round arithmetic (256 x 192 vs 256^2 tiles, the half-a-chip threshold) is done for the part's own CU count."""
    assert L.gemm_plan(L_TOK // 4, D, D, compute_units=256) == 192          # 32 x 6 = 192 tiles on 256 CUs: 192-wide tiles fill the round
    assert L.gemm_plan(L_TOK // 4, D, D, compute_units=192) == DEFAULT_256  # ... on 192 CUs the 256-wide tiles ARE one whole round
    assert L.gemm_plan(20 * 256, 4 * 256, 512, compute_units=304) == 128    # 80 tiles < half of 304 CUs: the 128^2 kernel
    assert L.gemm_plan(20 * 256, 4 * 256, 512, compute_units=128) != 128     # ... more than half of 128 CUs: a 256-row tile


@pytest.mark.parametrize("sq,skv,heads,want", [
    (L_TOK, L_TOK, 12, dict(kernel=2, whole=1536, pieces=1, workgroups=1536)),        # C2 single rank: 6.0 rounds, nothing to cut
    (L_TOK, L_TOK, 3, dict(kernel=2, whole=256, pieces=2, workgroups=512)),           # P = 4 rank: 384 items -> 256 whole + 128 in two key halves
    (L_TOK, L_TOK, 6, dict(kernel=2, whole=768, pieces=1, workgroups=768)),           # P = 2 rank: three whole rounds
    (L_TOK, L_TOK, 2, dict(kernel=2, whole=256, pieces=1, workgroups=256)),           # P = 6 rank: one whole round
    (L_TOK, L_TOK, 1, dict(kernel=2, whole=0, pieces=2, workgroups=256)),             # P = 12 rank: half a round -> every item in two pieces
    (75600, 75600, 2, dict(kernel=2, whole=512, pieces=3, workgroups=752)),           # 720p, a two-head group: 592 items -> 80 left over, three pieces each
    (L_TOK, 512, 12, dict(kernel=1, whole=3072, pieces=1, workgroups=3072)),          # text cross-attention: the short-key kernel, 128-row q-blocks
    (8190, 4200, 3, dict(kernel=2, whole=96, pieces=1, workgroups=96)),               # key axis too short to cut (pieces keep >= 4096 keys)
])
def test_attention_launch_plan(sq, skv, heads, want):
    assert L.attention_plan(sq, skv, heads) == want


def test_attention_plan_follows_the_switches():
    L.set_switch("SVI_FLASH_SPLIT", 1)
    assert L.attention_plan(L_TOK, L_TOK, 3)["pieces"] == 1
    L.set_switch("SVI_FLASH_SPLIT", 2)
    assert L.attention_plan(L_TOK, L_TOK, 12) == dict(kernel=2, whole=0, pieces=2, workgroups=3072)
    L.set_switch("SVI_FLASH_SPLIT", 4)
    assert L.attention_plan(L_TOK, 9000, 12)["pieces"] == 2       # 9000 keys: at most two pieces of >= 4096
    L.set_switch("SVI_FLASH_SPLIT", None)
    L.set_switch("SVI_FLASH_KERNEL", 1)
    assert L.attention_plan(L_TOK, L_TOK, 12)["kernel"] == 1
    L.set_switch("SVI_FLASH_KERNEL", None)
    # another chip (304 compute units): 1536 = 5 x 304 + 16 -> the 16 items left over are cut in four
    assert L.attention_plan(L_TOK, L_TOK, 12, compute_units=304) == dict(kernel=2, whole=1520, pieces=4, workgroups=1584)


def test_gemm_operands_of_4_gib_take_the_kernel_with_64_bit_addresses():
    """Guards: the 256-row kernels reach A and W through buffer descriptors with 32-bit byte offsets; a problem whose A or W holds
    2^31 elements or more must never run on them (it would read zeros past the wrap) — the planner sends it to the 128^2 kernel (64-bit pointers)."""
    M = 75600 * 2                                          # a stacked CFG pair at 720p
    assert L.gemm_plan(M, D, F) != 128 and M * F < 2 ** 31     # A = [151200, 8960]: 1.35e9 elements, still a 256-row tile
    big_m = 2 ** 31 // F + 1
    assert big_m * F >= 2 ** 31 and L.gemm_plan(big_m, D, F) == 128
    assert L.gemm_plan(1024, 2 ** 31 // 4096 + 1, 4096) == 128    # ... and the same for the weight operand
    for forced in (192, 259, 260):
        L.set_switch("SVI_GEMM_KERNEL", forced)
        try:
            assert L.gemm_plan(big_m, D, F) == 128                 # a forced 256-row kernel does not override the guard
        finally:
            L.set_switch("SVI_GEMM_KERNEL", None)
